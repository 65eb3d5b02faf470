#!/usr/bin/env python3
"""Timeline of the last `step` of a rocprofv3 kernel trace (…_kernel_trace.csv): per kernel start offset (us), duration, queue.
tools/ktimeline.py <kernel_trace.csv> <first kernel of a step (substring)> [max lines]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = t0
for r in rows[i0:i1][:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-44:]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-44s q%-3s start %9.1f  dur %8.1f  end %9.1f" % (name, r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3, e / 1e3))
print("step span %.1f us" % ((int(rows[i1 - 1]["End_Timestamp"]) - t0) / 1e3))
