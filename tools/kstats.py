#!/usr/bin/env python3
"""Condensed view of a rocprofv3 kernel_stats.csv: name (shortened), calls, average us, total ms.  tools/kstats.py <csv> [n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"HIP_vector_type<(\w+), (\d)u>", r"\1\2", name)
    name = name.split("(")[0][-70:]
    print("%-70s %5d  avg %9.1f us  total %8.2f ms" % (name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
