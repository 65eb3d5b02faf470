"""Workloads, launcher, executors and CPU baselines behind bench.py (the driver's contract lives there)."""
