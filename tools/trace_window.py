#!/usr/bin/env python3
"""Prints a window of consecutive kernels (start, duration, queue) from the middle of a rocprofv3 --kernel-trace CSV.
Usage: python tools/trace_window.py <kernel_trace.csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mid = len(rows) * 3 // 4
t0 = int(rows[mid]['Start_Timestamp'])
for r in rows[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 20]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  q{r.get('Queue_Id')} {r['Kernel_Name'][:50]}")
