"""Kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV.  usage: timeline.py dir [first-kernel-substr]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
first = sys.argv[2] if len(sys.argv) > 2 else 'k_tile_hist'
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']][-1]
prev = None
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f}  {r['Kernel_Name'][:44]:44s} dur {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}")
    prev = e
