"""Kernel timeline (with queue ids, so overlap between streams shows) of the last `n` steps from a rocprofv3
--kernel-trace CSV.  usage: timeline2.py dir [n_last_kernels]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))[-n:]
t0 = int(rows[0]['Start_Timestamp'])
qs = {}
for r in rows:
    q = qs.setdefault(r.get('Queue_Id', '?'), len(qs))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').split('(')[0][:28]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  q{q} {'    ' * q}{name:28s} {(e - s) / 1e3:8.1f} us")
