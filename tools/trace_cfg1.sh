# usage (GPU box): bash tools/trace_cfg1.sh  -> gpurun_out/trace_cfg1.txt : per-step kernel time vs wall time of one-segment batches
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_cfg1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg1 --steps 50 --warmup 5 --no-cpu-baseline > $out/trace_cfg1.log 2>&1 || exit 1
python3 - <<'P' > $out/trace_cfg1.txt
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_cfg1/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 40 steps: find k_tile_hist launches as step starts
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_tile_hist") or "k_tile_hist" in r["Kernel_Name"]]
sel = rows[starts[-41]:starts[-1]]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
wall = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(sel, sel[1:])]
print(f"40 steps: {len(sel)} kernels ({len(sel)/40:.1f} per step), busy {busy/40/1e3:.1f} us/step, wall {wall/40/1e3:.1f} us/step, mean gap {sum(gaps)/len(gaps)/1e3:.2f} us, median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us")
import collections
d = collections.defaultdict(list)
for r in sel: d[r["Kernel_Name"].split("(")[0][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda x: -sum(x[1])): print(f"{k:42s} {len(v)/40:5.1f}/step  {sum(v)/len(v)/1e3:7.2f} us")
P
rm -rf $out/trace_cfg1
