# kernel trace of one bench frame (forward only); prints the launches of the last step in order
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-leg --steps 1 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/tr1/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last occurrence of march_kernel marks the start of the last frame
idx = max(i for i, r in enumerate(rows) if 'march_kernel' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 0.15:
        print("%8.3f ms  +%8.3f  %s" % (d, (int(r['Start_Timestamp']) - t0) / 1e6, r['Kernel_Name'][:110]))
PY
