# timeline of ONE training step (the last of the run): every kernel in start order with its duration and the idle gap before it
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3t; mkdir -p $OUT
rm -rf /tmp/prof_tt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tt -o tr -- python3 $G/tools/probe_train.py --steps 4 > $OUT/probe_train.json 2>/tmp/err_tt.txt
python3 - <<'PY'
import csv, glob, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3t'
f=glob.glob('/tmp/prof_tt/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
starts=[i for i,r in enumerate(rows) if 'train_counts_kernel' in r['Kernel_Name'] or 'march_kernel' in r['Kernel_Name']]
# the last step starts at the last march_kernel
i0=max(i for i,r in enumerate(rows) if 'march_kernel' in r['Kernel_Name'])
# include the pack kernels just before (h2_pack etc. come after march?) -- keep from i0-6
i0=max(0,i0-8)
prev=None; tot=0; gap_tot=0
with open(out+'/train_step_timeline.txt','w') as o:
    t0=int(rows[i0]['Start_Timestamp'])
    for r in rows[i0:]:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        gap=(s-prev) if prev is not None else 0
        prev=e; tot+=e-s; gap_tot+=max(gap,0)
        o.write('%9.1f us  dur %8.1f  gap %6.1f  %s\n'%((s-t0)/1e3,(e-s)/1e3,gap/1e3,r['Kernel_Name'].split('(')[0][:70]))
    o.write('kernels %.1f us, gaps %.1f us, span %.1f us, launches %d\n'%(tot/1e3,gap_tot/1e3,(prev-t0)/1e3,len(rows)-i0))
print(open(out+'/train_step_timeline.txt').read()[-6000:])
PY
