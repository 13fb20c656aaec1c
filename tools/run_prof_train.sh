cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3d; mkdir -p $OUT
rm -rf /tmp/prof_t
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o tr -- python3 $G/tools/probe_train.py --steps 10 > $OUT/probe_train.json 2>/tmp/err_t.txt
cp /tmp/prof_t/*kernel_stats.csv $OUT/train_kernel_stats.csv 2>/dev/null; tail -2 /tmp/err_t.txt
