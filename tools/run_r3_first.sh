# round 3, first GPU contact: baseline train-step kernel stats, PMC evidence for chain_ws_kernel, power / clock trace
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3a; mkdir -p $OUT
# 1. train leg kernel stats (round-2 kernels)
rm -rf /tmp/prof_t
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o tr -- python3 $G/tools/probe_train.py --steps 10 > $OUT/probe_train.json 2>/tmp/err_t.txt
cp /tmp/prof_t/*kernel_stats.csv $OUT/train_v0_kernel_stats.csv 2>/dev/null; tail -2 /tmp/err_t.txt
# 2. PMC passes over the bench frame (kernel-trace only; one counter group per run)
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 420 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pc$i -o p -- python3 $G/bench.py --no-cpu-baseline --no-train-leg --steps 2 --warmup 1 > /tmp/pc$i.log 2>&1
  echo "pmc group $i rc=$?"
  cp /tmp/pc$i/*counter_collection.csv $OUT/pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pc$i.log
  cp /tmp/pc$i/*kernel_trace.csv $OUT/pmc_g${i}_trace.csv 2>/dev/null
done
# 3. power / clock trace during 300 back-to-back frames
timeout 600 python3 $G/tools/power_trace.py --steps 300 --out $OUT/power_trace.csv > $OUT/power_summary.json 2>$OUT/power_err.txt
cat $OUT/power_summary.json; tail -3 $OUT/power_err.txt
ls -la $OUT
