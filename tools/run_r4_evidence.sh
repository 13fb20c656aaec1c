# Round-4 evidence for the forward frame + the training step (one gpurun call): kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes),
# SQ / GRBM counter groups (MFMA busy), training-step traffic.  Summaries: tools/collect_traffic.py, collect_pmc.py, collect_train_traffic.py.
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r4_evidence; rm -rf $OUT; mkdir -p $OUT
B="python3 $G/bench.py --no-cpu-baseline --no-train-leg --no-f32-anchor"
rm -rf /tmp/st; timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o s -- $B --steps 4 --warmup 1 > $OUT/bench_stats.log 2>&1
mkdir -p $OUT/stats; cp /tmp/st/*kernel_stats.csv $OUT/stats/ 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- $B --steps 2 --warmup 1 > /tmp/pmc_$c.log 2>&1; echo "$c rc=$?"
  mkdir -p $OUT/pmc_$c/x; cp /tmp/pmc_$c/*counter_collection.csv $OUT/pmc_$c/x/ 2>/dev/null
done
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  timeout 420 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pc$i -o p -- $B --steps 2 --warmup 1 > /tmp/pc$i.log 2>&1; echo "pmc group $i rc=$?"
  cp /tmp/pc$i/*counter_collection.csv $OUT/pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pc$i.log
done
T=$G/gpurun_out/r4_evidence/train; mkdir -p $T
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmct_$c -o p -- python3 $G/tools/probe_train.py --steps 4 > /tmp/pmct_$c.log 2>&1; echo "train $c rc=$?"
  mkdir -p $T/pmc_$c/x; cp /tmp/pmct_$c/*counter_collection.csv $T/pmc_$c/x/ 2>/dev/null
done
rm -rf /tmp/stt; timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stt -o s -- python3 $G/tools/probe_train.py --steps 10 > $T/probe_train.json 2>/dev/null
cp /tmp/stt/*kernel_stats.csv $T/train_kernel_stats.csv 2>/dev/null
cd $G && python tools/collect_traffic.py gpurun_out/r4_evidence r04 | tail -12 && python tools/collect_pmc.py gpurun_out/r4_evidence r04 && python tools/collect_train_traffic.py gpurun_out/r4_evidence/train r04 | tail -8
cp profiles/r04_*.json gpurun_out/r4_evidence/ 2>/dev/null; ls gpurun_out/r4_evidence
