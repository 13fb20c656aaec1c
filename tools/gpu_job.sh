#!/bin/bash
# The one GPU-box runner: gpurun -- 'bash tools/gpu_job.sh <tag> <job> [<job> ...]'.  Output goes to gpurun_out/<tag>/.
# Jobs (run in the order given):
#   tests[:<pytest args>]  the -m gpu suite (or the given selection)
#   smoke                  __graft_entry__.smoke()
#   bench[:<args>]         bench.py, default arguments unless given; prints the line's headline fields
#   stats                  rocprofv3 --kernel-trace --stats of a 4-step frame-only bench  -> bench_kernel_stats.csv
#   traffic                FETCH_SIZE / WRITE_SIZE passes (separate runs)                 -> tools/collect_traffic.py <tag>
#   pmc                    SQ / GRBM counter groups (separate runs, no other tracing)     -> tools/collect_pmc.py <tag>
#   train_traffic          FETCH_SIZE / WRITE_SIZE passes over tools/probe_train.py       -> tools/collect_train_traffic.py <tag>
#   train_stats            kernel stats of tools/probe_train.py
#   train_timeline[:args]  per-launch timeline of the last step of tools/probe_train.py             -> train_timeline.txt
#   py:<script and args>   python3 <script ...> with stdout+stderr in <script>.txt
# PMC passes never combine --pmc with sys/hip/hsa tracing (gpurun refuses that), and the profiled program is python3 itself.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
G=$GRAFT_REPO_ROOT; O=$G/gpurun_out/$TAG; mkdir -p "$O"
FRAME="python3 $G/bench.py --no-cpu-baseline --no-train-leg --no-f32-anchor"
export TMPDIR=/tmp
for job in "$@"; do
  name=${job%%:*}; arg=""; [ "$job" != "$name" ] && arg=${job#*:}
  case $name in
    tests)
      timeout 3000 python3 -m pytest ${arg:-tests} -x -q -m gpu > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.txt"; tail -5 "$O/pytest_gpu.txt";;
    smoke)
      timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1; echo "smoke rc=$?"; tail -2 "$O/smoke.txt";;
    bench)
      timeout 1500 python3 bench.py $arg > "$O/bench.txt" 2> "$O/bench.err"; echo "bench rc=$?"; tail -2 "$O/bench.err"
      python3 tools/show_bench.py "$O/bench.txt";;
    stats)
      (cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o s -- $FRAME --steps 4 --warmup 1 > "$O/stats.log" 2>&1)
      cp /tmp/st/*kernel_stats.csv "$O/bench_kernel_stats.csv" 2>/dev/null; python3 tools/show_stats.py "$O/bench_kernel_stats.csv" 2>/dev/null | head -16;;
    traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && rm -rf /tmp/pmc_$c && timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- $FRAME --steps 2 --warmup 1 > /tmp/pmc_$c.log 2>&1); echo "$c rc=$?"
        mkdir -p "$O/pmc_$c/x"; cp /tmp/pmc_$c/*counter_collection.csv "$O/pmc_$c/x/" 2>/dev/null
      done
      python3 tools/collect_traffic.py "gpurun_out/$TAG" "${TAG%%_*}" | tail -12;;
    pmc)
      i=0
      for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" \
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "GRBM_GUI_ACTIVE GRBM_COUNT TA_BUSY_avr TA_TA_BUSY_sum"; do
        i=$((i+1))
        (cd /tmp && rm -rf /tmp/pc$i && timeout 420 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pc$i -o p -- $FRAME --steps 2 --warmup 1 > /tmp/pc$i.log 2>&1); echo "pmc group $i rc=$?"
        cp /tmp/pc$i/*counter_collection.csv "$O/pmc_g$i.csv" 2>/dev/null || tail -5 /tmp/pc$i.log
      done
      python3 tools/collect_pmc.py "gpurun_out/$TAG" "${TAG%%_*}";;
    train_traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && rm -rf /tmp/pmct_$c && timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmct_$c -o p -- python3 $G/tools/probe_train.py --steps 4 > /tmp/pmct_$c.log 2>&1); echo "train $c rc=$?"
        mkdir -p "$O/train/pmc_$c/x"; cp /tmp/pmct_$c/*counter_collection.csv "$O/train/pmc_$c/x/" 2>/dev/null
      done
      python3 tools/collect_train_traffic.py "gpurun_out/$TAG/train" "${TAG%%_*}" | tail -8;;
    train_stats)
      (cd /tmp && rm -rf /tmp/stt && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stt -o s -- python3 $G/tools/probe_train.py --steps 10 > "$O/probe_train.json" 2>/dev/null)
      cp /tmp/stt/*kernel_stats.csv "$O/train_kernel_stats.csv" 2>/dev/null;;
    train_timeline)
      (cd /tmp && rm -rf /tmp/stl && timeout 420 rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -o s -- python3 $G/tools/probe_train.py --steps 6 ${arg} > "$O/probe_train_tl.json" 2>/dev/null)
      python3 tools/timeline.py /tmp/stl/*kernel_trace.csv "$O/train_timeline.txt" march_kernel 12 | tail -3;;
    py)
      f=$(echo "$arg" | awk '{print $1}'); timeout 1500 python3 $arg > "$O/$(basename "$f" .py).txt" 2>&1; echo "$f rc=$?"; tail -12 "$O/$(basename "$f" .py).txt";;
    *) echo "unknown job $job";;
  esac
done
