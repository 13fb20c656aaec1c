# PMC passes over one bench frame, printed for the per-sample MLP kernels (merge stage, colour feature, mix-up)
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3m; mkdir -p $OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm$i -o p -- python3 $G/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg > /tmp/pm$i.log 2>&1
  cp /tmp/pm$i/*counter_collection.csv $OUT/mg_pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pm$i.log
done
python3 - <<'PY'
import csv, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3m'
for g in (1,2,3,4):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    try: rows=list(csv.DictReader(open('%s/mg_pmc_g%d.csv'%(out,g))))
    except Exception as e: print(g, e); continue
    for r in rows:
        per[r['Kernel_Name'].split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in per.items():
        if "merge_wp" in k or "mlp3_kernel" in k or "mixfinal" in k: print(g,k,{c:max(x) for c,x in v.items()})
PY
