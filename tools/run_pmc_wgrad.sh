cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3c; mkdir -p $OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pw$i -o p -- python3 $G/tools/probe_wgrad.py > /tmp/pw$i.log 2>&1
  cp /tmp/pw$i/*counter_collection.csv $OUT/wg_pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pw$i.log
done
python3 - <<'PY'
import csv, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3c'
for g in (1,2,3):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open('%s/wg_pmc_g%d.csv'%(out,g))):
        per[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in per.items():
        if 'h2wgrad_kernel' in k or 'h2lin' in k: print(g,k,{c:max(x) for c,x in v.items()})
PY
