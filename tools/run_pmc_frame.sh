# per-kernel busy fractions of one bench frame (HNR_PMC_CMD="<script> <args>" for another workload): texture addresser (vector-memory front end), VALU, MFMA, LDS
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3f; mkdir -p $OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pf$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pf$i -o p -- python3 ${HNR_PMC_CMD:-$G/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg} > /tmp/pf$i.log 2>&1
  cp /tmp/pf$i/*counter_collection.csv $OUT/fr_pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pf$i.log
done
python3 - <<'PY'
import csv, collections, os, json
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3f'
per=collections.defaultdict(dict)
for g in (1,2):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open('%s/fr_pmc_g%d.csv'%(out,g))):
        acc[r['Kernel_Name'].split('(')[0][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        for c,x in v.items(): per[k][c]=max(x)
rows=[]
for k,v in per.items():
    if 'GRBM_GUI_ACTIVE' not in v or v['GRBM_GUI_ACTIVE']<8*30000: continue
    cyc=v['GRBM_GUI_ACTIVE']/8.0
    rows.append((cyc,k,dict(ms=round(cyc/1.9e6,3), ta=round(v.get('TA_BUSY_avr',0)/cyc,2), valu=round(v.get('SQ_ACTIVE_INST_VALU',0)*4/1024/cyc,2),
        mfma=round(v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc,2), lds=round(v.get('SQ_LDS_IDX_ACTIVE',0)/256/cyc,2),
        ldsconf=round(v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1),2), wait=round(v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1),2))))
rows.sort(reverse=True)
for cyc,k,d in rows: print(k, d)
json.dump({k:d for _,k,d in rows}, open(out+'/frame_busy.json','w'), indent=1)
PY
