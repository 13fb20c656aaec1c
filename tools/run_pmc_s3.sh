# PMC passes over the split-bf16 dense-layer probe (SQ counters only, one group per run, kernel-trace only; each pass under its own timeout)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_s3; mkdir -p $OUT
export HNR_S3_DBG=999          # probe_s3.py: timing part only, product kernel
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/ps$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/ps$i -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_s3.py --rows 4000000 --reps 2 > /tmp/ps$i.log 2>&1
  cp /tmp/ps$i/*counter_collection.csv $OUT/g$i.csv 2>/dev/null || tail -5 /tmp/ps$i.log
done
ls -la $OUT
