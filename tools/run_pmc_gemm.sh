# PMC passes over the dense-layer probe (SQ counters only, one group per run, kernel-trace only; each pass under its own timeout)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_g; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pg$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pg$i -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_gemm1.py 4e6 > /tmp/pg$i.log 2>&1
  cp /tmp/pg$i/*counter_collection.csv $OUT/g$i.csv 2>/dev/null || tail -5 /tmp/pg$i.log
done
ls -la $OUT
