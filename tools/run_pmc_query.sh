# PMC passes over the query probe (one counter group per run, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_q; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY" \
           "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pq$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pq$i -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pq$i.log 2>&1
  cp /tmp/pq$i/*counter_collection.csv $OUT/g$i.csv 2>/dev/null || tail -5 /tmp/pq$i.log
done
ls -la $OUT
