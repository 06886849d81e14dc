# LDS / issue counters of the ping-pong kernel on 128 -> 32, 4 x 272 x 480 (separate --pmc passes, no tracing beside them)
mkdir -p gpurun_out/r06/pmc; O=$GRAFT_REPO_ROOT/gpurun_out/r06/pmc
cd /tmp; export TMPDIR=/tmp
CASE="python tools/exp/x6_pmc_case.py 4 128 32 1 272 480 x6"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_STALL"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && $CASE" > $O/p$i.log 2>&1
  F=$(find $O/p$i -name "*.db" | head -1)
  if [ -n "$F" ]; then (cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py $F "%ig_conv_x6p%") ; else tail -3 $O/p$i.log; fi
  rm -rf $O/p$i
done
