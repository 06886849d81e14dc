mkdir -p gpurun_out/r04/pmc; O=$GRAFT_REPO_ROOT/gpurun_out/r04/pmc
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E 'SQ_[A-Z0-9_]+' | sort -u > $O/sq_counters.txt; wc -l $O/sq_counters.txt
CASE="python tools/exp/x6_pmc_case.py 4 64 32 1 272 480 x6"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && $CASE" > $O/p$i.log 2>&1
  F=$(find $O/p$i -name "*.db" | head -1)
  if [ -n "$F" ]; then (cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py $F "%ig_conv_x6%") ; else tail -3 $O/p$i.log; fi
  rm -rf $O/p$i
done
