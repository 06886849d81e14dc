# MFMA utilisation of the x6 layers IN ISOLATION (100 back-to-back launches of one layer, nothing else on the chip): the figure of the pipelined
# bench (profiles/r06_k3_mfma_util_pmc.txt) divides a kernel's matrix-busy cycles by a duration that other streams' kernels stretch.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
for v in 1 0; do
  (cd /tmp && TS_X6P=$v timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $O/mfma_iso$v -o m -- bash -c "cd $GRAFT_REPO_ROOT && python tools/exp/x6p_check.py --child time" > $O/mfma_iso$v.log 2>&1)
  M=$(find $O/mfma_iso$v -name "*.db" | head -1)
  python tools/mfma_util.py $M "%ig_conv_x6%" > $O/mfma_util_isolated_x6p$v.txt
  rm -rf $O/mfma_iso$v
done
# and the pipelined bench's figure with one pass at a time (three streams of ONE pass still overlap)
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $O/mfma1 -o m -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --batch 4 --frames-in-flight 1" > $O/mfma1.log 2>&1)
M=$(find $O/mfma1 -name "*.db" | head -1)
python tools/mfma_util.py $M > $O/mfma_util_b4_one_pass_at_a_time.txt
rm -rf $O/mfma1
