set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06q
rm -rf $O; mkdir -p $O
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $O/mfma -o m -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --batch 4" > $O/mfma.log 2>&1)
M=$(find $O/mfma -name "*.db" | head -1)
python tools/mfma_util.py $M > $O/mfma_util_b4.txt
rm -rf $O/mfma
python tools/layer_table.py > $O/layer_table_b1.txt 2>&1; python tools/layer_table.py --batch 4 > $O/layer_table_b4.txt 2>&1
TS_X6P=0 python tools/layer_table.py --batch 4 > $O/layer_table_b4_x6p0.txt 2>&1
