# wgrad workgroups-per-CU sweep on the training step (tools/exp: experiment script)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
python -m pytest tests/test_conv_autograd_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for g in 1 2 4; do
  TS_WGRAD_GROUPS_PER_CU=$g python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groups/CU', $g, 'ms_per_step', d['ms_per_step'])"
done
bash tools/exp/train_trace.sh 2>&1 | tail -30
