cd $GRAFT_REPO_ROOT
for i in 1 2; do
for a in 1 0; do
  TS_TRAIN_LAYOUT_ARENA=$a python bench.py --mode train-graph --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena', $a, 'graph ms', d['ms_per_step'])"
done; done
TS_TRAIN_LAYOUT_ARENA=1 python bench.py --mode train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena 1 eager ms', d['ms_per_step'])"
TS_TRAIN_LAYOUT_ARENA=0 python bench.py --mode train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena 0 eager ms', d['ms_per_step'])"
python -m pytest tests/test_train_step_gpu.py tests/test_conv_autograd_gpu.py tests/test_aggregator_gpu.py -x -q -m gpu 2>&1 | tail -3
