# end-of-round soak at HEAD: wider random sweeps than the committed tests run, repeated-pass determinism, the pipelined determinism check
cd $GRAFT_REPO_ROOT
for seed in 101 102 103; do
  echo "## fuzz_ops seed $seed"; timeout 1500 python tests/fuzz_ops.py --n 60 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -16
  echo "## fuzz_e2e inference seed $seed"; timeout 1500 python tests/fuzz_e2e.py --n 16 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -4
  echo "## fuzz_e2e train seed $seed"; TS_FUZZ_TRAIN=1 timeout 1500 python tests/fuzz_e2e.py --n 8 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -4
done
echo "## stress_determinism 300"; timeout 1200 python tools/exp/stress_determinism.py 300 2>&1 | grep -v amdgpu.ids | tail -6
