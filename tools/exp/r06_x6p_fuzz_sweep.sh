# a wider random-geometry sweep of the ping-pong x6 kernel than the committed test runs (tools/exp/x6p_check.py --child fuzz):
# several seeds, strided input / output views, both tile heights forced, the default rule, split-K forced
cd $GRAFT_REPO_ROOT
for seed in 11 12 13; do
  for e in "TS_X6P_MIN_WGS=1 TS_X6P_HR=8" "TS_X6P_MIN_WGS=1 TS_X6P_HR=4" "TS_X6P_MIN_WGS=1" "TS_X6P_MIN_WGS=1 TS_X6P_KSPLIT=2" "TS_X6P_MIN_WGS=1 TS_X6P_KSPLIT=3 TS_X6P_HR=4" "TS_X6P_WGS=8 TS_X6P_MIN_WGS=1"; do
    echo "## seed $seed $e"
    env $e TS_FUZZ_SEED=$seed TS_FUZZ_CASES=120 timeout 600 python tools/exp/x6p_check.py --child fuzz 2>&1 | grep -v amdgpu.ids | tail -8
  done
done
