cd $GRAFT_REPO_ROOT
for a in "1 64 32 272 480" "1 128 32 272 480" "1 128 32 136 240" "1 32 32 136 240" "2 64 64 136 240"; do
  TS_X6P_TRACE=-2 python tools/exp/x6p_wgtimes.py $a 2>&1 | grep -v amdgpu.ids
done
