cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" CLIP=0 python tools/exp/dbg_train_graph.py 2>&1 | grep "^graph losses\|^graph ms\|^graph grad" | cut -c1-200; }
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_SERIALIZE_KERNEL=3
run HIP_LAUNCH_BLOCKING=1
