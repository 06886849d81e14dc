"""Worker of tests/test_ddp_gpu.py (not collected): one rank of a world-size-2 data-parallel training step of the REAL aggregator
on the GPU.  RCCL refuses two ranks on one device, so the single-GPU test box runs the collectives over gloo (TS_BENCH_BACKEND) with
both ranks on device 0 (TS_BENCH_DEVICE); everything else -- HIP forward / backward, SyncBatchNorm statistics exchange between the
BatchNorm kernels, bucketed gradient all-reduce launched from backward hooks, fused clip + RMSprop -- is the production path.
Rank r trains on sample r of a planted scene; rank 0 saves the averaged gradients and the updated parameters."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scene(dev, items):
    import synth
    H, W = 128, 192
    sc = synth.stereo_sequence(synth.SEED0 + 700, 2, H, W, frames=2, max_disp=64, fx=300.0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[items])).to(dev)
    frames = [([T(x) for x in lf], [T(x) for x in rf], T(il), T(ir)) for lf, rf, il, ir in sc["frames"]]
    cur = frames[-1]
    frames[-1] = ([x.requires_grad_(True) for x in cur[0]], [x.requires_grad_(True) for x in cur[1]], cur[2], cur[3])
    eye = torch.eye(4, device=dev).expand(len(items), 4, 4).contiguous()
    return frames, T(sc["gt"][-1]), T(sc["K"]), [(eye, eye), (T(sc["T"][1]), eye)]


def build(dev):
    import bench
    import parity_tools as PT
    net = bench.build_model(dev, 5, 4)
    net.load_state_dict(PT.load_checkpoint(), strict=True)
    return net


def main(out):
    import torch.distributed as dist
    from temporalstereo_amd import dist as tsd
    from temporalstereo_amd.train import TrainStep
    rank, world, _ = tsd.init_distributed(os.environ.get("TS_BENCH_BACKEND", "nccl"))
    dev = torch.device("cuda", int(os.environ.get("TS_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    net = build(dev)
    mode = os.environ.get("TS_DDP_MODE", "collectives")      # collectives | peer | peer_graph
    if mode == "collectives":
        step = TrainStep(net, max_disp=64, local_map_size=1, bucket_bytes=1 << 20)        # several buckets: they go out during backward
        assert step.sync_bn and step.buckets is not None and step.peer is None
    else:
        # SyncBatchNorm statistics through the peer mailboxes (csrc/peer.hip); peer_graph: the whole step replayed from a hipGraph
        step = TrainStep(net, max_disp=64, local_map_size=1, bucket_bytes=1 << 20, sync_bn="peer", graph=(mode == "peer_graph"))
        assert step.sync_bn and step.peer is not None
    frames, gt, K, poses = scene(dev, [rank])
    losses = [float(step(frames, gt, K, poses))]
    first = {"g1::" + k: p.grad.detach().cpu().numpy() for k, p in step.net.named_parameters() if p.grad is not None}
    losses.append(float(step(frames, gt, K, poses)))
    torch.cuda.synchronize()
    if step.peer is not None:
        step.peer.check()                                    # no exchange timed out waiting for the other rank
    if rank == 0:
        from temporalstereo_amd import functional as TF
        np.savez(out, losses=np.array(losses), launched_in_backward=step.buckets.launched_in_backward if step.buckets is not None else -1,
                 exchanges=TF._EXCHANGES[0], **first,
                 **{"g::" + k: p.grad.detach().cpu().numpy() for k, p in step.net.named_parameters() if p.grad is not None},
                 **{"p::" + k: p.detach().cpu().numpy() for k, p in step.net.named_parameters()},
                 **{"b::" + k: b.detach().cpu().numpy() for k, b in step.net.named_buffers() if b.dtype.is_floating_point})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
