"""Worker of tests/test_peer_gpu.py (not collected): one rank of a peer-mailbox group (temporalstereo_amd/peer.py) on the GPU.
One rank per device over RCCL when the box has the devices, all ranks on device 0 with the handles over gloo otherwise
(TS_BENCH_BACKEND / TS_BENCH_DEVICE, set by tests/helpers.multi_rank_env)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out, mode):
    import torch.distributed as dist
    from temporalstereo_amd import dist as tsd
    from temporalstereo_amd.peer import PeerGroup
    rank, world, _ = tsd.init_distributed(os.environ.get("TS_BENCH_BACKEND", "nccl"))
    dev = torch.device("cuda", int(os.environ.get("TS_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    pg = PeerGroup()
    res = {}
    if mode == "missing_peer":
        # rank 1 never issues the exchange: rank 0's kernel must give up after its bound (no hang), and check() -- which agrees on the
        # outcome across the ranks -- must raise on BOTH
        if rank == 0:
            src = torch.ones(8, device=dev); dst = torch.empty(world, 8, device=dev)
            pg.all_gather(src, dst)
        polled = 0
        try:
            pg.poll()                                   # queues the copy of the err word behind the exchange
            torch.cuda.synchronize()
            pg.poll()                                   # ... and this one reads what it brought back
        except RuntimeError:
            polled = 1
        raised = 0
        try:
            pg.check()
        except RuntimeError:
            raised = 1
        # the group is dead until every rank resets it; afterwards an ordinary exchange works again
        pg.reset()
        src = torch.full((8,), float(rank + 1), device=dev); dst = torch.empty(world, 8, device=dev)
        pg.all_gather(src, dst)
        pg.check()
        ok = int(bool((dst.cpu() == torch.arange(1, world + 1, dtype=torch.float32).view(-1, 1)).all()))
        np.savez(out + ".rank%d.npz" % rank, raised=raised, polled=polled, after_reset_ok=ok)
        pg.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    g = torch.Generator().manual_seed(100 + rank)
    for it, n in enumerate([1, 17, 129, 257, 1024, 33, 64, 5, 300, 2]):       # more exchanges than slots: the slots wrap
        src = torch.randn(n, generator=g).to(dev)
        dst = torch.empty(world, n, device=dev)
        pg.all_gather(src, dst)
        buf = src.clone()
        scale = torch.full((1,), 0.5, device=dev)
        pg.all_reduce_sum(buf, scale if it % 2 else None)
        res["src%d" % it], res["gather%d" % it], res["sum%d" % it] = src.cpu().numpy(), dst.cpu().numpy(), buf.cpu().numpy()
    pg.check()
    if mode == "graph":
        # the same exchanges replayed from a hipGraph: the device-side sequence counter advances with every replay
        src = torch.zeros(65, device=dev); dst = torch.empty(world, 65, device=dev); buf = torch.zeros(65, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            pg.all_gather(src, dst); buf.copy_(src); pg.all_reduce_sum(buf)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            pg.all_gather(src, dst); buf.copy_(src); pg.all_reduce_sum(buf)
        for k in range(5):
            src.copy_(torch.full((65,), float(10 * k + rank), device=dev))
            gr.replay()
            torch.cuda.synchronize()
            res["gsrc%d" % k], res["ggather%d" % k], res["gsum%d" % k] = src.cpu().numpy(), dst.cpu().numpy(), buf.cpu().numpy()
        pg.check()
    np.savez(out + ".rank%d.npz" % rank, **res)
    pg.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "eager")
