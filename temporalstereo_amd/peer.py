"""Peer-mapped mailboxes of the ranks of one node (csrc/peer.hip): the set-up side.

    pg = PeerGroup()                      # after torch.distributed is initialised, one GPU per rank (or one shared GPU in tests)
    pg.all_gather(src, dst)               # dst [world, n] <- every rank's src [n]       (one kernel, capturable in a hipGraph)
    pg.all_reduce_sum(buf, scale=None)    # buf <- sum over ranks (rank order) * scale
    pg.check()                            # raises if a peer did not answer an exchange

The handles travel through torch.distributed (all_gather of 64 bytes per rank, any backend) ONCE; after that no communicator is
involved.  Used by functional._ConvBNAct for the SyncBatchNorm statistics when installed with `install()` (train.TrainStep does
that for sync_bn="peer"), reference: Lightning's sync_batchnorm=True, projects/TemporalStereo/dist_train.py:94.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


class _Ctx(ctypes.Structure):           # == ts_peer_ctx (include/ts_hip.h)
    _fields_ = [("region", ctypes.c_void_p * 8), ("rank", ctypes.c_int), ("world", ctypes.c_int)]


class PeerGroup:
    def __init__(self, group=None, device=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("PeerGroup needs an initialised torch.distributed process group")
        L = _lib._real_lib()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > int(L.ts_peer_max_ranks()):
            raise RuntimeError("PeerGroup: %d ranks, a node holds at most %d" % (self.world, int(L.ts_peer_max_ranks())))
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.max_floats = int(L.ts_peer_max_floats())
        with torch.cuda.device(self.device):
            mine = ctypes.c_void_p()
            handle = (ctypes.c_ubyte * 64)()
            _lib.check(L.ts_peer_alloc(ctypes.byref(mine), handle), "ts_peer_alloc")
            self._mine = mine
            # the handles through the process group (gloo moves CPU tensors, RCCL device tensors)
            backend = dist.get_backend(group)
            hdev = self.device if backend == "nccl" else torch.device("cpu")
            h = torch.tensor(list(bytes(handle)), dtype=torch.uint8, device=hdev)
            allh = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(allh, h, group=group)
            self._opened = []
            self.ctx = _Ctx()
            self.ctx.rank, self.ctx.world = self.rank, self.world
            for r in range(self.world):
                if r == self.rank:
                    self.ctx.region[r] = mine.value
                    continue
                raw = (ctypes.c_ubyte * 64)(*allh[r].cpu().tolist())
                p = ctypes.c_void_p()
                _lib.check(L.ts_peer_open(raw, ctypes.byref(p)), "ts_peer_open")
                self._opened.append(p)
                self.ctx.region[r] = p.value
        self._ctxp = ctypes.cast(ctypes.pointer(self.ctx), ctypes.c_void_p)
        dist.barrier(group=group)             # nobody exchanges before everybody has mapped everybody
        self.exchanges = 0                    # issued from this process (eager calls and captures; replays are not counted)

    def all_gather(self, src, dst):
        n = src.numel()
        if dst.numel() != n * self.world or src.dtype != torch.float32 or dst.dtype != torch.float32:
            raise ValueError("PeerGroup.all_gather: dst must hold world x src floats")
        from .functional import _stream
        _lib.check(_lib._real_lib().ts_peer_all_gather(self._ctxp, _lib.ptr(src), _lib.ptr(dst), n, _stream()), "ts_peer_all_gather")
        self.exchanges += 1
        return dst

    def all_reduce_sum(self, buf, scale=None):
        from .functional import _stream
        _lib.check(_lib._real_lib().ts_peer_all_reduce_sum(self._ctxp, _lib.ptr(buf), buf.numel(), _lib.ptr(scale), _stream()),
                   "ts_peer_all_reduce_sum")
        self.exchanges += 1
        return buf

    def check(self):
        """Synchronises the current stream; raises when some exchange timed out waiting for a peer."""
        from .functional import _stream
        st = ctypes.c_int(0)
        _lib.check(_lib._real_lib().ts_peer_status(self._ctxp, ctypes.byref(st), _stream()), "ts_peer_status")
        if st.value:
            raise RuntimeError("PeerGroup: rank %d did not answer an exchange within the bound (are all ranks issuing the same "
                               "sequence of exchanges?)" % (st.value - 1))

    def close(self):
        L = _lib._real_lib()
        torch.cuda.synchronize(self.device)
        if dist.is_initialized():
            dist.barrier(group=self.group)    # nobody unmaps while a peer may still write
        for p in self._opened:
            L.ts_peer_close(p)
        self._opened = []
        if self._mine is not None:
            L.ts_peer_free(self._mine)
            self._mine = None


_INSTALLED = None


def install(pg):
    """Route the SyncBatchNorm exchanges of functional._ConvBNAct through `pg` (None: back to torch.distributed collectives)."""
    global _INSTALLED
    _INSTALLED = pg


def installed():
    return _INSTALLED
