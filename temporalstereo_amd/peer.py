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
        backend = dist.get_backend(group)
        self._cdev = self.device if backend == "nccl" else torch.device("cpu")      # where this backend wants its tensors
        self._mine, self._opened, failure = None, [], None
        with torch.cuda.device(self.device):
            mine = ctypes.c_void_p()
            handle = (ctypes.c_ubyte * 64)()
            try:
                _lib.check(L.ts_peer_alloc(ctypes.byref(mine), handle), "ts_peer_alloc")
                self._mine = mine
            except RuntimeError as e:
                failure = str(e)
            # the handles through the process group (gloo moves CPU tensors, RCCL device tensors).  EVERY rank takes part in every
            # collective of the set-up whatever happened to it locally, and the outcome is agreed on at the end: a rank that raised
            # alone would leave the others waiting in the next collective.
            h = torch.tensor(list(bytes(handle)), dtype=torch.uint8, device=self._cdev)
            allh = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(allh, h, group=group)
            self.ctx = _Ctx()
            self.ctx.rank, self.ctx.world = self.rank, self.world
            for r in range(self.world):
                if r == self.rank:
                    self.ctx.region[r] = mine.value
                    continue
                if failure is not None:
                    continue
                raw = (ctypes.c_ubyte * 64)(*allh[r].cpu().tolist())
                p = ctypes.c_void_p()
                try:
                    _lib.check(L.ts_peer_open(raw, ctypes.byref(p)), "ts_peer_open")
                    self._opened.append(p)
                    self.ctx.region[r] = p.value
                except RuntimeError as e:
                    failure = "mailbox of rank %d: %s" % (r, e)
        self._ctxp = ctypes.cast(ctypes.pointer(self.ctx), ctypes.c_void_p)
        ok = torch.tensor([0.0 if failure is None else 1.0], device=self._cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MAX, group=group)      # also the barrier: nobody exchanges before everybody has mapped everybody
        if float(ok.item()) != 0.0:
            self._release()
            raise RuntimeError("PeerGroup: the mailboxes could not be mapped on every rank (%s)" % (failure or "another rank failed"))
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

    def check(self, collective=True):
        """Synchronises the current stream; raises when some exchange timed out waiting for a peer.  collective: the ranks agree on
        the outcome first (one all_reduce), so that either all of them raise or none does -- call it from every rank."""
        from .functional import _stream
        st = ctypes.c_int(0)
        _lib.check(_lib._real_lib().ts_peer_status(self._ctxp, ctypes.byref(st), _stream()), "ts_peer_status")
        bad = st.value
        if collective and dist.is_initialized():
            t = torch.tensor([float(bad)], device=self._cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            bad = int(t.item())
        if bad:
            raise RuntimeError("PeerGroup: rank %d did not answer an exchange within the bound (are all ranks issuing the same "
                               "sequence of exchanges?)" % (bad - 1))

    def _release(self):
        L = _lib._real_lib()
        for p in self._opened:
            L.ts_peer_close(p)
        self._opened = []
        if self._mine is not None:
            L.ts_peer_free(self._mine)
            self._mine = None

    def close(self):
        torch.cuda.synchronize(self.device)
        if dist.is_initialized():
            dist.barrier(group=self.group)    # nobody unmaps while a peer may still write
        self._release()


_INSTALLED = None


def install(pg):
    """Route the SyncBatchNorm exchanges of functional._ConvBNAct through `pg` (None: back to torch.distributed collectives)."""
    global _INSTALLED
    _INSTALLED = pg


def installed():
    return _INSTALLED
