"""Peer-mapped mailboxes of the ranks of one node (csrc/peer.hip): the set-up side.

    pg = PeerGroup()                      # after torch.distributed is initialised, one GPU per rank (or one shared GPU in tests)
    pg.all_gather(src, dst)               # dst [world, n] <- every rank's src [n]       (one kernel, capturable in a hipGraph)
    pg.all_reduce_sum(buf, scale=None)    # buf <- sum over ranks (rank order) * scale
    pg.check()                            # raises if a peer did not answer an exchange (synchronises; the ranks agree first)
    pg.poll()                             # the same question without synchronising: answers for the work queued before the PREVIOUS poll
    pg.reset()                            # collective: back to the state after construction (after a time-out the group is dead until then)
    set_timeout_ms(ms)                    # bound of one wait (default 120 s / TS_PEER_TIMEOUT_MS), process-wide

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
        self._status = torch.zeros(1, dtype=torch.int32).pin_memory()         # poll(): the err word lands here, one poll late
        self._status_ev = None

    def _check_args(self, what, *tensors):
        """float32, contiguous, on this group's device.  CONTRACT (not checkable here): the exchanges of a rank are ordered on the
        device -- one stream, or streams ordered by events as TrainStep's warm-up / capture / replay are; the device-side sequence
        counter is read and advanced by every exchange kernel, two concurrent ones would take the same slot."""
        from .functional import _stream
        for t in tensors:
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("PeerGroup.%s: tensors must be contiguous float32 on %s (got %s, %s, contiguous=%s)"
                                 % (what, self.device, t.dtype, t.device, t.is_contiguous()))
        return _stream()

    def all_gather(self, src, dst):
        n = src.numel()
        if dst.numel() != n * self.world:
            raise ValueError("PeerGroup.all_gather: dst must hold world x src floats")
        st = self._check_args("all_gather", src, dst)
        _lib.check(_lib._real_lib().ts_peer_all_gather(self._ctxp, _lib.ptr(src), _lib.ptr(dst), n, st), "ts_peer_all_gather")
        self.exchanges += 1
        return dst

    def all_reduce_sum(self, buf, scale=None):
        st = self._check_args("all_reduce_sum", buf, scale)
        _lib.check(_lib._real_lib().ts_peer_all_reduce_sum(self._ctxp, _lib.ptr(buf), buf.numel(), _lib.ptr(scale), st),
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

    def poll(self):
        """The err word WITHOUT synchronising: queues an asynchronous copy of it behind the work issued so far and returns what the
        copy queued by the PREVIOUS poll brought back (that one has normally long completed: train.TrainStep polls once per step, so
        a time-out is noticed one step late and training stops there).  Raises like check(collective=False)."""
        from .functional import _stream
        bad = 0
        if self._status_ev is not None:
            self._status_ev.synchronize()
            bad = int(self._status[0])
        with torch.cuda.device(self.device):
            _lib.check(_lib._real_lib().ts_peer_status_async(self._ctxp, ctypes.c_void_p(self._status.data_ptr()), _stream()), "ts_peer_status_async")
            self._status_ev = torch.cuda.Event()
            self._status_ev.record()
        if bad:
            raise RuntimeError("PeerGroup: rank %d did not answer an exchange within the bound; the statistics exchanged since then are "
                               "undefined -- stop, or PeerGroup.reset() on every rank and restore the last good state" % (bad - 1))

    def reset(self):
        """Collective: every rank, nothing in flight.  Flags, sequence counter and err word cleared -- the state after construction."""
        from .functional import _stream
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)                     # nobody clears while a peer may still write
        with torch.cuda.device(self.device):
            _lib.check(_lib._real_lib().ts_peer_reset(self._ctxp, _stream()), "ts_peer_reset")
        self._status_ev = None
        self._status.zero_()
        dist.barrier(group=self.group)                     # nobody exchanges before everybody has cleared

    def _release(self):
        L = _lib._real_lib()
        for p in self._opened:
            L.ts_peer_close(p)
        self._opened = []
        if self._mine is not None:
            L.ts_peer_free(self._mine)
            self._mine = None

    def close(self):
        if self._mine is None and not self._opened:
            return
        torch.cuda.synchronize(self.device)
        if dist.is_initialized():
            dist.barrier(group=self.group)    # nobody unmaps while a peer may still write
        self._release()
        self.closed = True
        if installed() is self:
            install(None)


def set_timeout_ms(ms):
    """Bound of one wait of an exchange kernel in milliseconds (process-wide; a captured graph keeps the bound it was captured with).
    Returns the previous bound.  Default 120 000 (TS_PEER_TIMEOUT_MS overrides): ranks of a real job drift apart by seconds."""
    return int(_lib._real_lib().ts_peer_set_timeout_ms(int(ms)))


_INSTALLED = None


def install(pg):
    """Route the SyncBatchNorm exchanges of functional._ConvBNAct through `pg` (None: back to torch.distributed collectives).
    Only exchanges over pg's own process group take it (`for_group`)."""
    global _INSTALLED
    _INSTALLED = pg


def installed():
    return _INSTALLED


def for_group(group, n_floats):
    """The installed PeerGroup if it spans exactly `group` (None == the default group) and can carry `n_floats` per exchange; else None
    (the caller uses torch.distributed)."""
    pg = _INSTALLED
    if pg is None or n_floats > pg.max_floats:
        return None
    a = pg.group if pg.group is not None else dist.group.WORLD
    b = group if group is not None else dist.group.WORLD
    return pg if a is b else None
