"""One-process-per-GPU data parallelism over RCCL (torch.distributed backend "nccl" on ROCm).

The reference has no multi-device code; this is the north-star's data-parallel
extension of `Model.fit`: tiles are independent units, so
  * training shards the minibatch over ranks and has ONE exchange per step: an all-reduce (sum)
    of the flat fp32 gradient buffer (bucketed, started while the encoder's backward still runs),
    averaged by the optimizer's grad_scale = 1/world.
    BatchNorm statistics stay per-replica by default (what tf.distribute does with plain
    BatchNormalization); `model.sync_bn = True` (or SATCV_SYNC_BN=1) averages the per-channel
    [Σx, Σx²] (forward) and [Σdy, Σdy·x̂] (backward) buffers over replicas, which makes N replicas of
    batch B/N arithmetically one device with batch B.
  * inference shards the chip list; no collective on the data path (templates are disjoint and
    are summed once at the end).
The helpers work on any device so the N>1 logic is covered by gloo tests on CPU.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


# test switch: run every collective at world size 1 as well (RCCL accepts a one-rank communicator), so that the nccl code
# path -- communicator init, ReduceOp.AVG, the async bucketed all-reduce on the side stream, the stream waits -- can execute on
# a single-GPU box (tests/test_dp_gpu.py::test_rccl_single_rank_path)
FORCE = os.environ.get('SATCV_FORCE_COLLECTIVES', '0') == '1'


# ---- the C-ABI communicator (include/satcv.h: satcv_comm_*): with the RCCL backend the gradient exchange and the SyncBN means CAN go
# through libsatcv's own ncclAllReduce wrapper -- what a caller binding the C ABI from the reference side would use -- on a
# communicator bootstrapped over the default process group.  torch.distributed stays the rendezvous (and the whole path for gloo,
# i.e. the CPU tests).  OPT-IN (SATCV_CABI_COMM=1) until a world >= 2 run on real hardware has been recorded: the path has only ever
# executed on a one-rank communicator (tests/test_dp_gpu.py), the default exchange is torch.distributed's ProcessGroupNCCL.
CABI_COMM = os.environ.get('SATCV_CABI_COMM', '0') == '1'
_comm = {'handle': None, 'tried': False, 'calls': 0}


def _all_ok(flag):
    """collective AND of a per-rank success flag over the default group (every rank gets the same answer)"""
    if dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def cabi_comm():
    """opaque satcv_comm* (an int) for the default group on RCCL, created collectively on first use; None otherwise.

    Every step is agreed on by ALL ranks before the next one starts, so that no rank enters ncclCommInitRank while a peer has
    already given up (it would block there forever): (1) every rank proves it can bind RCCL through the library (a throw-away
    unique id), all-reduce(MIN) of the flags; (2) rank 0's id is broadcast; (3) satcv_comm_init, all-reduce(MIN) of the results --
    if any rank failed, the ranks that succeeded destroy their communicator and EVERY rank stays on torch.distributed."""
    if _comm['tried'] or not (CABI_COMM and dist.is_initialized() and dist.get_backend() == 'nccl' and torch.cuda.is_available()):
        return _comm['handle']
    import ctypes as C
    import warnings
    from ._lib import lib
    _comm['tried'] = True
    rank, world = dist.get_rank(), dist.get_world_size()
    ident = (C.c_ubyte * 128)()
    can_bind = lib.satcv_comm_unique_id(ident) == 0          # (binds RCCL in this process; only rank 0's id is used)
    if not _all_ok(can_bind):
        warnings.warn('satcv C-ABI communicator: RCCL could not be bound on every rank (' + lib.satcv_last_error().decode() +
                      '); the exchange stays on torch.distributed')
        return None
    box = [bytes(ident)]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    buf = (C.c_ubyte * 128).from_buffer_copy(box[0])
    h = C.c_void_p()
    ok = lib.satcv_comm_init(C.byref(h), rank, world, buf) == 0
    err = '' if ok else lib.satcv_last_error().decode()
    if not _all_ok(ok):
        if ok:
            lib.satcv_comm_destroy(h.value)
        warnings.warn(f'satcv C-ABI communicator: satcv_comm_init failed on a rank ({err or "a peer"}); the exchange stays on torch.distributed')
        return None
    _comm['handle'] = h.value
    return _comm['handle']


def destroy_cabi_comm():
    if _comm['handle'] is not None:
        from ._lib import lib
        lib.satcv_comm_destroy(_comm['handle'])
    _comm['handle'], _comm['tried'] = None, False


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def active(group=None):
    """True when the collectives have to run: an initialised process group of more than one rank (or the test switch)."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE)


def allreduce_mean_(t, group=None):
    """In-place mean over ranks (RCCL has a native AVG; gloo sums then scales)."""
    if not active(group):
        return t
    comm = cabi_comm() if (group is None and t.is_cuda and t.is_contiguous() and t.dtype in _CABI_DT) else None
    if comm is not None:
        from ._lib import lib, check
        check(lib.satcv_allreduce(comm, t.data_ptr(), t.numel(), _CABI_DT[t.dtype], 1, torch.cuda.current_stream().cuda_stream))
        _comm['calls'] += 1
    elif dist.get_backend(group) == 'nccl':
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t


_CABI_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 4}


class GradSync:
    """All-reduce of a flat gradient buffer in buckets, overlapped with the rest of the backward pass.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU); the 74 MB fp32 gradient is sent as a few large buckets (default
    16 MiB) rather than per-layer tensors so that each collective is bandwidth- not latency-bound.  Buckets are cut from the
    END of the buffer: backward finishes the layers in decreasing offset order (head, dec0 .. dec4, center, enc4 .. enc0), so
    `ready_above(flat, lo, stream)` -- called by the engine after every layer -- can start the collectives of all buckets that
    lie above the highest still-pending offset while the encoder's backward is still running; `__call__` sends what is left
    (the first ~10 MB) and makes the current stream wait for everything."""

    def __init__(self, numel, bucket_bytes=16 << 20, group=None, overlap=True, payload=None):
        self.group, self.overlap = group, overlap
        per = max(bucket_bytes // 4, 1)
        self.per = per
        # wire format of the gradient: fp32 (74.1 MB for get_unet_model(2, 4)) or bf16 (37 MB: rounded, summed in bf16, widened
        # back; C-ABI communicator only).  SATCV_GRAD_PAYLOAD=bf16 selects it for every GradSync of the process.
        self.payload = payload or os.environ.get('SATCV_GRAD_PAYLOAD', 'fp32')
        self._warned_payload = False
        self._scratch = None
        self._side_ev = None
        self.bounds = []
        hi = numel
        while hi > 0:
            lo = max(hi - per, 0)
            self.bounds.append((lo, hi))
            hi = lo
        self.bounds.reverse()
        self._next = len(self.bounds) - 1          # highest bucket not yet launched
        self._works = []

    def _active(self):
        return active(self.group)

    def ready_above(self, flat, lo, stream=None):
        """every gradient element at offset >= lo is final once the work queued so far on the current stream and on `stream`
        (the weight-gradient stream, or None) has run: start the collectives of the buckets inside [lo, numel)."""
        if not (self.overlap and self._active() and flat.is_cuda):
            return
        if self._next < 0 or self.bounds[self._next][0] < lo:
            return
        cur = torch.cuda.current_stream()
        run_on = stream if stream is not None else cur
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            stream.wait_event(ev)
        comm = self._comm(flat)
        if comm is not None:
            # every whole bucket above lo as ONE range: satcv_allreduce_grads cuts it from the end into the same buckets.  The
            # collective runs on a stream of its OWN behind an event of `run_on` (as ProcessGroupNCCL's internal stream does for the
            # torch path): on the weight-gradient stream itself it would hold back every later weight gradient until the peers have
            # answered
            hi_b = self.bounds[self._next][1]
            while self._next >= 0 and self.bounds[self._next][0] >= lo:
                lo_b = self.bounds[self._next][0]
                self._next -= 1
            xs = self._exchange_stream(flat.device)
            ev = torch.cuda.Event()
            ev.record(run_on)
            xs.wait_event(ev)
            self._send(comm, flat, lo_b, hi_b, xs)
            self._side_ev = torch.cuda.Event()
            self._side_ev.record(xs)
            return
        with torch.cuda.stream(run_on):
            while self._next >= 0 and self.bounds[self._next][0] >= lo:
                a, b = self.bounds[self._next]
                self._works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self._next -= 1

    def _comm(self, flat):
        comm = cabi_comm() if (self.group is None and flat.is_cuda and flat.dtype == torch.float32) else None
        if comm is None and self.payload == 'bf16' and not self._warned_payload:
            import warnings
            self._warned_payload = True          # (only satcv_allreduce_grads implements the 37 MB wire format)
            warnings.warn("GradSync(payload='bf16') needs the C-ABI communicator (SATCV_CABI_COMM=1, RCCL backend, default group): "
                          'the exchange runs over torch.distributed in fp32')
        return comm

    def _exchange_stream(self, device):
        if getattr(self, '_xs', None) is None:
            self._xs = torch.cuda.Stream(device=device)
        return self._xs

    def _send(self, comm, flat, a, b, stream):
        from ._lib import lib, check
        scratch = None
        if self.payload == 'bf16':
            if self._scratch is None or self._scratch.numel() < flat.numel():
                self._scratch = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            scratch = self._scratch.data_ptr() + 2 * a
        check(lib.satcv_allreduce_grads(comm, flat.data_ptr(), a, b, self.per, 1 if scratch else 0, scratch, stream.cuda_stream))
        _comm['calls'] += 1

    def __call__(self, flat):
        if self._active():
            comm = self._comm(flat)
            if comm is not None:
                # (every collective of the communicator on the ONE exchange stream; the current stream -- Adam comes next -- waits for it)
                cur = torch.cuda.current_stream()
                xs = self._exchange_stream(flat.device)
                if self._next >= 0:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    xs.wait_event(ev)
                    self._send(comm, flat, 0, self.bounds[self._next][1], xs)
                    self._next = -1
                    self._side_ev = torch.cuda.Event()
                    self._side_ev.record(xs)
                if self._side_ev is not None:
                    cur.wait_event(self._side_ev)
                    self._side_ev = None
            while self._next >= 0:
                a, b = self.bounds[self._next]
                self._works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self._next -= 1
            for w in self._works:
                w.wait()                           # the CURRENT stream waits for the collective
        self._works = []
        self._next = len(self.bounds) - 1


def broadcast_state(tensors, src=0, group=None):
    if active(group):
        for t in tensors:
            dist.broadcast(t, src, group=group)


def make_grad_sync(model, bucket_bytes=16 << 20, overlap=None):
    """Replicate rank 0's weights, set the optimizer's gradient scale to 1/world and return the
    callable that `Model.train_step_device` runs between backward and Adam."""
    rt = model.runtime
    world = dist.get_world_size() if dist.is_initialized() else 1
    broadcast_state([rt.pflat, rt.sflat])
    rt.repack()
    model._weights_version = getattr(model, '_weights_version', 0) + 1
    rt.adam_state[2:3].fill_(1.0 / world)
    if overlap is None:
        overlap = os.environ.get('SATCV_OVERLAP_ALLREDUCE', '1') != '0'
    sync = GradSync(rt.gflat.numel(), bucket_bytes, overlap=overlap)
    model._sync_grads = sync
    return sync


def shard_list(items, rank, world):
    """Round-robin partition of independent units (chips / tiles) over ranks."""
    return list(items)[rank::world]


def reduce_templates(template, group=None):
    """Sum the per-rank stitched outputs (disjoint writes elsewhere zero) on every rank."""
    if active(group):
        dist.all_reduce(template, op=dist.ReduceOp.SUM, group=group)
    return template
