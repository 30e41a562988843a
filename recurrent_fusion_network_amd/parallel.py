"""Synchronous data parallelism for the caption batch: one process per GPU, RCCL over xGMI.

The reference has no data parallelism (its "8 GPUs" are 8 independent seeds); this is the
north-star's sharded training.  Every batch row is independent through the whole path, so
the batch is split on dim 0, each rank runs the full path on its shard, and the ONLY exchange is one sum
all-reduce per flat gradient bucket (2M+2 buckets, in the order backward finishes them: decoder, fusion core,
then two per encoder).  The reference divides the loss by the LOCAL batch size, so the sum is scaled by
1/world_size inside the fused optimizer BEFORE the element-wise clamp -- which keeps an N-rank step equal
to the 1-rank step on the concatenated batch.
"""
import os
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, local)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('RFN_FORCE_DIST', '0') == '1'   # test hook: build a process group even for one rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        backend = os.environ.get('RFN_DIST_BACKEND', backend)   # test hook: gloo lets two ranks share one GPU
        if backend == 'nccl':
            local_dev = int(os.environ.get('RFN_DEVICE_INDEX', local))
            torch.cuda.set_device(local_dev)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rows(n_rows, rank, world):
    """Contiguous, near-equal row ranges; the shard of `rank` is [lo, hi)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_loss_scale(n_local, n_global, world):
    """Factor a rank multiplies its LOCAL loss with before backward when the shards are not all the same size.

    The criteria divide by the local batch size (misc/utils.py:184), so a rank's gradient is the mean over ITS rows; the
    exchange sums the ranks and the optimizer scales by 1/world.  For the result to equal the single-process gradient of the
    concatenated batch -- the mean over ALL rows -- rank r's loss has to weigh n_r / n_global instead of 1 / world:
    multiply it by n_r * world / n_global (exactly 1.0 for equal shards)."""
    return float(n_local) * float(world) / float(n_global)


def shard_pad(world):
    """Flat buckets of a sharded update are padded to a multiple of this many elements: `world` equal shards, each a whole
    number of 16-B vectors (the fused update moves float4s)."""
    return 4 * max(1, int(world))


def shard_bounds(total_padded, rank, world):
    """[lo, hi) of rank's shard of a flat bucket of `total_padded` elements (a multiple of shard_pad(world))."""
    if total_padded % shard_pad(world):
        raise ValueError('a bucket of %d elements cannot be cut into %d aligned shards' % (total_padded, world))
    sl = total_padded // max(1, world)
    return rank * sl, (rank + 1) * sl


def gather_shards(full, rank, world, group=None, async_op=False):
    """In-place all-gather of a flat buffer whose shard `rank` is up to date on this rank: afterwards every rank holds
    every shard.  nccl: one all_gather_into_tensor; other backends: the list form onto views of `full`."""
    lo, hi = shard_bounds(full.numel(), rank, world)
    if dist.get_backend(group) == 'nccl':
        return dist.all_gather_into_tensor(full, full[lo:hi], group=group, async_op=async_op)
    sl = hi - lo
    return dist.all_gather([full[r * sl:(r + 1) * sl] for r in range(world)], full[lo:hi].clone(), group=group, async_op=async_op)


def allreduce_flat(buffers, world, async_op=False):
    """Sum all-reduce of the flat gradient buffers, in the given order.  Returns work handles when async."""
    if world <= 1:
        return []
    works = []
    for b in buffers:
        w = dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            works.append(w)
    return works


class OverlappedUpdate:
    """One process, opt-in: the clamp + Adam of a bucket is launched on a side stream the moment backward has queued the
    kernels that finish the bucket's gradients (the model's `grad_ready_hook`: decoder first, then the fusion core, then the
    encoders) instead of in one launch after backward -- the update of bucket k streams its 28 bytes per parameter under the
    matrix-bound weight-gradient products of the buckets after it.  Nothing in the rest of backward reads a bucket's parameters
    once its gradients are complete, and the next forward waits for the side stream where it first hands parameters to a
    kernel (`param_wait_hook`).  Same element arithmetic as the single launch (FusedClampAdam.update_bucket_early): parameters
    and moments bit-identical.  `optimizer.step()` afterwards only counts the step (and updates what no hook delivered)."""

    def __init__(self, model, optimizer):
        if optimizer.sharded:
            raise ValueError('the sharded optimizer already pipelines its update (GradSync(shard_optimizer=...))')
        self.opt = optimizer
        self.side = torch.cuda.Stream()
        model.grad_ready_hook = self.on_bucket
        model.param_wait_hook = self.wait
        optimizer.add_joiner(self.wait)     # state_dict() / snapshot() / restore() read what the side stream may still write

    def on_bucket(self, name, flat):
        main = torch.cuda.current_stream(flat.device)
        self.side.wait_stream(main)
        flat.record_stream(self.side)
        with torch.cuda.stream(self.side):
            self.opt.update_bucket_early(name, flat, 1.0)

    def wait(self, which=None):
        torch.cuda.current_stream().wait_stream(self.side)

    def finish(self):
        return 1.0

    # the part of GradSync's interface bench.py drives (nothing is exchanged here)
    record = False

    def exposed_ms(self):
        return None, {}, None

    def abandon(self):
        pass


class GradSync:
    """Overlapped gradient exchange: registers as the model's `grad_ready_hook`, launches one asynchronous SUM
    all-reduce per gradient bucket the moment backward has queued the kernels that finish it (decoder first,
    then the fusion core, then one bucket per encoder), and `finish()` makes the compute stream wait for all of
    them.  Each collective starts behind the compute stream at its call point (process-group semantics), so
    bucket i's transfer runs under the weight-gradient GEMMs of the encoders after it."""

    def __init__(self, model, world, shard_optimizer=None):
        """shard_optimizer: a FusedClampAdam built with shard=(rank, world).  Each bucket is then REDUCE-SCATTERED (half the
        bytes of the all-reduce; the other half is the optimizer's all-gather of the updated parameters): rank r receives
        the sum of shard r into `shard_optimizer.reduced_shards[name]`, which is what its update reads.  `.grad` then holds
        the LOCAL gradient.  Backends without reduce-scatter (gloo) all-reduce instead -- same numbers."""
        self.world = world
        self.sharded = shard_optimizer
        if shard_optimizer is not None and shard_optimizer.shard_world != max(1, world):
            raise ValueError('the optimizer shards over %d ranks, the exchange runs over %d' % (shard_optimizer.shard_world, world))
        # sharded update, pipelined: bucket k's reduce-scatter, its 1/world update and the all-gather of its parameters run
        # on a side stream under the rest of backward (FusedClampAdam.update_bucket_early); the compute stream meets them
        # again where the next forward first reads parameters (model.param_wait_hook)
        self.side = None
        if shard_optimizer is not None and shard_optimizer.sharded and torch.cuda.is_available():
            self.side = torch.cuda.Stream()
        self._held, self._held_events = [], []   # gradient buffers the side stream may still read, and the events that say when not
        self._shard_bufs = {}            # bucket -> this rank's reduce-scatter output, allocated once (nothing in the step allocates)
        self.works = []
        self.buckets = []
        # exposed-wait bookkeeping (bench.py `exposed_ms`): off unless `record` is set.  On the nccl backend a wait() only
        # makes the compute stream wait for the collective's stream, so the time that stream really stalls is the distance
        # between two HIP events recorded around the wait -- read after the timed region, no synchronisation added.  Other
        # backends block the host inside wait(): the host clock around it is the exposed time.
        self.record = False
        self._pending = []               # per finish(): (bucket names, [event_0 .. event_n]) or (bucket names, [seconds])
        model.grad_ready_hook = self.on_bucket
        if world > 1 and hasattr(model, 'gemm_flags'):
            # A one-round weight-gradient GEMM holds its LDS on every CU for its whole 6 ms: keep the big tiles lean
            # (64 KB per CU instead of 128) while collectives run beside them, so RCCL's kernels can co-reside (rfn.h)
            from . import _native as N
            model.gemm_flags |= N.GEMM_OPT_LDS_LEAN

    def on_bucket(self, name, flat):
        self.buckets.append(name)
        if self.world > 1 or (dist.is_available() and dist.is_initialized()):
            opt = self.sharded
            if opt is not None and opt.sharded and flat.is_cuda:
                st = opt.flat[name]
                main = torch.cuda.current_stream(flat.device)
                if len(self.buckets) == 1 and self._held:
                    # first bucket of a step: the previous step's gradient buffers, kept alive for the side stream, go back
                    # to the allocator -- behind the events that mark the side stream's last read of them (long past: a whole
                    # forward lies in between).  Deterministic two-generation recycling instead of record_stream(), whose
                    # deferred frees made the allocator miss now and then (device_mallocs_frees_in_timed_region).
                    for ev in self._held_events:
                        main.wait_event(ev)
                    self._held.clear()
                    self._held_events.clear()
                self.side.wait_stream(main)               # the kernels that finish this bucket are queued on `main`
                with torch.cuda.stream(self.side):
                    if dist.get_backend(opt.group) == 'nccl':
                        shard = self._shard_bufs.get(name)       # reused every step: its reader (this bucket's update of the
                        if shard is None or shard.numel() != st['hi'] - st['lo'] or shard.device != flat.device:   # previous step)
                            shard = torch.empty(st['hi'] - st['lo'], device=flat.device, dtype=flat.dtype)        # is ahead on
                            self._shard_bufs[name] = shard                                                        # this stream
                        dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=opt.group)
                    else:                                 # no reduce-scatter on this backend (gloo): all-reduce, read the slice
                        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=opt.group)
                        shard = flat[st['lo']:st['hi']]
                    opt.update_bucket_early(name, shard, 1.0 / self.world)
                    ev = torch.cuda.Event()
                    ev.record(self.side)                  # the side stream's last read of `flat` (gloo: the shard is a view of it)
                self._held.append(flat)
                self._held_events.append(ev)
            else:
                self.works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        if self.record and self.works:
            on_stream = dist.get_backend() == 'nccl'
            marks = []
            for w in self.works:
                if on_stream:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    marks.append(e)
                    w.wait()
                else:
                    t0 = time.perf_counter()
                    w.wait()
                    marks.append(time.perf_counter() - t0)
            if on_stream:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
            self._pending.append((list(self.buckets), marks, on_stream))
        else:
            for w in self.works:
                w.wait()
        self.works.clear()
        self.buckets.clear()
        return 1.0 / self.world

    def abandon(self):
        """Forget queued collectives and measurements (a leg that raised half way through a step)."""
        self.works.clear()
        self.buckets.clear()
        self._pending.clear()

    def exposed_ms(self):
        """-> (mean ms per step the compute stream waited for gradient exchange, {bucket: mean ms}, how it was measured)
        over the steps finished since the last call.  Call after the device has been synchronised."""
        steps, total, per, how = len(self._pending), 0.0, {}, None
        for names, marks, on_stream in self._pending:
            how = 'hip events around each wait on the compute stream' if on_stream else 'host clock around each blocking wait'
            for k, name in enumerate(names):
                ms = marks[k].elapsed_time(marks[k + 1]) if on_stream else marks[k] * 1e3
                per[name] = per.get(name, 0.0) + ms
                total += ms
        self._pending.clear()
        if not steps:
            return None, {}, None
        return total / steps, {k: v / steps for k, v in per.items()}, how


def allreduce_model_grads(model, world):
    """All-reduce the gradients the last backward produced (decoder bucket first: it is ready first)."""
    flats = getattr(model, '_last_flat_grads', {})
    order = [flats[k] for k in model.bucket_names() if k in flats]
    allreduce_flat(order, world)
    return 1.0 / world


def max_over_ranks(value, world, device):
    """MAX all-reduce of a Python float (the bench timing rule)."""
    if world <= 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
