"""Fused clip_gradient + Adam (misc/utils.py:292-296 + train.py:69-71,162-163) on flat buffers.

The model keeps one flat gradient buffer per bucket (decoder, fusion core, two per encoder); ``FusedClampAdam``
re-points the parameters at flat buffers with the same layout, so the whole update -- clamp to +-grad_clip, L2
weight decay, Adam moments, bias correction -- is one rfn_adam_step launch per bucket (7 streams over 1.56 GB at
the headline config) instead of ~625 per-tensor updates, and a data-parallel run all-reduces 2M+2 buffers.

Trainer-facing surface of ``torch.optim.Adam`` that the reference's loops touch: ``zero_grad()``, ``step()``,
``param_groups[0]['lr']`` (``utils.set_lr``, misc/utils.py:286-290; train.py:102-104), ``state_dict()`` /
``load_state_dict()`` (``optimizer_<id>.pth``, train.py:86-88,232-233).

Opt-in under data parallelism: ``FusedClampAdam(model, ..., shard=(rank, world))`` shards the UPDATE -- the one part of the
step that does not shrink with the batch shard (10.9 GB of HBM traffic per step at the headline model whatever B is).  Every
flat bucket is cut into `world` equal 16-B aligned shards; a rank keeps Adam moments for its shard only, receives the summed
gradient of its shard (parallel.GradSync(shard_optimizer=True): reduce-scatter instead of all-reduce, half the wire bytes),
updates its shard of the parameters with the same element arithmetic, and the shards are all-gathered into every rank's full
parameter buffer -- asynchronously, in the order the next forward uses them, the model waiting (``param_wait_hook``) right
before the phase that reads them.  Same wire bytes as the all-reduce in total; element for element the same update arithmetic.  The
parameters are bit-identical to the unsharded step whenever the two exchanges sum in the same order: on gloo (an all-reduce
followed by a slice) and for 2 ranks (a + b commutes) -- which is what tests/test_parallel_gloo.py and
tests/test_parallel_gpu.py assert; RCCL's reduce-scatter and all-reduce at N > 2 may associate the N terms differently
(last-bit differences, not checked on hardware yet).
"""
import ctypes as C

import torch

from . import _native as N


def _dist_ready():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


class FusedClampAdam:
    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_clip=1.0, shard=None,
                 process_group=None):
        self.model = model
        self.shard_rank, self.shard_world = (int(shard[0]), int(shard[1])) if shard else (0, 1)
        self.group = process_group
        if not 0 <= self.shard_rank < self.shard_world:
            raise N.RfnError('shard = (rank, world) with 0 <= rank < world, got %r' % (shard,))
        # the sharded code path (reduce-scatter, shard update, all-gather, padded buckets) is taken whenever a shard was asked
        # for and there is a process group to run its collectives on -- also by a one-rank group (shard = (0, 1)), which is
        # how the RCCL branches are exercised on a single GPU (RFN_FORCE_DIST=1, tests/test_parallel_gpu.py)
        self.sharded = bool(shard) and (self.shard_world > 1 or _dist_ready())
        self._joiners = []              # callables that order the current stream after pending asynchronous updates
        if self.sharded:
            from . import parallel as DP
            model.flat_pad = DP.shard_pad(self.shard_world)       # before any flat buffer of this model is laid out
        # one group, torch.optim layout: utils.set_lr writes group['lr'], which step() reads
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, grad_clip=grad_clip,
                                  params=list(model.parameters()))]
        self.step_count = 0
        self.flat = {}
        if not next(model.parameters()).is_cuda:
            raise N.RfnError('move the model to the GPU before building FusedClampAdam')
        for name in model.bucket_names():
            params, offs, total = model.bucket_layout(name)     # same layout as the gradient buckets
            buf = torch.zeros(total, device=params[0].device, dtype=torch.float32)
            for o, p in zip(offs, params):
                buf[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = buf[o:o + p.numel()].view_as(p)
            lo, hi = 0, total                             # the whole bucket when the update is not sharded
            if self.sharded:
                from . import parallel as DP
                lo, hi = DP.shard_bounds(total, self.shard_rank, self.shard_world)
            self.flat[name] = dict(p=buf, m=torch.zeros(hi - lo, device=buf.device), v=torch.zeros(hi - lo, device=buf.device),
                                   n=total, lo=lo, hi=hi)
        self._gathers = {'prefix': [], 'decoder': []}
        self._updated_early = set()
        self.reduced_shards = {}        # bucket -> this rank's summed gradient shard (parallel.GradSync(shard_optimizer=True))
        if self.sharded:
            model.param_wait_hook = self.wait_params

    # ---- torch.optim-style accessors ---------------------------------------------------------------
    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, value):
        self.param_groups[0]['lr'] = value

    @property
    def betas(self):
        return self.param_groups[0]['betas']

    @property
    def eps(self):
        return self.param_groups[0]['eps']

    @property
    def weight_decay(self):
        return self.param_groups[0]['weight_decay']

    @property
    def grad_clip(self):
        return self.param_groups[0]['grad_clip']

    def set_lr(self, lr):
        self.param_groups[0]['lr'] = lr

    def zero_grad(self):
        for p in self.param_groups[0]['params']:      # the cached list: walking the module tree costs ~1 ms per call at 330 parameters
            p.grad = None
        self.model._last_flat_grads.clear()

    def _full_moment(self, st, key):
        """The whole bucket's moment (a collective when the update is sharded: every rank must call it)."""
        if not self.sharded:
            return st[key].detach().clone()
        import torch.distributed as dist
        parts = [torch.empty_like(st[key]) for _ in range(self.shard_world)]
        dist.all_gather(parts, st[key].contiguous(), group=self.group)
        return torch.cat(parts)

    def state_dict(self):
        """Adam moments per bucket (flat, same layout as the parameters), the step count and the hyper-parameters.  With a
        sharded update this gathers every rank's moment shards: call it on every rank."""
        self.wait_params()      # the moments may still be written by an overlapped / sharded update on another stream
        hyper = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        return {'step_count': self.step_count, 'hyper': hyper,
                'buckets': {name: {'m': self._full_moment(st, 'm'), 'v': self._full_moment(st, 'v'), 'n': st['n']}
                            for name, st in self.flat.items()}}

    def load_state_dict(self, sd):
        """Accepts this class's own format and the ``torch.optim.Adam`` format the reference's checkpoints hold
        (``optimizer_<id>.pth``, train.py:86-88: ``{'state': {index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups':
        [...]}`` with ``index`` = position in ``model.parameters()``, which is the reference's order: the module
        registers its parameters in the same sequence)."""
        if 'state' in sd and 'param_groups' in sd:
            return self._load_torch_adam(sd)
        if 'buckets' not in sd:
            raise N.RfnError('unrecognised optimizer state: expected FusedClampAdam\'s {buckets, step_count, hyper} or '
                             'torch.optim.Adam\'s {state, param_groups}, got keys %s' % sorted(sd))
        if set(sd['buckets']) != set(self.flat):
            raise N.RfnError('optimizer state has buckets %s, the model has %s' % (sorted(sd['buckets']), sorted(self.flat)))
        for name, st in self.flat.items():
            src = sd['buckets'][name]
            if int(src['n']) != st['n']:
                raise N.RfnError('optimizer bucket %s holds %d values, expected %d' % (name, int(src['n']), st['n']))
            st['m'].copy_(src['m'][st['lo']:st['hi']])
            st['v'].copy_(src['v'][st['lo']:st['hi']])
        self.step_count = int(sd['step_count'])
        for k, v in sd.get('hyper', {}).items():
            self.param_groups[0][k] = tuple(v) if k == 'betas' else v

    def _load_torch_adam(self, sd):
        groups = sd['param_groups']
        order = [i for g in groups for i in g['params']]
        params = list(self.model.parameters())
        if len(order) != len(params):
            raise N.RfnError('torch.optim.Adam state covers %d parameters, the model has %d' % (len(order), len(params)))
        where = {}      # id(parameter) -> (bucket, offset)
        for name in self.model.bucket_names():
            ps, offs, _ = self.model.bucket_layout(name)
            for p, o in zip(ps, offs):
                where[id(p)] = (name, o)
        # validate everything first: a rejected checkpoint must leave the moments and the step count as they were
        g0 = groups[0]
        if g0.get('amsgrad', False):
            raise N.RfnError('amsgrad state is not supported by the fused update')
        steps, todo = set(), []
        for pos, key in enumerate(order):
            ent = sd['state'].get(key)
            if ent is None:         # a parameter Adam never stepped: moments stay zero
                continue
            p = params[pos]
            for field in ('exp_avg', 'exp_avg_sq'):
                if tuple(ent[field].shape) != tuple(p.shape):
                    raise N.RfnError('optimizer state %s.%s has shape %s, parameter %d has %s'
                                     % (key, field, tuple(ent[field].shape), pos, tuple(p.shape)))
            steps.add(int(ent['step']))
            todo.append((p, ent))
        if len(steps) > 1:
            raise N.RfnError('torch.optim.Adam state has per-parameter step counts %s; the fused update keeps one' % sorted(steps))
        full = {name: (torch.zeros(st['n'], device=st['p'].device), torch.zeros(st['n'], device=st['p'].device))
                for name, st in self.flat.items()}
        for p, ent in todo:
            name, o = where[id(p)]
            n = p.numel()
            full[name][0][o:o + n].copy_(ent['exp_avg'].reshape(-1))
            full[name][1][o:o + n].copy_(ent['exp_avg_sq'].reshape(-1))
        for name, st in self.flat.items():
            st['m'].copy_(full[name][0][st['lo']:st['hi']])
            st['v'].copy_(full[name][1][st['lo']:st['hi']])
        self.step_count = steps.pop() if steps else 0
        for k in ('lr', 'eps', 'weight_decay'):
            if k in g0:
                self.param_groups[0][k] = g0[k]
        if 'betas' in g0:
            self.param_groups[0]['betas'] = tuple(g0['betas'])

    def snapshot(self):
        """Parameters, moments and step count as they are now (device copies): `restore` puts a run back at this point,
        so two legs of a benchmark can take the same number of updates from the same start."""
        self.wait_params()
        return {'step_count': self.step_count,
                'buckets': {name: tuple(st[k].detach().clone() for k in ('p', 'm', 'v')) for name, st in self.flat.items()}}

    def restore(self, snap):
        self.wait_params()
        for name, st in self.flat.items():
            for k, src in zip(('p', 'm', 'v'), snap['buckets'][name]):
                st[k].copy_(src)
        self.step_count = int(snap['step_count'])
        self.model._weights_epoch = getattr(self.model, '_weights_epoch', 0) + 1

    # ---- the update ------------------------------------------------------------------------------------
    def coefficients(self, step):
        """(lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)) as float32, computed exactly as rfn_adam_step_multi computes them
        (float arguments widened to double, the quotient rounded to float): what rfn_adam_step_multi_coef reads from device
        memory when the update is replayed from a captured graph (graphed.GraphedTrainStep)."""
        import numpy as np
        g0 = self.param_groups[0]
        lr, b1, b2 = (np.float64(np.float32(x)) for x in (g0['lr'], g0['betas'][0], g0['betas'][1]))
        bc1, bc2 = 1.0 - np.power(b1, np.float64(step)), 1.0 - np.power(b2, np.float64(step))
        return float(np.float32(lr / bc1)), float(np.float32(1.0 / np.sqrt(bc2)))

    def step(self, grad_scale=1.0, coef_dev=None):
        """grad_scale multiplies the gradient before the clamp (1/world_size after a sum all-reduce).
        coef_dev: a 2-float device tensor holding `coefficients(step_count + 1)` -- the launch then takes its step-dependent
        scalars from there instead of its kernel arguments (the graph-replayable form)."""
        g0 = self.param_groups[0]
        self.step_count += 1
        self.model._weights_epoch = getattr(self.model, '_weights_epoch', 0) + 1   # invalidates reuse_prefix entries
        todo, names = [], []
        early, self._updated_early = self._updated_early, set()
        for name, st in self.flat.items():
            g = self.model._last_flat_grads.get(name)
            if g is None or name in early:          # early: update_bucket_early() already ran this step's update of it
                continue
            if g.numel() != st['n']:
                raise N.RfnError('flat gradient layout changed')
            if self.sharded:
                # this rank's shard: the reduce-scattered gradient when GradSync delivered one, else the slice of the
                # (all-reduced or single-process) full buffer -- the same numbers either way
                gs = self.reduced_shards.pop(name, None)
                if gs is None:
                    gs = g[st['lo']:st['hi']]
                todo.append((dict(p=st['p'][st['lo']:st['hi']], m=st['m'], v=st['v'], n=st['hi'] - st['lo']), gs))
            else:
                todo.append((st, g))
            names.append(name)
        # every bucket in one launch (rfn_adam_step_multi: same element arithmetic, no per-bucket ramp and tail)
        for lo in range(0, len(todo), N.ADAM_MAXBUCKET):
            part = todo[lo:lo + N.ADAM_MAXBUCKET]
            sizes = (C.c_int64 * len(part))(*[st['n'] for st, _ in part])
            ptrs = (N.ptr_array([st['p'] for st, _ in part]), N.ptr_array([g for _, g in part]),
                    N.ptr_array([st['m'] for st, _ in part]), N.ptr_array([st['v'] for st, _ in part]))
            if coef_dev is not None:
                N.check(N.lib.rfn_adam_step_multi_coef(len(part), *ptrs, sizes, coef_dev.data_ptr(), g0['betas'][0],
                                                       g0['betas'][1], g0['eps'], g0['weight_decay'], g0['grad_clip'],
                                                       grad_scale, N.stream_ptr()), 'rfn_adam_step_multi_coef')
            else:
                N.check(N.lib.rfn_adam_step_multi(len(part), *ptrs, sizes, g0['lr'], g0['betas'][0], g0['betas'][1], g0['eps'],
                                                  g0['weight_decay'], g0['grad_clip'], grad_scale, self.step_count,
                                                  N.stream_ptr()), 'rfn_adam_step_multi')
        if self.sharded:
            self._gather_params(names)

    def update_bucket_early(self, name, grad_shard, grad_scale):
        """Sharded update of ONE bucket the moment its summed gradient shard is there -- from inside backward, on the caller's
        current (side) stream: clamp + Adam on this rank's shard with the scalars of the step that the coming step() will
        count, then the all-gather of the bucket's parameters.  Nothing in the rest of backward reads a bucket's parameters
        once its gradients are complete (backward hands the buckets over in that order), so the exchange of bucket k and its
        update run under the weight-gradient GEMMs of the buckets after it; step() then only counts the step."""
        g0, st = self.param_groups[0], self.flat[name]
        part = dict(p=st['p'][st['lo']:st['hi']], m=st['m'], v=st['v'], n=st['hi'] - st['lo'])
        sizes = (C.c_int64 * 1)(part['n'])
        N.check(N.lib.rfn_adam_step_multi(1, N.ptr_array([part['p']]), N.ptr_array([grad_shard]), N.ptr_array([part['m']]),
                                          N.ptr_array([part['v']]), sizes, g0['lr'], g0['betas'][0], g0['betas'][1], g0['eps'],
                                          g0['weight_decay'], g0['grad_clip'], grad_scale, self.step_count + 1, N.stream_ptr()),
                'rfn_adam_step_multi')
        self._updated_early.add(name)
        if self.sharded:
            self._gather_params([name])

    # ---- sharded update: parameters back to every rank ---------------------------------------------------------------
    def _gather_params(self, names):
        """All-gather the updated shards into every rank's full parameter buffers, asynchronously, in the order the next
        forward reads them: the stage-I / stage-II buckets first (rfn_prefix_fwd), the decoder bucket last -- it is not read
        before the whole prefix has run."""
        from . import parallel as DP
        order = [n for n in names if n != 'decoder'] + [n for n in names if n == 'decoder']
        for name in order:
            w = DP.gather_shards(self.flat[name]['p'], self.shard_rank, self.shard_world, self.group, async_op=True)
            self._gathers['decoder' if name == 'decoder' else 'prefix'].append(w)

    def wait_params(self, which=None):
        """The model's `param_wait_hook`: the compute stream (nccl) or the host (other backends) waits for the gathers of the
        buckets the phase `which` ('prefix', 'decoder'; None: all) is about to read."""
        for key in (('prefix', 'decoder') if which is None else (which,)):
            works = self._gathers[key]
            for w in works:
                w.wait()
            works.clear()
        if which is None:
            for join in self._joiners:
                join()

    def add_joiner(self, fn):
        """`fn()` orders the current stream after an asynchronous update of this optimizer's buffers that lives outside it
        (parallel.OverlappedUpdate's side stream): called by wait_params(), i.e. before state_dict / snapshot / restore read
        parameters or moments and by the model's state_dict pre-hook."""
        self._joiners.append(fn)
