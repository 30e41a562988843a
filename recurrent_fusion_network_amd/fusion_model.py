"""RecurrentFusionModel: host-side mirror of the reference's module for the HIP path.

Same constructor argument (an ``opt`` Namespace), same ``forward(fc_feats, att_feats, seq)`` /
``sample(fc_feats, att_feats, opt)`` / ``sample_beam`` / ``get_init_state`` / ``get_thought_vectors`` /
``one_time_step`` surface and the same ``state_dict`` keys and shapes as the reference's
``misc/RecurrentFusionModel.py:117-658`` -- so ``models.setup(opt)``, ``train.py`` and ``eval.py`` drop in.
All arithmetic runs in librfn_hip.so (include/rfn.h); PyTorch only owns device memory, the stream and
the autograd graph edges between the two phases.  There is no CPU path: inputs must be on the GPU.

The sub-modules below (``_AttParams`` ...) hold parameters only; they exist to reproduce the reference's
parameter names (``review_steps_individual.{t}.lstm.{i}.att_model.att_2_att_h.weight`` ...).
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import re
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _native as N

_INIT = 0.1


def _uniform(*tensors):
    for t in tensors:
        t.data.uniform_(-_INIT, _INIT)


class _AttParams(nn.Module):
    """Parameters of AttentionModelCore (misc/AttentionModelCore.py:16-29)."""

    def __init__(self, rnn_size, feat_size, att_hid):
        super().__init__()
        self.att_2_att_h = nn.Linear(feat_size, att_hid)
        self.h_2_att_h = nn.Linear(rnn_size, att_hid)
        self.att_h_2_out = nn.Linear(att_hid, 1)
        for lin in (self.att_2_att_h, self.h_2_att_h, self.att_h_2_out):
            _uniform(lin.weight, lin.bias)


class _FusionCellParams(nn.Module):
    """Parameters of LSTMFusionNoInputCore (misc/RecurrentFusionModel.py:30-45); biases keep nn.Linear's
    default init as in the reference."""

    def __init__(self, H_size, rnn_size, feat_size, att_hid):
        super().__init__()
        self.att_model = _AttParams(rnn_size, feat_size, att_hid)
        self.H2h = nn.Linear(H_size, 4 * rnn_size)
        self.z2h = nn.Linear(feat_size, 4 * rnn_size)
        _uniform(self.H2h.weight, self.z2h.weight)


class _FusionStepParams(nn.Module):
    """FeatArrayFusionNoInputCore (misc/RecurrentFusionModel.py:93-97)."""

    def __init__(self, M, rnn_size, feat_sizes, att_hid):
        super().__init__()
        self.lstm = nn.ModuleList([_FusionCellParams(M * rnn_size, rnn_size, feat_sizes[i], att_hid) for i in range(M)])


class _ReviewStepParams(nn.Module):
    """LSTMSoftMultiAttentionFeatArrayNoInputCore (misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:24-38)."""

    def __init__(self, M, rnn_size, att_hid, maxout=0):
        super().__init__()
        gw = (5 if maxout else 4) * rnn_size
        self.h2h = nn.Linear(rnn_size, gw)
        self.z_2_h = nn.ModuleList([nn.Linear(rnn_size, gw) for _ in range(M)])
        self.att_model = nn.ModuleList([_AttParams(rnn_size, rnn_size, att_hid) for _ in range(M)])
        _uniform(self.h2h.weight, self.h2h.bias)


class _DecoderParams(nn.Module):
    """LSTMSoftAttentionCore (misc/LSTMSoftAttentionCore.py:24-58)."""

    def __init__(self, enc_size, rnn_size, att_hid, maxout=0):
        super().__init__()
        gw = (5 if maxout else 4) * rnn_size
        self.i2h = nn.Linear(enc_size, gw)
        self.h2h = nn.Linear(rnn_size, gw)
        self.z2h = nn.Linear(rnn_size, gw)
        self.att_2_att_h = nn.Linear(rnn_size, att_hid)
        self.h_2_att_h = nn.Linear(rnn_size, att_hid)
        self.att_h_2_out = nn.Linear(att_hid, 1)
        for lin in (self.i2h, self.h2h, self.z2h, self.att_2_att_h, self.h_2_att_h, self.att_h_2_out):
            _uniform(lin.weight, lin.bias)


def _fresh_seed() -> int:
    # dropout masks are Philox streams keyed by this seed; drawn from torch's CPU generator so that
    # torch.manual_seed(seed + rank) (train.py:23) makes runs reproducible
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _PrefixFn(torch.autograd.Function):
    """Phase 1 (fc2h + fusion stages I and II) as one autograd node: rfn_prefix_fwd / rfn_prefix_bwd."""

    @staticmethod
    def forward(ctx, model, save_bwd, drop, seed, M, *tensors):
        fc = [N.require_cuda_f32(t, 'fc_feats') for t in tensors[:M]]
        att = [N.require_cuda_f32(t, 'att_feats') for t in tensors[M:2 * M]]
        params = tensors[2 * M:]
        d = model._dims_for(drop)
        train = bool(save_bwd)
        B = fc[0].shape[0]
        dev = fc[0].device
        R, T2, K = d.R, d.T2, d.K
        table = model._param_table(params, model._prefix_slots)
        ws_bytes = N.lib.rfn_prefix_ws_bytes(C.byref(d), B, int(train))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        comb = torch.empty(T2, B, R, device=dev)
        h = torch.empty(B, R, device=dev)
        c = torch.empty(B, R, device=dev)
        reason = torch.empty(M + 1, B, K, device=dev)
        N.check(N.lib.rfn_prefix_fwd(C.byref(d), B, table, N.ptr_array(fc), N.ptr_array(att), comb.data_ptr(),
                                     h.data_ptr(), c.data_ptr(), reason.data_ptr(), ws.data_ptr(), ws_bytes,
                                     int(train), seed, N.stream_ptr()), 'rfn_prefix_fwd')
        if train:
            ctx.model, ctx.seed, ctx.M, ctx.B, ctx.drop = model, seed, M, B, drop
            ctx.fc, ctx.att, ctx.params, ctx.consumed, ctx.snapshot = fc, att, params, False, None
            # the multi-GB workspace is a SAVED tensor, not an attribute: autograd frees it right after a non-retained
            # backward instead of keeping it for as long as anything references the graph (a trainer's `loss` variable)
            ctx.save_for_backward(ws)
        return comb, h, c, reason

    @staticmethod
    def backward(ctx, d_comb, d_h, d_c, d_reason):
        model, M, B = ctx.model, ctx.M, ctx.B
        ws, = ctx.saved_tensors
        d = model._dims_for(ctx.drop)
        dev = ctx.fc[0].device
        table = model._param_table(ctx.params, model._prefix_slots)
        flats, by_slot, gtable = model._grad_buffers(model._prefix_buckets, dev)
        cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_comb, d_h, d_c, d_reason = cont(d_comb), cont(d_h), cont(d_c), cont(d_reason)
        ws_bytes = ws.numel()
        att_ptrs = N.ptr_array(ctx.att)
        if ctx.consumed and ctx.snapshot is not None:
            # model.retain_activations: the activations of the ORIGINAL forward (see below).  Through .data: `ws` is a
            # saved tensor, and a versioned in-place write would make autograd refuse the NEXT backward over this graph
            # (ppo_k = 10 passes in the reference's loop, train_rl.py:190-201)
            ws.data.copy_(ctx.snapshot)
        elif ctx.consumed:
            # backward overwrites activations in place (projections -> their gradients, gates -> gate gradients), so a
            # second backward over the same graph (loss.backward(retain_graph=True) in the PPO loop, train_rl.py:190-201)
            # first recomputes phase 1 into the workspace: same inputs, same dropout seed, the weights AS THEY ARE NOW.
            # If optimizer.step() ran in between (it does in that loop) these are the activations of the updated weights:
            # a self-consistent gradient at the new weights.  The reference instead backpropagates the activations saved
            # at the original forward through the updated weights; set model.retain_activations = True for exactly that
            # (costs one copy of the workspace per retained graph).
            R, T2, K = d.R, d.T2, d.K
            scratch = torch.empty(T2 * B * R + 2 * B * R + (M + 1) * B * K, device=dev)
            o1, o2, o3 = T2 * B * R, T2 * B * R + B * R, T2 * B * R + 2 * B * R
            N.check(N.lib.rfn_prefix_fwd(C.byref(d), B, table, N.ptr_array(ctx.fc), att_ptrs, scratch.data_ptr(),
                                         scratch[o1:].data_ptr(), scratch[o2:].data_ptr(), scratch[o3:].data_ptr(),
                                         ws.data_ptr(), ws_bytes, 1, ctx.seed, N.stream_ptr()),
                    'rfn_prefix_fwd (recompute)')
        elif getattr(model, 'retain_activations', False):
            ctx.snapshot = ws.clone()
        ctx.consumed = True
        # everything except the per-encoder stage-I weight gradients ...
        N.check(N.lib.rfn_prefix_bwd(C.byref(d), B, table, N.ptr_array(ctx.fc), att_ptrs, N.ptr(d_comb),
                                     N.ptr(d_h), N.ptr(d_c), N.ptr(d_reason), gtable, ws.data_ptr(), ws_bytes,
                                     ctx.seed, 1, N.stream_ptr()), 'rfn_prefix_bwd')
        model._bucket_done('core', flats['core'])
        # ... then the per-encoder stage-I weight gradients, the BIG buckets first: part a of every encoder (H2h, z2h,
        # h_2_att_h: 277 MB each at C3, short K = B GEMMs) is produced and handed over before the first long att_2_att_h
        # product (part b: 34 MB, the largest GEMMs of backward) starts, so a data-parallel host has 71 % of the gradient
        # bytes in flight under ALL of those GEMMs instead of encoder i's bucket under encoder i's GEMM only -- at a
        # 32-caption shard the b-products are 3.3 ms against 5+ ms of exchange
        todo = [(i, part, tag) for part, tag in ((1, 'a'), (2, 'b')) for i in range(M)]
        if getattr(model, '_wgrad_interleaved', False):      # tools A/B only: encoder by encoder (a0 b0 a1 b1 ...)
            todo = [(i, part, tag) for i in range(M) for part, tag in ((1, 'a'), (2, 'b'))]
        for i, part, tag in todo:
            N.check(N.lib.rfn_prefix_bwd_wgrad(C.byref(d), B, att_ptrs, gtable, ws.data_ptr(), ws_bytes, i,
                                               part, N.stream_ptr()), 'rfn_prefix_bwd_wgrad')
            model._bucket_done('enc%d%s' % (i, tag), flats['enc%d%s' % (i, tag)])
        return (None, None, None, None, None) + (None,) * (2 * M) + (None,) * len(ctx.params)


class _DecoderFn(torch.autograd.Function):
    """Phase 2 (decoder + logit + log-softmax): rfn_decoder_fwd / rfn_decoder_bwd.

    ss_prob > 0 (scheduled sampling, misc/RecurrentFusionModel.py:260-270): the pass runs one step at a time
    (rfn_decoder_fwd_begin / rfn_decoder_fwd_step) and, between steps, replaces a row's next input token with
    probability ss_prob by a draw from the distribution the step just produced.  The step-wise pass leaves the
    workspace exactly as the batched pass on the final ids would (bit for bit), so backward is the same call: the pass
    that is sampled IS the pass that is differentiated -- dropout masks included -- and nothing is computed twice.  No
    host read-back: the draw is made for every row and kept where the row's coin says so."""

    @staticmethod
    def forward(ctx, model, save_bwd, drop, seed, ids, comb, h0, c0, ss_prob, inv_temp, *params):
        d = model._dims_for(drop)
        train = bool(save_bwd)
        B, S = ids.shape
        dev = comb.device
        comb, h0, c0 = comb.contiguous(), h0.contiguous(), c0.contiguous()
        ids = ids.contiguous()
        table = model._param_table(params, model._decoder_slots)
        ws_bytes = N.lib.rfn_decoder_ws_bytes(C.byref(d), B, S, int(train))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        log_prob = torch.empty(B, S, d.V1, device=dev)
        if ss_prob > 0.0 and S > 1:
            ids = ids.clone()
            # one call queues begin + S x (draw, step): the host is off the critical path (uniforms drawn up front)
            r = torch.rand(2, S, B, device=dev)
            N.check(N.lib.rfn_decoder_fwd_sampled(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                                  ids.data_ptr(), ids.stride(0), float(ss_prob), float(inv_temp),
                                                  r[0].data_ptr(), r[1].data_ptr(), log_prob.data_ptr(), ws.data_ptr(),
                                                  ws_bytes, int(train), seed, N.stream_ptr()), 'rfn_decoder_fwd_sampled')
            if getattr(model, '_trace_ss', False):
                model._ss_ids = ids.clone()      # test hook: the token matrix the pass ended up feeding
        else:
            N.check(N.lib.rfn_decoder_fwd(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                          ids.data_ptr(), ids.stride(0), log_prob.data_ptr(), ws.data_ptr(), ws_bytes,
                                          int(train), seed, N.stream_ptr()), 'rfn_decoder_fwd')
        if train:
            ctx.model, ctx.seed, ctx.B, ctx.S, ctx.drop = model, seed, B, S, drop
            ctx.ids, ctx.params, ctx.consumed, ctx.snapshot = ids, params, False, None
            ctx.save_for_backward(comb, h0, c0, log_prob, ws)     # ws: freed with the graph (see _PrefixFn.forward)
        ctx.mark_non_differentiable(ids)
        return log_prob, ids

    @staticmethod
    def backward(ctx, d_log_prob, _d_ids=None):
        model, B, S = ctx.model, ctx.B, ctx.S
        comb, h0, c0, log_prob, ws = ctx.saved_tensors
        d = model._dims_for(ctx.drop)
        dev = comb.device
        table = model._param_table(ctx.params, model._decoder_slots)
        flats, by_slot, gtable = model._grad_buffers(['decoder'], dev)
        d_log_prob = d_log_prob.contiguous()
        d_comb = torch.empty_like(comb)
        d_h0 = torch.empty_like(h0)
        d_c0 = torch.empty_like(c0)
        if ctx.consumed and ctx.snapshot is not None:
            ws.data.copy_(ctx.snapshot)          # .data: no version bump on a saved tensor (see _PrefixFn.backward)
        elif ctx.consumed:
            # second backward over the same graph: recompute phase 2 first (see _PrefixFn.backward), log-probs included,
            # so the pass differentiated is self-consistent at the current weights
            log_prob = torch.empty_like(log_prob)
            N.check(N.lib.rfn_decoder_fwd(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                          ctx.ids.data_ptr(), ctx.ids.stride(0), log_prob.data_ptr(),
                                          ws.data_ptr(), ws.numel(), 1, ctx.seed, N.stream_ptr()),
                    'rfn_decoder_fwd (recompute)')
        elif getattr(model, 'retain_activations', False):
            ctx.snapshot = ws.clone()
        ctx.consumed = True
        N.check(N.lib.rfn_decoder_bwd(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                      ctx.ids.data_ptr(), ctx.ids.stride(0), log_prob.data_ptr(),
                                      d_log_prob.data_ptr(), d_comb.data_ptr(), d_h0.data_ptr(), d_c0.data_ptr(),
                                      gtable, ws.data_ptr(), ws.numel(), ctx.seed, N.stream_ptr()),
                'rfn_decoder_bwd')
        model._bucket_done('decoder', flats['decoder'])
        return (None, None, None, None, None, d_comb, d_h0, d_c0, None, None) + (None,) * len(ctx.params)


class _DecoderLossFn(torch.autograd.Function):
    """Phase 2 + the language term of the XE criterion as ONE autograd node: rfn_decoder_fwd stops at the
    logits (time-major rows in the workspace), rfn_xe_logits_fwd turns them into the masked (label-smoothed) NLL and the rows'
    logsumexps; backward writes d logits over the logits (rfn_xe_logits_bwd, scaled by the upstream gradient read on the
    device) and rfn_decoder_bwd runs from there.  Neither log_prob (B, S, V+1) nor its gradient is ever materialised
    (misc/RecurrentFusionModel.py:276 + misc/utils.py:163-184)."""

    @staticmethod
    def forward(ctx, model, save_bwd, drop, seed, ids, target, mask, eps, comb, h0, c0, *params):
        d = model._dims_for(drop)
        B, S = ids.shape
        dev = comb.device
        comb, h0, c0, ids = comb.contiguous(), h0.contiguous(), c0.contiguous(), ids.contiguous()
        table = model._param_table(params, model._decoder_slots)
        train = bool(save_bwd)
        ws_bytes = N.lib.rfn_decoder_ws_bytes(C.byref(d), B, S, int(train))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        st = N.stream_ptr()
        N.check(N.lib.rfn_decoder_fwd(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(), ids.data_ptr(),
                                      ids.stride(0), None, ws.data_ptr(), ws_bytes, int(train), seed, st), 'rfn_decoder_fwd')
        logits = N.lib.rfn_decoder_logits(C.byref(d), B, S, int(train), ws.data_ptr())
        lse = torch.empty(S * B, device=dev)
        scratch = torch.empty(B * S, device=dev)
        loss = torch.zeros(1, device=dev)
        N.check(N.lib.rfn_xe_logits_fwd(logits, d.V1, B, S, d.V1, target.data_ptr(), target.stride(0), mask.data_ptr(),
                                        mask.stride(0), float(eps), lse.data_ptr(), scratch.data_ptr(), loss.data_ptr(), 0, st),
                'rfn_xe_logits_fwd')
        if train:
            ctx.model, ctx.seed, ctx.B, ctx.S, ctx.drop, ctx.eps = model, seed, B, S, drop, float(eps)
            ctx.ids, ctx.target, ctx.mask, ctx.params, ctx.consumed = ids, target, mask, params, False
            ctx.save_for_backward(comb, h0, c0, lse, ws)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        model, B, S = ctx.model, ctx.B, ctx.S
        comb, h0, c0, lse, ws = ctx.saved_tensors
        d = model._dims_for(ctx.drop)
        dev = comb.device
        st = N.stream_ptr()
        table = model._param_table(ctx.params, model._decoder_slots)
        flats, by_slot, gtable = model._grad_buffers(['decoder'], dev)
        if ctx.consumed:
            # the first backward turned the logits into their gradient and the activations into theirs: a second backward over
            # the same graph recomputes the pass first (same inputs, same dropout seed, the weights as they are now)
            N.check(N.lib.rfn_decoder_fwd(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                          ctx.ids.data_ptr(), ctx.ids.stride(0), None, ws.data_ptr(), ws.numel(), 1, ctx.seed, st),
                    'rfn_decoder_fwd (recompute)')
            lse = torch.empty_like(lse)
            tmp, junk = torch.empty(B * S, device=dev), torch.zeros(1, device=dev)
            N.check(N.lib.rfn_xe_logits_fwd(N.lib.rfn_decoder_logits(C.byref(d), B, S, 1, ws.data_ptr()), d.V1, B, S, d.V1,
                                            ctx.target.data_ptr(), ctx.target.stride(0), ctx.mask.data_ptr(), ctx.mask.stride(0),
                                            ctx.eps, lse.data_ptr(), tmp.data_ptr(), junk.data_ptr(), 0, st), 'rfn_xe_logits_fwd')
        ctx.consumed = True
        g = g.contiguous().float()
        logits = N.lib.rfn_decoder_logits(C.byref(d), B, S, 1, ws.data_ptr())
        N.check(N.lib.rfn_xe_logits_bwd(logits, d.V1, B, S, d.V1, ctx.target.data_ptr(), ctx.target.stride(0),
                                        ctx.mask.data_ptr(), ctx.mask.stride(0), ctx.eps, lse.data_ptr(), 1.0, g.data_ptr(), st),
                'rfn_xe_logits_bwd')
        d_comb, d_h0, d_c0 = torch.empty_like(comb), torch.empty_like(h0), torch.empty_like(c0)
        N.check(N.lib.rfn_decoder_bwd(C.byref(d), B, S, table, comb.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                      ctx.ids.data_ptr(), ctx.ids.stride(0), None, None, d_comb.data_ptr(), d_h0.data_ptr(),
                                      d_c0.data_ptr(), gtable, ws.data_ptr(), ws.numel(), ctx.seed, st), 'rfn_decoder_bwd')
        model._bucket_done('decoder', flats['decoder'])
        return (None,) * 8 + (d_comb, d_h0, d_c0) + (None,) * len(ctx.params)


class RecurrentFusionModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        # the fields the reference reads (misc/RecurrentFusionModel.py:120-151)
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_type = getattr(opt, 'rnn_type', 'lstm')
        self.rnn_size = opt.rnn_size
        self.num_layers = getattr(opt, 'num_layers', 1)
        self.drop_prob_lm = opt.drop_prob_lm
        self.drop_prob_reason = opt.drop_prob_reason
        self.drop_prob_fusion = opt.drop_prob_fusion
        self.seq_length = opt.seq_length
        self.num_review_steps = opt.num_review_steps
        self.num_review_steps_0 = opt.num_review_steps_0
        self.top_words_count = opt.top_words_count
        self.att_hid_size = opt.att_hid_size
        self.ss_prob = 0.0
        self.review_maxout = opt.review_maxout
        self.decoder_maxout = opt.maxout
        self.fusion_maxout = opt.fusion_maxout  # accepted and ignored, exactly like the reference (:94-96)
        self.use_cuda = getattr(opt, 'use_cuda', 1)
        self.feat_array_info = opt.feat_array_info
        M = self.num_feat_array = len(self.feat_array_info)
        self.fc_feat_size = [f['fc_feat_size'] for f in self.feat_array_info]
        self.att_feat_size = [f['att_feat_size'] for f in self.feat_array_info]
        self.att_num = [f['att_num'] for f in self.feat_array_info]
        R, A, E = self.rnn_size, self.att_hid_size, self.input_encoding_size

        self.fc2h = nn.ModuleList([nn.Linear(self.fc_feat_size[i], R) for i in range(M)])
        self.embed = nn.Embedding(self.vocab_size + 1, E)
        self.logit = nn.Linear(R, self.vocab_size + 1)
        self.review_steps_individual = nn.ModuleList(
            [_FusionStepParams(M, R, self.att_feat_size, A) for _ in range(self.num_review_steps_0)])
        self.reason_linear_individual = nn.ModuleList([nn.Linear(R, self.top_words_count) for _ in range(M)])
        self.review_steps = nn.ModuleList([_ReviewStepParams(M, R, A, self.review_maxout)
                                           for _ in range(self.num_review_steps)])
        self.reason_linear = nn.Linear(R, self.top_words_count)
        self.decoder = _DecoderParams(E, R, A, self.decoder_maxout)
        self.init_weights()

        # RFN_GEMM_OPT_* bits handed to every GEMM of the path (rfn.h); a data-parallel host sets
        # N.GEMM_OPT_LDS_LEAN so RCCL's kernels can co-reside with the long weight-gradient GEMMs (parallel.GradSync does)
        self.gemm_flags = 0
        self.path_flags = 0              # RFN_PATH_OPT_* bits (rfn.h); 0 = defaults
        self._probe = None
        self._dims = {}
        self._param_cache = {}
        self._slot_names = N.param_names(self._dims_for(False))
        schema = dict(self.named_parameters())
        missing = [n for n in self._slot_names if n not in schema]
        if missing or len(schema) != len(self._slot_names):
            raise N.RfnError('parameter schema mismatch between module and librfn_hip.so: %s' % missing[:3])
        is_dec = lambda n: n.startswith(('embed.', 'logit.', 'decoder.'))  # noqa: E731
        self._prefix_slots = [i for i, n in enumerate(self._slot_names) if not is_dec(n)]
        self._decoder_slots = [i for i, n in enumerate(self._slot_names) if is_dec(n)]
        # gradient buckets, in the order their gradients become final during backward: the decoder, the
        # fusion "core" (everything of phase 1 except the per-encoder stage-I weights) and two buckets per
        # encoder (rfn_prefix_bwd_wgrad parts 1 and 2; all a-buckets are produced before the first b-bucket).  Each bucket is one flat buffer = one all-reduce = one Adam launch.
        enc_re = re.compile(r'^review_steps_individual\.\d+\.lstm\.(\d+)\.(att_model\.att_2_att_h|att_model\.h_2_att_h|H2h|z2h)\.')
        self._bucket_slots = {'decoder': list(self._decoder_slots), 'core': []}
        for i in range(M):
            self._bucket_slots['enc%da' % i] = []    # H2h, z2h, h_2_att_h of encoder i (rfn_prefix_bwd_wgrad part 1)
            self._bucket_slots['enc%db' % i] = []    # att_2_att_h of encoder i            (part 2)
        for idx in self._prefix_slots:
            m = enc_re.match(self._slot_names[idx])
            if not m:
                self._bucket_slots['core'].append(idx)
            else:
                self._bucket_slots['enc%s%s' % (m.group(1), 'b' if 'att_2_att_h' in m.group(2) else 'a')].append(idx)
        self._prefix_buckets = ['core'] + ['enc%da' % i for i in range(M)] + ['enc%db' % i for i in range(M)]   # production order
        self.grad_ready_hook = None      # callable(bucket_name, flat_grad_tensor), see parallel.GradSync
        # callable('prefix' | 'decoder'): called right before the parameters of that phase are handed to a kernel.  A
        # sharded optimizer (FusedClampAdam(shard=...)) waits there for the all-gather of the parameters it updated.
        self.param_wait_hook = None
        # state_dict() (the reference's checkpoint pattern right after optimizer.step(), train.py:226-233) must not read
        # parameters an asynchronous update is still writing: join first
        self.register_state_dict_pre_hook(_join_before_state_dict)
        # flat buckets (gradients, and the optimizer's parameter / moment buffers) are padded to a multiple of this many
        # elements: 4 * world_size under a sharded optimizer, so every rank's shard is the same 16-B aligned length
        self.flat_pad = 1
        # Opt-in: real batches hold each image's features `seq_per_img` times in a row (dataloader.py:251-252).
        # With this set to that count, stages I/II run once per image and their outputs are fanned out to the
        # caption rows (exactly the same numbers: rows are independent); only legal while drop_prob_fusion and
        # drop_prob_reason are 0, which forward() checks.
        self.dedup_seq_per_img = 0
        # Opt-in: the self-critical loop samples twice from the same batch (train_rl.py:160-166: a multinomial sample
        # with grad, then the greedy baseline under no_grad).  With this set, a no-grad call whose input tensors are
        # the very same objects (unchanged `_version`) as the previous call's, with unchanged weights and no active
        # dropout, reuses that call's stage-I/II outputs instead of recomputing them.
        self.reuse_prefix = False
        # Opt-in for loops that call loss.backward(retain_graph=True) repeatedly with optimizer steps in between (PPO,
        # train_rl.py:190-201): keep a copy of the activations backward consumes, so later passes differentiate the
        # ORIGINAL forward's activations through the current weights, as the reference does.  Off: they are recomputed
        # at the current weights.
        self.retain_activations = False
        self._prefix_cache = None
        self._weights_epoch = 0          # bumped by FusedClampAdam.step (it writes parameters behind autograd's back)
        self._last_flat_grads = {}
        self._steps_cache = None
        # set by graphed.GraphedTrainStep while it captures / warms up: run exactly this many decoder steps instead of
        # stopping at the first all-zero label column (the loader's masks are zero on the extra steps)
        self.fixed_decoder_steps = None
        self.done_beams = []

    def init_weights(self):
        """misc/RecurrentFusionModel.py:188-196."""
        _uniform(self.embed.weight, self.logit.weight, self.reason_linear.weight)
        self.logit.bias.data.fill_(0)
        for i in range(self.num_feat_array):
            _uniform(self.reason_linear_individual[i].weight, self.fc2h[i].weight)

    # ---- plumbing -------------------------------------------------------------------------------
    def set_probe_events(self, events):
        """Measurement hook (rfn.h rfn_dims.probe_events): 4*M `torch.cuda.Event(enable_timing=True)` objects that the path
        records around its dominant launches -- [2i], [2i+1] encoder i's hoisted projection (forward), [2M+2i], [2M+2i+1] its
        att_2_att_h weight gradient (backward) -- or None to switch the hook off.  The events must have been recorded once
        (that is when PyTorch creates the HIP event behind them); the caller keeps them alive while the hook is set."""
        if events is None:
            self._probe = None
            return
        if len(events) != 4 * self.num_feat_array:
            raise N.RfnError('set_probe_events: expected %d events' % (4 * self.num_feat_array))
        arr = (C.c_void_p * len(events))(*[int(e.cuda_event) for e in events])
        self._probe = (arr, list(events))

    def _dims_for(self, train: bool) -> N.Dims:
        d = self._dims_cached(train)
        probe = getattr(self, '_probe', None)
        d.probe_events = C.cast(probe[0], C.c_void_p) if probe is not None else None
        return d

    def _dims_cached(self, train: bool) -> N.Dims:
        key = (bool(train), int(self.gemm_flags), int(self.path_flags))
        if key not in self._dims:
            self._dims[key] = N.make_dims(
                self.num_feat_array, self.rnn_size, self.att_hid_size, self.input_encoding_size,
                self.num_review_steps_0, self.num_review_steps, self.top_words_count, self.vocab_size + 1,
                self.att_num, self.att_feat_size, self.fc_feat_size,
                review_maxout=self.review_maxout, decoder_maxout=self.decoder_maxout,
                drop_fusion=self.drop_prob_fusion if train else 0.0,
                drop_reason=self.drop_prob_reason if train else 0.0,
                drop_lm=self.drop_prob_lm if train else 0.0, gemm_flags=self.gemm_flags, path_flags=self.path_flags)
        return self._dims[key]

    def _params_of(self, slots):
        """The parameters of a phase, about to be handed to its kernels: the one place the compute stream meets an
        asynchronous update of them again (`param_wait_hook`: a sharded optimizer's all-gathers, an overlapped update's side
        stream).  Forward entry points only -- bookkeeping that merely needs the parameter OBJECTS (bucket layouts, gradient
        views) uses `_params_lookup` and waits for nothing (the hook used to fire from inside backward)."""
        if self.param_wait_hook is not None:
            self.param_wait_hook('decoder' if slots is self._decoder_slots else 'prefix')
        return self._params_lookup(slots)

    def _params_lookup(self, slots):
        # Parameter objects are stable under .cuda()/.to() (their .data is swapped in place), so the
        # by-name lookup is done once per slot list
        key = id(slots)
        if key not in self._param_cache:
            named = dict(self.named_parameters())
            self._param_cache[key] = [named[self._slot_names[i]] for i in slots]
        return self._param_cache[key]

    def _param_table(self, params, slots):
        table = (C.c_void_p * len(self._slot_names))()
        for p, i in zip(params, slots):
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise N.RfnError('parameter %s must be a contiguous float32 GPU tensor' % self._slot_names[i])
            table[i] = p.data_ptr()
        return table

    def bucket_names(self):
        return ['decoder'] + list(self._prefix_buckets)

    def bucket_layout(self, name):
        """(params, offsets, total) of one bucket: parameters back to back, each start 16-B aligned."""
        params = self._params_lookup(self._bucket_slots[name])
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) & ~3
        pad = max(1, int(self.flat_pad))
        return params, offs, -(-total // pad) * pad

    def _grad_buffers(self, buckets, dev):
        """One flat gradient buffer per bucket; returns ({name: flat}, {slot: view}, pointer table)."""
        flats, by_slot = {}, {}
        table = (C.c_void_p * len(self._slot_names))()
        for name in buckets:
            params, offs, total = self.bucket_layout(name)
            flat = torch.empty(total, device=dev, dtype=torch.float32)
            if self.flat_pad > 1:       # the shard padding behind the last parameter is exchanged and updated like data
                flat[offs[-1] + params[-1].numel():].zero_()
            flats[name] = flat
            for slot, p, o in zip(self._bucket_slots[name], params, offs):
                v = flat[o:o + p.numel()].view_as(p)
                by_slot[slot] = v
                table[slot] = v.data_ptr()
        return flats, by_slot, table

    def _bucket_done(self, name, flat):
        """Backward has queued every kernel that writes bucket `name`: hand its gradients over.

        `.grad` of the bucket's parameters become VIEWS of the flat buffer, so `.grad`, the all-reduce operand and the
        fused optimizer's operand are the same memory (returning the gradients through autograd instead makes
        AccumulateGrad clone most of them: hundreds of small D2D copies per step).  A second backward before
        zero_grad() accumulates into the buffer already registered -- as autograd would -- so the optimizer operand
        always equals `.grad`."""
        params, offs, _ = self.bucket_layout(name)
        live = self._last_flat_grads.get(name)
        first = next((i for i, p in enumerate(params) if p.requires_grad), None)
        if (live is not None and first is not None and params[first].grad is not None
                and params[first].grad.data_ptr() == live.data_ptr() + 4 * offs[first]):
            if self.grad_ready_hook is not None:
                raise N.RfnError('bucket %s was produced twice before zero_grad(): gradient accumulation over several '
                                 'backward passes cannot be combined with a grad_ready_hook (GradSync all-reduces each '
                                 'bucket once per step)' % name)
            live.add_(flat)
            return
        for p, o in zip(params, offs):
            if not p.requires_grad:
                continue
            v = flat[o:o + p.numel()].view_as(p)
            if p.grad is not None:      # a gradient that did not come from this path (or was never zeroed): fold it in
                v.add_(p.grad)
            p.grad = v
        self._last_flat_grads[name] = flat
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(name, flat)

    def _check_inputs(self, fc_feats, att_feats):
        M = self.num_feat_array
        if len(fc_feats) != M or len(att_feats) != M:
            raise N.RfnError('expected %d fc / att feature tensors' % M)
        B = fc_feats[0].size(0)
        for i in range(M):
            if tuple(fc_feats[i].shape) != (B, self.fc_feat_size[i]):
                raise N.RfnError('fc_feats[%d] has shape %s, expected %s' % (
                    i, tuple(fc_feats[i].shape), (B, self.fc_feat_size[i])))
            if tuple(att_feats[i].shape) != (B, self.att_num[i], self.att_feat_size[i]):
                raise N.RfnError('att_feats[%d] has shape %s, expected %s' % (
                    i, tuple(att_feats[i].shape), (B, self.att_num[i], self.att_feat_size[i])))
        return B

    def _prefix(self, fc_feats, att_feats, drop, seed):
        """-> comb (T2,B,R) time-major, h, c (B,R), reason (M+1,B,K).  `drop`: apply dropout (training mode)."""
        self._check_inputs(fc_feats, att_feats)
        params = self._params_of(self._prefix_slots)
        cacheable = self.reuse_prefix and not (drop and (self.drop_prob_fusion > 0 or self.drop_prob_reason > 0))
        if cacheable:
            ins = list(fc_feats) + list(att_feats)
            stamp = (tuple(t._version for t in ins), sum(p._version for p in params), self._weights_epoch)
            hit = self._prefix_cache
            if (hit is not None and not torch.is_grad_enabled() and hit[1] == stamp and len(hit[0]) == len(ins)
                    and all(r() is t for r, t in zip(hit[0], ins))):
                return hit[2]
        out = _PrefixFn.apply(self, torch.is_grad_enabled(), bool(drop), seed, self.num_feat_array, *fc_feats,
                              *att_feats, *params)
        if cacheable:
            # weak references: a dead input invalidates the entry, so a recycled address can never alias it
            self._prefix_cache = ([weakref.ref(t) for t in ins], stamp, tuple(o.detach() for o in out))
        return out

    def _decode_teacher_forced(self, ids, comb, h, c, drop, seed, ss_prob=0.0):
        params = self._params_of(self._decoder_slots)
        return _DecoderFn.apply(self, torch.is_grad_enabled(), bool(drop), seed, ids, comb, h, c, float(ss_prob), 1.0,
                                *params)[0]

    def _decode_sampled(self, comb, h, c, drop, seed, steps, inv_temp):
        """Free-running multinomial decode as ONE step-wise pass of the training decoder (every row's next token is
        drawn from the distribution the step just produced): -> (log_prob (B, steps, V+1), ids (B, steps) fed)."""
        params = self._params_of(self._decoder_slots)
        ids0 = torch.zeros(h.size(0), steps, dtype=torch.long, device=h.device)     # column 0 = BOS
        return _DecoderFn.apply(self, torch.is_grad_enabled(), bool(drop), seed, ids0, comb, h, c, 1.0, float(inv_temp),
                                *params)

    # ---- reference API ----------------------------------------------------------------------------
    def forward(self, fc_feats, att_feats, seq):
        """misc/RecurrentFusionModel.py:198-281 -> (log_prob (B,T,V+1), reason_pred list[M+1] of (B,K))."""
        train = bool(self.training)
        seed = _fresh_seed() if train else 0
        S = self._decoder_steps(seq)          # may read `seq` back once: do it before queueing phase 1
        g = int(self.dedup_seq_per_img)
        if g > 1:
            if train and (self.drop_prob_fusion > 0 or self.drop_prob_reason > 0):
                raise N.RfnError('dedup_seq_per_img needs drop_prob_fusion = drop_prob_reason = 0')
            rows, n_feat = seq.size(0), fc_feats[0].size(0)
            if n_feat == rows and rows % g == 0:      # caption rows (the loader's layout): keep one row per image
                fc_u, att_u = [f[::g] for f in fc_feats], [a[::g] for a in att_feats]
            elif n_feat * g == rows:                  # unique images already (FeatureFeeder.batch(expand=False))
                fc_u, att_u = list(fc_feats), list(att_feats)
            else:
                raise N.RfnError('dedup_seq_per_img = %d: %d feature rows do not match %d caption rows' % (g, n_feat, rows))
            comb, h, c, reason = self._prefix(fc_u, att_u, train, seed)
            comb = comb.repeat_interleave(g, dim=1)       # autograd sums the caption rows back onto the image
            h, c = h.repeat_interleave(g, dim=0), c.repeat_interleave(g, dim=0)
            reason = reason.repeat_interleave(g, dim=1)
        else:
            comb, h, c, reason = self._prefix(fc_feats, att_feats, train, seed)
        log_prob = self._decode_teacher_forced(seq[:, :S], comb, h, c, train, seed, self.ss_prob)
        return log_prob, list(reason.unbind(0))

    def forward_loss(self, fc_feats, att_feats, seq, masks, top_words, crit, reason_weight=1.0):
        """The XE train step's forward AND criterion in one call (opt-in) ->
        (loss, reason_pred): the value and the gradients of
            log_prob, reason_pred = model(fc_feats, att_feats, seq)
            loss = crit(log_prob, seq[:, 1:], masks[:, 1:], reason_pred, top_words, reason_weight)     (train.py:154-159)
        with `crit` a ReviewNetEnsembleCriterion (its label-smoothing setting is honoured), without the (B, T, V+1) log_prob
        tensor, its gradient and the four passes over them: the language term comes straight from the logits and its
        backward writes d logits in place (rfn_xe_logits_fwd / _bwd).  Equal to the two-call form to rounding (the row
        logsumexp is the same; the loss terms are summed in the same order).  Scheduled sampling needs the per-step
        distributions and takes the two-call form."""
        from .criteria import _MLMFn
        if self.ss_prob > 0:
            log_prob, reason = self.forward(fc_feats, att_feats, seq)
            return crit(log_prob, seq[:, 1:], masks[:, 1:], reason, top_words, reason_weight), reason
        train = bool(self.training)
        seed = _fresh_seed() if train else 0
        S = self._decoder_steps(seq)
        if seq.size(1) < S + 1 or masks.size(1) < S + 1:
            raise N.RfnError('labels / masks need %d columns for %d decoder steps' % (S + 1, S))
        if int(self.dedup_seq_per_img) > 1:
            raise N.RfnError('forward_loss does not combine with dedup_seq_per_img; use forward() + the criterion')
        dev = fc_feats[0].device
        comb, h, c, reason = self._prefix(fc_feats, att_feats, train, seed)
        seq = seq.to(dev).long()
        masks = masks.to(dev).float()
        if seq.stride(1) != 1:
            seq = seq.contiguous()
        if masks.stride(1) != 1:
            masks = masks.contiguous()
        eps = float(crit.label_smoothing_epsilon) if crit.use_label_smoothing else 0.0
        params = self._params_of(self._decoder_slots)
        lang = _DecoderLossFn.apply(self, torch.is_grad_enabled(), train, seed, seq[:, :S], seq[:, 1:S + 1], masks[:, 1:S + 1], eps,
                                    comb, h, c, *params)
        heads = list(reason.unbind(0))
        return lang + _MLMFn.apply(float(reason_weight) / len(heads), top_words, *heads), heads

    def _decoder_steps(self, seq):
        """Number of decoder steps: the reference breaks at the first all-zero column i >= 1 (:274).  The
        answer for a given (tensor, version) is cached, so a caller that reuses one label tensor pays the
        device read-back once instead of the reference's 17 syncs per forward."""
        if self.fixed_decoder_steps is not None:      # a captured step runs a fixed count (graphed.GraphedTrainStep)
            if not 1 <= self.fixed_decoder_steps <= seq.size(1):
                raise N.RfnError('fixed_decoder_steps = %d does not fit labels of %d columns' % (self.fixed_decoder_steps, seq.size(1)))
            return int(self.fixed_decoder_steps)
        key = (seq.data_ptr(), seq._version, tuple(seq.shape))
        hit = self._steps_cache
        if hit is not None and hit[0] == key and hit[1] is seq:
            return hit[2]
        nz = (seq != 0).any(0).tolist()
        S = seq.size(1)
        for i in range(1, seq.size(1)):
            if not nz[i]:
                S = i
                break
        self._steps_cache = (key, seq, S)     # holds `seq` so its storage cannot be recycled under the key
        return S

    def get_init_state(self, fc_feats):
        """misc/RecurrentFusionModel.py:333-343 (inference helper, no autograd)."""
        out = []
        with torch.no_grad():
            for i in range(self.num_feat_array):
                h0 = N.linear(fc_feats[i], self.fc2h[i].weight, self.fc2h[i].bias).unsqueeze(0)
                out.append((h0, h0.clone()))
        return out

    def get_thought_vectors(self, fc_feats, att_feats, state_list=None):
        """misc/RecurrentFusionModel.py:283-331 -> (thought_vectors_comb (B,T2,R), reason_pred, state_review).
        `state_list`: the M (h, c) pairs of shape (1,B,R) that get_init_state returns -- used as given, like the
        reference; None recomputes them from fc_feats."""
        with torch.no_grad():
            if state_list is None:
                comb, h, c, reason = self._prefix(fc_feats, att_feats, False, 0)
            else:
                comb, h, c, reason = self._prefix_from_state(att_feats, state_list)
        return (comb.transpose(0, 1).contiguous(), list(reason.unbind(0)),
                (h.unsqueeze(0), c.unsqueeze(0)))

    def _prefix_from_state(self, att_feats, state_list):
        M, R = self.num_feat_array, self.rnn_size
        if len(state_list) != M or len(att_feats) != M:
            raise N.RfnError('expected %d (h, c) states and att feature tensors' % M)
        B = att_feats[0].size(0)
        hs = [N.require_cuda_f32(s_[0].reshape(-1, R), 'state h') for s_ in state_list]
        cs = [N.require_cuda_f32(s_[1].reshape(-1, R), 'state c') for s_ in state_list]
        att = [N.require_cuda_f32(a, 'att_feats') for a in att_feats]
        for i in range(M):
            if hs[i].size(0) != B or cs[i].size(0) != B:
                raise N.RfnError('state %d has batch %d, features have %d' % (i, hs[i].size(0), B))
            if tuple(att[i].shape) != (B, self.att_num[i], self.att_feat_size[i]):
                raise N.RfnError('att_feats[%d] has shape %s' % (i, tuple(att[i].shape)))
        d = self._dims_for(False)
        dev = att[0].device
        K = self.top_words_count
        table = self._param_table(self._params_of(self._prefix_slots), self._prefix_slots)
        ws_bytes = N.lib.rfn_prefix_ws_bytes(C.byref(d), B, 0)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        comb = torch.empty(d.T2, B, R, device=dev)
        h, c = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev)
        reason = torch.empty(M + 1, B, K, device=dev)
        N.check(N.lib.rfn_prefix_fwd_from_state(C.byref(d), B, table, N.ptr_array(hs), N.ptr_array(cs),
                                                N.ptr_array(att), comb.data_ptr(), h.data_ptr(), c.data_ptr(),
                                                reason.data_ptr(), ws.data_ptr(), ws_bytes, N.stream_ptr()),
                'rfn_prefix_fwd_from_state')
        return comb, h, c, reason

    def one_time_step(self, xt, fc_feats, thought_vectors_comb, state_decode):
        """misc/RecurrentFusionModel.py:345-350 -> (logit (B,V+1) pre-softmax, state).  `xt` is the embedded
        input (B, E) exactly as in the reference (its callers do xt = model.embed(it), eval_utils.py:368), or int64
        token ids (B,), in which case the embedding is fused into the step."""
        with torch.no_grad():
            comb = thought_vectors_comb.transpose(0, 1).contiguous()
            drop = bool(self.training)
            stepper = _Stepper(self, comb, state_decode[0][-1].clone(), state_decode[1][-1].clone(), drop,
                               _fresh_seed() if drop else 0)
            logits = stepper.step(xt.contiguous(), want='logits')
            return logits, (stepper.h.unsqueeze(0), stepper.c.unsqueeze(0))

    def sample(self, fc_feats, att_feats, opt={}):
        """misc/RecurrentFusionModel.py:545-658."""
        sample_max = opt.get('sample_max', 1)
        beam_size = opt.get('beam_size', 1)
        temperature = opt.get('temperature', 1.0)
        if beam_size > 1:
            return self.sample_beam(fc_feats, att_feats, opt)
        want_grad = torch.is_grad_enabled() and not sample_max
        # dropout follows the module's mode, as the reference's nn.Dropout layers do (train_rl.py samples in train()
        # mode)
        train = bool(self.training)
        seed = _fresh_seed() if train else 0
        with torch.set_grad_enabled(want_grad):
            comb, h, c, reason = self._prefix(fc_feats, att_feats, train, seed)
        B, S, V1 = fc_feats[0].size(0), self.seq_length, self.vocab_size + 1
        dev = comb.device
        reason_pred = list(reason.unbind(0))
        force = opt.get('force_ids', None)
        if not sample_max or force is not None:
            # multinomial (:623-631) or replayed ids.  ONE pass of the training decoder is both the pass that is sampled
            # and, under grad, the pass that is differentiated (train_rl.py:160-166): step-wise with a device-side
            # inverse-CDF draw between steps, or -- when the ids are given -- simply teacher-forced.
            with torch.set_grad_enabled(want_grad):
                if force is not None:
                    raw = torch.zeros(B, S + 1, dtype=torch.long, device=dev)      # column t = token fed at step t
                    raw[:, 1:] = force[:, :S].to(dev)
                else:
                    logp_full, raw = self._decode_sampled(comb, h, c, train, seed, S + 1, 1.0 / float(temperature))
                tok = raw[:, 1:]
                unf = torch.cumprod((tok > 0).long(), 1)                           # a row is finished after its first 0
                seq = tok * unf
                alive = unf.sum(0).tolist()                                        # one read-back: the early exit (:645)
                t_stop = next((t for t in range(1, S + 1) if alive[t - 1] == 0), S + 1)
                n_seq = t_stop - 1
                if force is not None:
                    logp = self._decode_teacher_forced(raw[:, :t_stop].contiguous(), comb, h, c, train, seed)
                else:
                    logp = logp_full[:, :t_stop]
                seq_lp = logp[:, :n_seq].gather(2, tok[:, :n_seq].unsqueeze(2)).squeeze(2)
            if getattr(self, '_trace_ss', False):
                self._sample_ids = raw[:, :t_stop].clone()       # test hook: the tokens the pass fed
            return seq[:, :n_seq], seq_lp, logp.contiguous(), reason_pred
        with torch.no_grad():      # greedy: the whole free-running loop (pick, embed, cell, logit, log-softmax) in one call
            stepper = _Stepper(self, comb.detach(), h.detach().clone(), c.detach().clone(), train, seed)
            logp_all = torch.empty(B, S + 1, V1, device=dev)
            seq = torch.zeros(B, S, dtype=torch.long, device=dev)
            seq_lp = torch.zeros(B, S, device=dev)
            unf = torch.zeros(S + 1, B, dtype=torch.int32, device=dev)
            it = torch.empty(B, dtype=torch.long, device=dev)
            N.check(N.lib.rfn_decoder_loop(C.byref(stepper.d), B, S + 1, stepper.table, stepper.comb.data_ptr(),
                                           stepper.cproj.data_ptr(), stepper.h.data_ptr(), stepper.c.data_ptr(), 0, 1.0, None,
                                           logp_all.data_ptr(), logp_all.stride(0), logp_all.stride(1), seq.data_ptr(),
                                           seq.stride(0), seq_lp.data_ptr(), seq_lp.stride(0), unf.data_ptr(), it.data_ptr(),
                                           stepper.ws.data_ptr(), stepper.ws_bytes, stepper.seed, N.stream_ptr()),
                    'rfn_decoder_loop')
            # the reference's early exit (:645): stop at the first t >= 1 with no unfinished row
            alive = unf[1:].sum(1).tolist()
        t_stop = next((t for t in range(1, S + 1) if alive[t - 1] == 0), S + 1)
        n_seq = t_stop - 1
        return seq[:, :n_seq], seq_lp[:, :n_seq], logp_all[:, :t_stop].contiguous(), reason_pred

    def sample_beam(self, fc_feats, att_feats, opt={}):
        """misc/RecurrentFusionModel.py:352-543, batched and device-resident.

        Stages I/II run ONCE for the batch (the reference recomputes them per image on beam_size identical rows),
        all images' beams share one decoder batch of B*beam rows, and rfn_beam_step does the reference's
        candidate / stable-sort / fork / done-beam bookkeeping on the device -- no per-step host read-back.  The
        done beams are sorted (stably, by -p, as :529) on the host once at the end."""
        beam_size = opt.get('beam_size', 10)
        B, S, V1 = fc_feats[0].size(0), self.seq_length, self.vocab_size + 1
        assert beam_size <= V1, 'lets assume this for now'
        if beam_size > 32 or S > 64:
            raise N.RfnError('beam search supports beam_size <= 32 and seq_length <= 64')
        W = beam_size
        with torch.no_grad():
            drop = bool(self.training)
            seed = _fresh_seed() if drop else 0
            comb_b, h_b, c_b, reason = self._prefix(fc_feats, att_feats, drop, seed)
            dev = comb_b.device
            # the W beam rows of image k (rows k*W .. k*W+W-1) read the image's thought vectors: comb stays (T2, B, R)
            stepper = _Stepper(self, comb_b, h_b.repeat_interleave(W, dim=0).contiguous(),
                               c_b.repeat_interleave(W, dim=0).contiguous(), drop, seed)
            rows, max_done = B * W, W * S
            bs = torch.zeros(S, B, W, dtype=torch.long, device=dev)
            bl = torch.zeros(S, B, W, device=dev)
            bsum = torch.zeros(B, W, device=dev)
            order = torch.zeros(rows, dtype=torch.int32, device=dev)
            ids = torch.zeros(rows, dtype=torch.long, device=dev)
            done_seq = torch.zeros(B, max_done, S, dtype=torch.long, device=dev)
            done_lp = torch.zeros(B, max_done, S, device=dev)
            done_p = torch.zeros(B, max_done, device=dev)
            done_n = torch.zeros(B, dtype=torch.int32, device=dev)
            active = torch.ones(B, dtype=torch.int32, device=dev)
            logp = torch.empty(2 * rows * W, device=dev)           # the rows' top-W lists (rfn_beam_loop)
            h_alt, c_alt = torch.empty_like(stepper.h), torch.empty_like(stepper.c)
            # the whole search in one call: S x (bookkeeping, state re-gather, decoder step on the B * W rows)
            N.check(N.lib.rfn_beam_loop(C.byref(stepper.d), B, W, S, stepper.table, stepper.comb.data_ptr(),
                                        stepper.cproj.data_ptr(), stepper.h.data_ptr(), stepper.c.data_ptr(), h_alt.data_ptr(),
                                        c_alt.data_ptr(), logp.data_ptr(), bs.data_ptr(), bl.data_ptr(), bsum.data_ptr(),
                                        order.data_ptr(), ids.data_ptr(), done_seq.data_ptr(), done_lp.data_ptr(),
                                        done_p.data_ptr(), done_n.data_ptr(), active.data_ptr(), max_done,
                                        stepper.ws.data_ptr(), stepper.ws_bytes, stepper.seed, N.stream_ptr()), 'rfn_beam_loop')
            seq, seq_lp, top_seq, top_prob, done_beams = _sorted_done_beams(done_seq, done_lp, done_p, done_n, S, max_done)
            heads = reason.unsqueeze(2).expand(-1, -1, W, -1)                       # (M+1, B, W, K) broadcast view
        self.done_beams = done_beams
        reason_batch = _LazyList(B, lambda: [list(t.unbind(0)) for t in heads.unbind(1)])
        return seq, seq_lp, top_seq, top_prob, reason_batch


def _sorted_done_beams(done_seq, done_lp, done_p, done_n, S, max_done):
    """Done beams sorted by -p, stably, as the reference's sorted(..., key=-p) (:529) -- on the device, for all images at
    once: the caller returns with everything queued and nothing read back, so the host's next batch (and its stage-I/II
    GEMMs) starts while this one is still decoding.  -> (seq (B, S) best done beam per image, its log-probs, and the
    per-image Python structures top_seq / top_prob / done_beams: thousands of small objects that need the done counts on
    the host, so they are lists that fill themselves on first access -- a loop that only consumes the returned captions
    never waits for them)."""
    dev, B = done_p.device, done_p.size(0)
    key = torch.where(torch.arange(max_done, device=dev)[None, :] < done_n[:, None], -done_p,
                      torch.full_like(done_p, float('inf')))
    rank = torch.sort(key, dim=1, stable=True).indices
    pick = rank[:, :, None].expand(-1, -1, S)
    s_all, l_all, p_all = done_seq.gather(1, pick), done_lp.gather(1, pick), done_p.gather(1, rank)
    src = _BeamResults(s_all, l_all, p_all, done_n)
    return (s_all[:, 0].contiguous(), l_all[:, 0].contiguous(), _LazyList(B, lambda: src.top_seq()),
            _LazyList(B, lambda: src.top_prob()), _LazyList(B, lambda: src.done_beams()))


class _BeamResults:
    """Host copies of the sorted done beams, fetched once, on demand."""

    def __init__(self, s_all, l_all, p_all, done_n):
        self._dev = (s_all, l_all, p_all, done_n)
        self._host = None

    def host(self):
        if self._host is None:
            s_all, l_all, p_all, done_n = self._dev
            self._host = (s_all.cpu(), l_all.cpu(), p_all.cpu().tolist(), done_n.cpu().tolist())
        return self._host

    def top_seq(self):
        s_all, _, _, counts = self.host()
        return [s_all[k, :n] for k, n in enumerate(counts)]

    def top_prob(self):
        _, _, probs, counts = self.host()
        return [probs[k][:n] for k, n in enumerate(counts)]

    def done_beams(self):
        """misc/RecurrentFusionModel.py:529-531: per image the list of {'seq', 'logps', 'p'} dicts, best first."""
        s_all, l_all, probs, counts = self.host()
        return [[{'seq': a, 'logps': b_, 'p': c_} for a, b_, c_ in zip(s_all[k, :n].unbind(0), l_all[k, :n].unbind(0), probs[k][:n])]
                for k, n in enumerate(counts)]


class _LazyList(collections.abc.MutableSequence):
    """A sequence of known length whose entries are produced (all at once) by `fill()` the first time anything but its
    length is asked for.  Deliberately NOT a subclass of `list`: C fast paths that take a list subclass (`PySequence_Fast`,
    `PyList_GET_ITEM`: json's C encoder, `str.join`, some torch / numpy converters) read the list's item array directly and
    would see unfilled placeholders without any Python-level hook running.  As a plain `MutableSequence` every consumer goes
    through `__getitem__` / `__iter__` / `__len__` (which fill first), and a consumer that insists on a real list fails
    loudly (`json.dumps(x)` raises TypeError; `json.dumps(list(x))` / `x.materialize()` is the spelling).  Indexing,
    slicing, iteration, comparison with lists, `+`, `in`, `reversed`, `sorted`, printing, copying, pickling (as a plain
    list) and in-place edits behave like the list the reference returns (misc/RecurrentFusionModel.py:529-543)."""

    __slots__ = ('_n', '_fill', '_items')
    __hash__ = None

    def __init__(self, n, fill):
        self._n, self._fill, self._items = int(n), fill, None

    def materialize(self):
        """The plain `list` behind this object (filled now if it was not)."""
        if self._items is None:
            fill, self._fill = self._fill, None
            items = list(fill())
            if len(items) != self._n:
                raise N.RfnError('lazy list promised %d entries, its producer made %d' % (self._n, len(items)))
            self._items = items
        return self._items

    def __len__(self):
        return self._n if self._items is None else len(self._items)

    def __getitem__(self, i):
        return self.materialize()[i]

    def __setitem__(self, i, v):
        self.materialize()[i] = v

    def __delitem__(self, i):
        del self.materialize()[i]

    def insert(self, i, v):
        self.materialize().insert(i, v)

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return repr(self.materialize())

    def _other(self, other):
        return other.materialize() if isinstance(other, _LazyList) else other

    def __eq__(self, other):
        return self.materialize() == self._other(other)

    def __ne__(self, other):
        return self.materialize() != self._other(other)

    def __lt__(self, other):
        return self.materialize() < self._other(other)

    def __le__(self, other):
        return self.materialize() <= self._other(other)

    def __gt__(self, other):
        return self.materialize() > self._other(other)

    def __ge__(self, other):
        return self.materialize() >= self._other(other)

    def __add__(self, other):
        return self.materialize() + list(self._other(other))

    def __radd__(self, other):
        return list(other) + self.materialize()

    def __mul__(self, k):
        return self.materialize() * k

    __rmul__ = __mul__

    def copy(self):
        return list(self.materialize())

    def sort(self, **kw):
        self.materialize().sort(**kw)

    def __reduce_ex__(self, protocol):
        return (list, (list(self.materialize()),))


def _join_before_state_dict(module, prefix, keep_vars):
    hook = getattr(module, 'param_wait_hook', None)
    if hook is not None:
        hook(None)


class _Stepper:
    """Free-running decoder state for sample / beam / one_time_step: rfn_decoder_prepare + rfn_decoder_step."""

    def __init__(self, model, comb, h, c, drop=False, seed=0):
        """drop / seed: apply the decoder dropout masks of (seed, step index) -- the ones rfn_decoder_fwd applies at
        the same steps -- so a free-running pass in training mode reproduces the teacher-forced pass bit for bit."""
        self.model = model
        self.d = model._dims_for(bool(drop))
        self.seed, self.t = int(seed), 0
        self.comb = comb.contiguous()
        self.h, self.c = h.contiguous(), c.contiguous()
        self.B = self.h.size(0)
        dev = self.h.device
        self.table = model._param_table(model._params_of(model._decoder_slots), model._decoder_slots)
        # the loop-invariant products of the thought vectors: att_2_att_h(comb) and U = comb . z2h.weight^T (rfn.h)
        self.Bc = self.comb.size(1)          # rows per thought vector: B, or the images of a beam search (B = Bc * beam)
        self.cproj = torch.empty(N.lib.rfn_decoder_cproj_floats(C.byref(self.d), self.Bc), device=dev)
        N.check(N.lib.rfn_decoder_prepare(C.byref(self.d), self.Bc, self.table, self.comb.data_ptr(),
                                          self.cproj.data_ptr(), N.stream_ptr()), 'rfn_decoder_prepare')
        self.ws_bytes = N.lib.rfn_decoder_step_ws_bytes(C.byref(self.d), self.B)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)

    def step(self, ids, out=None, want='logp'):
        """ids: int64 token ids (B,) -- or an already embedded float input (B, E), as the reference's one_time_step."""
        V1 = self.d.V1
        if self.Bc != self.B:
            raise N.RfnError('a stepper whose rows share thought vectors (beam search) is driven by rfn_beam_loop only')
        if out is None:
            out = torch.empty(self.B, V1, device=self.h.device)
        logits_ptr = out.data_ptr() if want == 'logits' else None
        logp_ptr = out.data_ptr() if want == 'logp' else None
        if ids.is_floating_point():
            xt = N.require_cuda_f32(ids, 'xt')
            if tuple(xt.shape) != (self.B, self.d.E):
                raise N.RfnError('embedded xt must be (%d, %d), got %s' % (self.B, self.d.E, tuple(xt.shape)))
            N.check(N.lib.rfn_decoder_step_embedded(C.byref(self.d), self.B, self.table, self.comb.data_ptr(),
                                                    self.cproj.data_ptr(), xt.data_ptr(), xt.stride(0),
                                                    self.h.data_ptr(), self.c.data_ptr(), logits_ptr, logp_ptr,
                                                    out.stride(0), self.ws.data_ptr(), self.ws_bytes, self.seed,
                                                    self.t, N.stream_ptr()),
                    'rfn_decoder_step_embedded')
            self.t += 1
            return out
        if ids.dtype != torch.long or not ids.is_contiguous():
            ids = ids.long().contiguous()
        N.check(N.lib.rfn_decoder_step(C.byref(self.d), self.B, self.table, self.comb.data_ptr(),
                                       self.cproj.data_ptr(), ids.data_ptr(), self.h.data_ptr(), self.c.data_ptr(),
                                       logits_ptr, logp_ptr, out.stride(0), self.ws.data_ptr(), self.ws_bytes,
                                       self.seed, self.t, N.stream_ptr()), 'rfn_decoder_step')
        self.t += 1
        return out

    def reorder(self, order):
        """Row r continues from row order[r] (int32, device): rfn_gather_rows on h and c."""
        if order.dtype != torch.int32 or not order.is_contiguous():
            order = order.to(torch.int32).contiguous()
        h2, c2 = torch.empty_like(self.h), torch.empty_like(self.c)
        st = N.stream_ptr()
        N.check(N.lib.rfn_gather_rows(self.h.data_ptr(), h2.data_ptr(), order.data_ptr(), self.B, self.d.R, st), 'rfn_gather_rows')
        N.check(N.lib.rfn_gather_rows(self.c.data_ptr(), c2.data_ptr(), order.data_ptr(), self.B, self.d.R, st), 'rfn_gather_rows')
        self.h, self.c = h2, c2
