"""One XE train step -- zero_grad, forward, criterion, backward, clamp + Adam (train.py:143-166) -- captured ONCE in a HIP
graph and replayed: for hosts whose launch path, not the device, paces a small configuration (BASELINE config 2 is 352
launches of 14 us: 4.84 ms of kernel time that a slow host stretches to 5.3-6.0 ms).  Opt-in; the eager path is the product's
default and the large configuration is device-bound either way.

What a capture freezes, and how each is dealt with:
  * kernel arguments.  The only step-dependent ones are Adam's two bias-correction scalars: the update is captured in its
    rfn_adam_step_multi_coef form, which reads them from a 2-float device tensor the wrapper fills before every replay.
  * the dropout seed (a kernel argument of every cell kernel): training-mode dropout > 0 and scheduled sampling are refused.
  * shapes: the batch is static; a replayed batch whose tensors do not have the captured shapes is refused (`copy_` would
    silently broadcast a smaller one).
  * the number of decoder steps (the reference breaks at the first all-zero label column, :274, so it varies with the longest
    caption of a batch): by default the step is captured with ALL `labels.size(1) - 1` decoder steps
    (`model.fixed_decoder_steps`), so a batch of shorter captions replays correctly -- the loader's masks are zero on the steps
    the reference would not have run (dataloader.py:312-314), they add exact zeros to the loss and to every gradient.  With
    `full_length=False` the capture keeps the example batch's own count and every replayed label tensor is checked against it
    (one small read-back per new label tensor; a mismatch raises).
  * hyper-parameters that are kernel arguments (betas, eps, weight_decay, grad_clip, reason_weight, the criterion's
    smoothing, train / eval mode, dropout, ss_prob): recorded at capture, compared before every replay, a change raises
    (only `lr` is re-read per step, through `coef`).
  * addresses: inputs are copied into static device buffers; the workspaces, flat gradient buffers and the loss live in the
    graph's private memory pool; `.grad` of every parameter is a view of those buffers after each replay, as in eager mode.
A captured step is bit-identical to the eager step that runs the SAME number of decoder steps (same kernels, same arguments;
tests/test_trainer_contract_gpu.py).  A `full_length=True` capture replayed on a batch the eager path would cut short (the break
at the first all-zero column) runs longer weight-gradient reductions than that eager step: equal to rounding, not to the bit.
"""
import torch

from . import _native as N


class GraphedTrainStep:
    def __init__(self, model, crit, opt, fc_feats, att_feats, labels, masks, top_words, reason_weight=1.0, warmup=3,
                 full_length=True):
        self.model, self.crit, self.opt, self.reason_weight = model, crit, opt, float(reason_weight)
        self._check_capturable()
        dev = fc_feats[0].device
        if labels.dim() != 2 or labels.size(1) < 2:
            raise N.RfnError('GraphedTrainStep: labels must be (batch, seq_length + 2)')
        own = model._decoder_steps(labels)
        if own > labels.size(1) - 1:
            raise N.RfnError('GraphedTrainStep: the last label column must be the all-zero END column (dataloader.py:300-314)')
        self.steps = labels.size(1) - 1 if full_length else own
        self.fc = [t.clone() for t in fc_feats]
        self.att = [t.clone() for t in att_feats]
        self.labels, self.masks, self.top = labels.clone(), masks.clone(), top_words.clone()
        self.coef = torch.zeros(2, device=dev)
        # eager warm-up on a side stream (first-use attribute calls, allocator, the cached decoder-step count); the run must
        # not train: parameters, moments and step count are put back afterwards
        snap = opt.snapshot()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        prev_fixed = model.fixed_decoder_steps
        model.fixed_decoder_steps = self.steps
        try:
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self._fill_coef()
                    self._body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            opt.restore(snap)
            self.graph = torch.cuda.CUDAGraph()
            count = opt.step_count
            with torch.cuda.graph(self.graph):
                self.loss = self._body()
            opt.step_count = count             # the capture ran opt.step() on the host without executing anything
        finally:
            model.fixed_decoder_steps = prev_fixed
        self.frozen = self._frozen_state()
        self._checked_labels = None

    def _check_capturable(self):
        model = self.model
        if model.grad_ready_hook is not None:
            raise N.RfnError('GraphedTrainStep: a grad_ready_hook (parallel.GradSync) cannot be combined with a captured step')
        if model.training and (model.drop_prob_lm > 0 or model.drop_prob_reason > 0 or model.drop_prob_fusion > 0):
            raise N.RfnError('GraphedTrainStep: dropout > 0 draws a fresh seed per step, a captured graph would freeze it')
        if model.ss_prob > 0:
            raise N.RfnError('GraphedTrainStep: scheduled sampling draws per step; not capturable')

    def _frozen_state(self):
        """Everything the captured launches hold as kernel arguments or as a choice of code path."""
        g0, m = self.opt.param_groups[0], self.model
        crit = {k: v for k, v in vars(self.crit).items() if isinstance(v, (int, float, bool, str, type(None)))}
        return dict(betas=tuple(g0['betas']), eps=g0['eps'], weight_decay=g0['weight_decay'], grad_clip=g0['grad_clip'],
                    reason_weight=self.reason_weight, training=m.training, ss_prob=m.ss_prob,
                    dropout=(m.drop_prob_lm, m.drop_prob_reason, m.drop_prob_fusion), gemm_flags=m.gemm_flags,
                    path_flags=int(getattr(m, 'path_flags', 0)),
                    dedup=int(m.dedup_seq_per_img), micro_batches=getattr(m, 'micro_batches', None), crit=crit)

    def _fill_coef(self):
        c0, c1 = self.opt.coefficients(self.opt.step_count + 1)
        self.coef[0].fill_(c0)                 # two scalar fills: stream-ordered, no host synchronisation
        self.coef[1].fill_(c1)

    def _body(self):
        self.opt.zero_grad()
        log_prob, top_pred = self.model(self.fc, self.att, self.labels)
        loss = self.crit(log_prob, self.labels[:, 1:], self.masks[:, 1:], top_pred, self.top, self.reason_weight)
        loss.backward()
        self.opt.step(coef_dev=self.coef)
        return loss

    def __call__(self, fc_feats=None, att_feats=None, labels=None, masks=None, top_words=None):
        """Replays the captured step on the given batch (None: the batch already in the static buffers) -> the loss tensor
        (a static buffer: read it before the next call)."""
        self._check_capturable()
        now = self._frozen_state()
        if now != self.frozen:
            diff = sorted(k for k in now if now[k] != self.frozen[k])
            raise N.RfnError('GraphedTrainStep: %s changed since the capture (%s -> %s); the captured launches hold the old '
                             'values -- build a new GraphedTrainStep' % (', '.join(diff), [self.frozen[k] for k in diff],
                                                                         [now[k] for k in diff]))
        if labels is not None:
            self._check_steps(labels)

        def put(dst, src, what):
            if tuple(src.shape) != tuple(dst.shape) or src.dtype != dst.dtype:
                raise N.RfnError('GraphedTrainStep: %s is %s %s, the captured step holds %s %s' %
                                 (what, tuple(src.shape), src.dtype, tuple(dst.shape), dst.dtype))
            dst.copy_(src, non_blocking=True)

        for name, dsts, srcs in (('fc_feats', self.fc, fc_feats), ('att_feats', self.att, att_feats)):
            if srcs is not None:
                if len(srcs) != len(dsts):
                    raise N.RfnError('GraphedTrainStep: %d %s, the captured step holds %d' % (len(srcs), name, len(dsts)))
                for i, (dst, src) in enumerate(zip(dsts, srcs)):
                    put(dst, src, '%s[%d]' % (name, i))
        for what, dst, src in (('labels', self.labels, labels), ('masks', self.masks, masks), ('top_words', self.top, top_words)):
            if src is not None:
                put(dst, src, what)
        self._fill_coef()
        self.graph.replay()
        self.opt.step_count += 1
        self.model._weights_epoch = getattr(self.model, '_weights_epoch', 0) + 1
        return self.loss

    def _check_steps(self, labels):
        """A batch the captured step count cannot serve is refused.  Full-length captures serve every loader batch (the END
        column is all zero); a capture at the example's own count serves batches of exactly that count."""
        if tuple(labels.shape) != tuple(self.labels.shape):
            raise N.RfnError('GraphedTrainStep: labels are %s, the captured step holds %s' % (tuple(labels.shape), tuple(self.labels.shape)))
        key = (labels.data_ptr(), labels._version)
        if self._checked_labels == key:
            return
        if self.steps == labels.size(1) - 1:
            # only the END column has to be checked, and only for tensors already on the host (a device tensor is not read
            # back: its last column is the loader's END column by contract)
            if not labels.is_cuda and bool((labels[:, -1] != 0).any()):
                raise N.RfnError('GraphedTrainStep: the last label column must be all zero (END)')
        else:
            prev = self.model.fixed_decoder_steps
            self.model.fixed_decoder_steps = None
            try:
                own = self.model._decoder_steps(labels)
            finally:
                self.model.fixed_decoder_steps = prev
            if own != self.steps:
                raise N.RfnError('GraphedTrainStep: this batch runs %d decoder steps, the step was captured with %d '
                                 '(full_length=False); capture with full_length=True to serve batches of any caption length'
                                 % (own, self.steps))
        self._checked_labels = key
