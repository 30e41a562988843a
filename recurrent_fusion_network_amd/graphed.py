"""One XE train step -- zero_grad, forward, criterion, backward, clamp + Adam (train.py:143-166) -- captured ONCE in a HIP
graph and replayed: for hosts whose launch path, not the device, paces a small configuration (BASELINE config 2 is 352
launches of 14 us: 4.84 ms of kernel time that a slow host stretches to 5.3-6.0 ms).  Opt-in; the eager path is the product's
default and the large configuration is device-bound either way.

What a capture freezes, and how each is dealt with:
  * kernel arguments.  The only step-dependent ones are Adam's two bias-correction scalars: the update is captured in its
    rfn_adam_step_multi_coef form, which reads them from a 2-float device tensor the wrapper fills before every replay.
  * the dropout seed (a kernel argument of every cell kernel): training-mode dropout > 0 and scheduled sampling are refused.
  * shapes and the number of decoder steps (the reference breaks at the first all-zero label column, :274): the batch is
    static; every batch replayed must have the captured shapes and decoder-step count (pad the labels as the loader does).
  * addresses: inputs are copied into static device buffers; the workspaces, flat gradient buffers and the loss live in the
    graph's private memory pool; `.grad` of every parameter is a view of those buffers after each replay, as in eager mode.
A captured step is bit-identical to the eager step (same kernels, same arguments; tests/test_trainer_contract_gpu.py).
"""
import torch

from . import _native as N


class GraphedTrainStep:
    def __init__(self, model, crit, opt, fc_feats, att_feats, labels, masks, top_words, reason_weight=1.0, warmup=3):
        if model.grad_ready_hook is not None:
            raise N.RfnError('GraphedTrainStep: a grad_ready_hook (parallel.GradSync) cannot be combined with a captured step')
        if model.training and (model.drop_prob_lm > 0 or model.drop_prob_reason > 0 or model.drop_prob_fusion > 0):
            raise N.RfnError('GraphedTrainStep: dropout > 0 draws a fresh seed per step, a captured graph would freeze it')
        if model.ss_prob > 0:
            raise N.RfnError('GraphedTrainStep: scheduled sampling draws per step; not capturable')
        self.model, self.crit, self.opt, self.reason_weight = model, crit, opt, float(reason_weight)
        dev = fc_feats[0].device
        self.fc = [t.clone() for t in fc_feats]
        self.att = [t.clone() for t in att_feats]
        self.labels, self.masks, self.top = labels.clone(), masks.clone(), top_words.clone()
        self.coef = torch.zeros(2, device=dev)
        # eager warm-up on a side stream (first-use attribute calls, allocator, the cached decoder-step count); the run must
        # not train: parameters, moments and step count are put back afterwards
        snap = opt.snapshot()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
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
        opt.step_count = count                 # the capture ran opt.step() on the host without executing anything

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
        if fc_feats is not None:
            for dst, src in zip(self.fc, fc_feats):
                dst.copy_(src, non_blocking=True)
        if att_feats is not None:
            for dst, src in zip(self.att, att_feats):
                dst.copy_(src, non_blocking=True)
        for dst, src in ((self.labels, labels), (self.masks, masks), (self.top, top_words)):
            if src is not None:
                dst.copy_(src, non_blocking=True)
        self._fill_coef()
        self.graph.replay()
        self.opt.step_count += 1
        self.model._weights_epoch = getattr(self.model, '_weights_epoch', 0) + 1
        return self.loss
