"""Criteria of the reference's misc/utils.py, same constructor / call signatures, HIP kernels inside.

* ReviewNetEnsembleCriterion (misc/utils.py:153-192): masked NLL (optionally label-smoothed) / batch
  + reason_weight / (M+1) * sum_j MultiLabelMarginLoss(top_pred[j], top_true)  -> rfn_xe_loss +
  rfn_multilabel_margin, fixed summation order.
* ReviewNetRewardCriterion (misc/utils.py:44-84): REINFORCE / PPO-clip term + entropy regulariser + the
  same reason loss.  The (B,T) element-wise policy term is host glue on torch tensors (it is not on the
  timed path); the reason loss uses the HIP kernel.
* clip_gradient (misc/utils.py:292-296): element-wise clamp; FusedClampAdam fuses it into the update.
"""
import torch
import torch.nn as nn

from . import _native as N


class _XEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eps, reason_weight, target, mask, top_true, log_prob, *top_pred):
        log_prob = N.require_cuda_f32(log_prob, 'log_prob')
        B, T, V1 = log_prob.shape
        dev = log_prob.device
        target = target.to(dev)
        if target.dtype != torch.long:
            target = target.long()
        mask = mask.to(dev).float()
        top_true = top_true.to(dev).long().contiguous()
        if target.size(1) < T or mask.size(1) < T:
            raise N.RfnError('target / mask have fewer columns than log_prob has steps')
        if target.stride(1) != 1:
            target = target.contiguous()
        if mask.stride(1) != 1:
            mask = mask.contiguous()
        preds = [N.require_cuda_f32(p, 'top_pred') for p in top_pred]
        K = preds[0].size(1)
        loss = torch.zeros(1, device=dev)
        scratch = torch.empty(max(B * T, B), device=dev)
        st = N.stream_ptr()
        N.check(N.lib.rfn_xe_loss(log_prob.data_ptr(), B, T, V1, target.data_ptr(), target.stride(0),
                                  mask.data_ptr(), mask.stride(0), eps, 1.0, scratch.data_ptr(), loss.data_ptr(), 0,
                                  None, st), 'rfn_xe_loss')
        scale = reason_weight / len(preds)
        for p in preds:
            N.check(N.lib.rfn_multilabel_margin(p.data_ptr(), B, K, top_true.data_ptr(), scale, 1.0,
                                                scratch.data_ptr(), loss.data_ptr(), 1, None, st),
                    'rfn_multilabel_margin')
        ctx.eps, ctx.scale, ctx.shape = eps, scale, (B, T, V1, K)
        ctx.save_for_backward(log_prob, target, mask, top_true, *preds)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        log_prob, target, mask, top_true, *preds = ctx.saved_tensors
        B, T, V1, K = ctx.shape
        st = N.stream_ptr()
        dlogp = torch.empty_like(log_prob)
        N.check(N.lib.rfn_xe_loss(log_prob.data_ptr(), B, T, V1, target.data_ptr(), target.stride(0),
                                  mask.data_ptr(), mask.stride(0), ctx.eps, 1.0, None, None, 0, dlogp.data_ptr(),
                                  st), 'rfn_xe_loss (grad)')
        dlogp.mul_(g)
        dpreds = []
        for p in preds:
            dp = torch.empty_like(p)
            N.check(N.lib.rfn_multilabel_margin(p.data_ptr(), B, K, top_true.data_ptr(), ctx.scale, 1.0, None, None,
                                                0, dp.data_ptr(), st), 'rfn_multilabel_margin (grad)')
            dpreds.append(dp.mul_(g))
        return (None, None, None, None, None, dlogp) + tuple(dpreds)


class ReviewNetEnsembleCriterion(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.use_label_smoothing = opt.use_label_smoothing
        self.label_smoothing_epsilon = opt.label_smoothing_epsilon
        self.use_cuda = getattr(opt, 'use_cuda', 1)

    def forward(self, log_prob, target, mask, top_pred, top_true, reason_weight):
        eps = float(self.label_smoothing_epsilon) if self.use_label_smoothing else 0.0
        return _XEFn.apply(eps, float(reason_weight), target, mask, top_true, log_prob, *top_pred)


class _MLMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, top_true, *top_pred):
        preds = [N.require_cuda_f32(p, 'top_pred') for p in top_pred]
        B, K = preds[0].shape
        dev = preds[0].device
        top_true = top_true.to(dev).long().contiguous()
        loss = torch.zeros(1, device=dev)
        scratch = torch.empty(B, device=dev)
        for j, p in enumerate(preds):
            N.check(N.lib.rfn_multilabel_margin(p.data_ptr(), B, K, top_true.data_ptr(), scale, 1.0,
                                                scratch.data_ptr(), loss.data_ptr(), int(j > 0), None,
                                                N.stream_ptr()), 'rfn_multilabel_margin')
        ctx.scale = scale
        ctx.save_for_backward(top_true, *preds)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        top_true, *preds = ctx.saved_tensors
        out = []
        for p in preds:
            B, K = p.shape
            dp = torch.empty_like(p)
            N.check(N.lib.rfn_multilabel_margin(p.data_ptr(), B, K, top_true.data_ptr(), ctx.scale, 1.0, None, None,
                                                0, dp.data_ptr(), N.stream_ptr()), 'rfn_multilabel_margin (grad)')
            out.append(dp.mul_(g))
        return (None, None) + tuple(out)


class ReviewNetRewardCriterion(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.use_label_smoothing = opt.use_label_smoothing
        self.label_smoothing_epsilon = opt.label_smoothing_epsilon

    def forward(self, input, seq, reward, logprobs_all, entropy_reg, top_pred, top_true, reason_weight,
                sample_logprobs_old, opt):
        B, T = input.shape
        inp = input.contiguous().view(-1)
        reward = reward.to(inp.device).contiguous().view(-1)
        mask_0 = (seq > 0).float()
        mask = torch.cat([mask_0.new_ones(B, 1), mask_0[:, :-1]], 1).view(-1)
        lp = logprobs_all[:, :T, :]
        entropy_minus = (lp * torch.exp(lp)).sum(2) * mask_0
        if getattr(opt, 'use_ppo', 0):
            ratio = torch.exp(inp) / (1e-5 + torch.exp(sample_logprobs_old.contiguous().view(-1)))
            surr1 = ratio * reward
            surr2 = surr1.clamp(1 - opt.ppo_clip, 1 + opt.ppo_clip) * reward
            out = -torch.min(surr1, surr2) * mask
        else:
            out = -inp * reward * mask
        out = out.sum() / B + entropy_reg * entropy_minus.sum() / B
        preds = top_pred if isinstance(top_pred, (list, tuple)) else [top_pred]
        return out + _MLMFn.apply(float(reason_weight) / len(preds), top_true, *preds)


def clip_gradient(optimizer, grad_clip):
    """misc/utils.py:292-296 -- kept for unchanged training scripts that use a torch optimizer."""
    for group in optimizer.param_groups:
        for param in group['params']:
            if param.grad is not None:
                param.grad.data.clamp_(-grad_clip, grad_clip)
