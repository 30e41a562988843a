"""Criteria of the reference's misc/utils.py, same constructor / call signatures, HIP kernels inside.

* ReviewNetEnsembleCriterion (misc/utils.py:153-192): masked NLL (optionally label-smoothed) / batch
  + reason_weight / (M+1) * sum_j MultiLabelMarginLoss(top_pred[j], top_true)  -> rfn_xe_loss +
  rfn_multilabel_margin, fixed summation order.
* ReviewNetRewardCriterion (misc/utils.py:44-84): REINFORCE / PPO-clip term + entropy regulariser
  (rfn_rl_loss) + the same reason loss (rfn_multilabel_margin).
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
        scratch = torch.empty(max(B * T, len(preds) * B), device=dev)
        st = N.stream_ptr()
        N.check(N.lib.rfn_xe_loss(log_prob.data_ptr(), B, T, V1, target.data_ptr(), target.stride(0),
                                  mask.data_ptr(), mask.stride(0), eps, 1.0, scratch.data_ptr(), loss.data_ptr(), 0,
                                  None, st), 'rfn_xe_loss')
        scale = reason_weight / len(preds)
        # the M+1 reasoning heads in one launch; their losses are added to the language loss head by head
        N.check(N.lib.rfn_multilabel_margin_grouped(len(preds), N.ptr_array(preds), B, K, top_true.data_ptr(), scale, 1.0,
                                                    None, scratch.data_ptr(), loss.data_ptr(), 1, None, st),
                'rfn_multilabel_margin_grouped')
        ctx.eps, ctx.scale, ctx.shape = eps, scale, (B, T, V1, K)
        ctx.save_for_backward(log_prob, target, mask, top_true, *preds)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        log_prob, target, mask, top_true, *preds = ctx.saved_tensors
        B, T, V1, K = ctx.shape
        st = N.stream_ptr()
        g = g.contiguous().float()                       # d loss stays on the device: the kernels read it there
        dlogp = torch.empty_like(log_prob)
        N.check(N.lib.rfn_xe_loss_ex(log_prob.data_ptr(), B, T, V1, target.data_ptr(), target.stride(0),
                                     mask.data_ptr(), mask.stride(0), ctx.eps, 1.0, g.data_ptr(), None, None, 0,
                                     dlogp.data_ptr(), st), 'rfn_xe_loss_ex (grad)')
        dpreds = [torch.empty_like(p) for p in preds]
        N.check(N.lib.rfn_multilabel_margin_grouped(len(preds), N.ptr_array(preds), B, K, top_true.data_ptr(),
                                                    ctx.scale, 1.0, g.data_ptr(), None, None, 0, N.ptr_array(dpreds),
                                                    st), 'rfn_multilabel_margin_grouped (grad)')
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
        scratch = torch.empty(len(preds) * B, device=dev)
        N.check(N.lib.rfn_multilabel_margin_grouped(len(preds), N.ptr_array(preds), B, K, top_true.data_ptr(), scale, 1.0,
                                                    None, scratch.data_ptr(), loss.data_ptr(), 0, None,
                                                    N.stream_ptr()), 'rfn_multilabel_margin_grouped')
        ctx.scale = scale
        ctx.save_for_backward(top_true, *preds)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        top_true, *preds = ctx.saved_tensors
        B, K = preds[0].shape
        g = g.contiguous().float()
        out = [torch.empty_like(p) for p in preds]
        N.check(N.lib.rfn_multilabel_margin_grouped(len(preds), N.ptr_array(preds), B, K, top_true.data_ptr(),
                                                    ctx.scale, 1.0, g.data_ptr(), None, None, 0, N.ptr_array(out),
                                                    N.stream_ptr()), 'rfn_multilabel_margin_grouped (grad)')
        return (None, None) + tuple(out)


class _RLFn(torch.autograd.Function):
    """Policy + entropy terms of ReviewNetRewardCriterion (misc/utils.py:50-72): rfn_rl_loss."""

    @staticmethod
    def forward(ctx, seq, reward, entropy_reg, old_lp, use_ppo, ppo_clip, inp, logprobs_all):
        inp = N.require_cuda_f32(inp, 'input')
        lp = N.require_cuda_f32(logprobs_all, 'logprobs_all')
        B, T = inp.shape
        V1 = lp.size(2)
        if lp.size(1) < T:
            raise N.RfnError('logprobs_all has fewer steps than the sampled sequence')
        dev = inp.device
        seq = seq.to(dev).long().contiguous()
        reward = reward.to(dev).float().contiguous()
        old = None if old_lp is None else old_lp.to(dev).float().contiguous()
        loss = torch.zeros(1, device=dev)
        scratch = torch.empty(B * T, device=dev)
        N.check(N.lib.rfn_rl_loss(inp.data_ptr(), T, seq.data_ptr(), seq.stride(0), reward.data_ptr(), reward.stride(0),
                                  lp.data_ptr(), lp.stride(0), lp.stride(1), B, T, V1, float(entropy_reg), N.ptr(old),
                                  T, int(bool(use_ppo)), float(ppo_clip), scratch.data_ptr(), loss.data_ptr(), 0, None,
                                  0, None, 0, 0, N.stream_ptr()), 'rfn_rl_loss')
        ctx.args = (float(entropy_reg), int(bool(use_ppo)), float(ppo_clip), B, T, V1)
        ctx.save_for_backward(inp, lp, seq, reward, *(() if old is None else (old,)))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        inp, lp, seq, reward, *rest = ctx.saved_tensors
        old = rest[0] if rest else None
        ent, ppo, clip, B, T, V1 = ctx.args
        d_inp = torch.empty_like(inp)
        d_lp = torch.empty_like(lp)
        g = g.contiguous().float()                       # d loss stays on the device: the kernel reads it there
        # one launch writes every row of d_lp (rows t >= T do not enter the loss: zeros) already scaled by g
        N.check(N.lib.rfn_rl_loss_ex(inp.data_ptr(), T, seq.data_ptr(), seq.stride(0), reward.data_ptr(),
                                     reward.stride(0), lp.data_ptr(), lp.stride(0), lp.stride(1), B, T, lp.size(1), V1,
                                     ent, N.ptr(old), T, ppo, clip, g.data_ptr(), None, None, 0, d_inp.data_ptr(), T,
                                     d_lp.data_ptr(), d_lp.stride(0), d_lp.stride(1), N.stream_ptr()),
                'rfn_rl_loss_ex (grad)')
        return None, None, None, None, None, None, d_inp, d_lp


class ReviewNetRewardCriterion(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.use_label_smoothing = opt.use_label_smoothing
        self.label_smoothing_epsilon = opt.label_smoothing_epsilon

    def forward(self, input, seq, reward, logprobs_all, entropy_reg, top_pred, top_true, reason_weight,
                sample_logprobs_old, opt):
        use_ppo = getattr(opt, 'use_ppo', 0)
        out = _RLFn.apply(seq, reward, entropy_reg, sample_logprobs_old if use_ppo else None, use_ppo,
                          getattr(opt, 'ppo_clip', 0.2), input, logprobs_all)
        preds = top_pred if isinstance(top_pred, (list, tuple)) else [top_pred]
        return out + _MLMFn.apply(float(reason_weight) / len(preds), top_true, *preds)


def clip_gradient(optimizer, grad_clip):
    """misc/utils.py:292-296 -- kept for unchanged training scripts that use a torch optimizer."""
    for group in optimizer.param_groups:
        for param in group['params']:
            if param.grad is not None:
                param.grad.data.clamp_(-grad_clip, grad_clip)
