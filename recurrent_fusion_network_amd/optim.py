"""Fused clip_gradient + Adam (misc/utils.py:292-296 + train.py:69-71,162-163) on flat buffers.

The model keeps one flat gradient buffer per bucket (decoder, fusion core, one per encoder); ``FusedClampAdam``
re-points the parameters at flat buffers with the same layout, so the whole update -- clamp to +-grad_clip, L2
weight decay, Adam moments, bias correction -- is one rfn_adam_step launch per bucket (7 streams over 1.56 GB at
the headline config) instead of ~625 per-tensor updates, and a data-parallel run all-reduces M+2 buffers.
"""
import torch

from . import _native as N


class FusedClampAdam:
    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_clip=1.0):
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weight_decay, self.grad_clip = weight_decay, grad_clip
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
            self.flat[name] = dict(p=buf, m=torch.zeros_like(buf), v=torch.zeros_like(buf), n=total)

    def zero_grad(self):
        for p in self.model.parameters():
            p.grad = None
        self.model._last_flat_grads.clear()

    def set_lr(self, lr):
        self.lr = lr

    def step(self, grad_scale=1.0):
        """grad_scale multiplies the gradient before the clamp (1/world_size after a sum all-reduce)."""
        self.step_count += 1
        self.model._weights_epoch = getattr(self.model, '_weights_epoch', 0) + 1   # invalidates reuse_prefix entries
        for name, st in self.flat.items():
            g = self.model._last_flat_grads.get(name)
            if g is None:
                continue
            if g.numel() != st['n']:
                raise N.RfnError('flat gradient layout changed')
            N.check(N.lib.rfn_adam_step(st['p'].data_ptr(), g.data_ptr(), st['m'].data_ptr(), st['v'].data_ptr(),
                                        st['n'], self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                        self.grad_clip, grad_scale, self.step_count, N.stream_ptr()),
                    'rfn_adam_step')
