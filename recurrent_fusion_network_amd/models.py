"""Model factory: the drop-in boundary of the reference (its models.py, lines 14-38).

``setup(opt)`` keeps the reference's contract -- build the captioner named by ``opt.caption_model`` and, when
``opt.start_from`` is set, resume from ``<start_from>/model_<load_model_id>.pth`` (same state_dict keys, so
checkpoints written by the reference load unchanged).  ``recurrent_fusion_model`` is the accelerated path;
``show_tell`` (BASELINE config 1) is the reference's simplest captioner on stock PyTorch modules for CPU plumbing
checks; every other name raises.
"""
import os

import torch

from .fusion_model import RecurrentFusionModel
from .show_tell import ShowTellModel

_REGISTRY = {'recurrent_fusion_model': RecurrentFusionModel, 'show_tell': ShowTellModel}


def _resume_paths(opt):
    """(checkpoint, infos) paths of a run to continue, or None.  The reference insists that the infos pickle
    exists next to the checkpoint even though the factory only reads the weights."""
    root = getattr(opt, 'start_from', None)
    if root is None:
        return None
    tag = opt.load_model_id
    return os.path.join(root, 'model_%s.pth' % tag), os.path.join(root, 'infos_%s.pkl' % tag)


def setup(opt):
    try:
        cls = _REGISTRY[opt.caption_model]
    except KeyError:
        raise Exception('Caption model not supported by the MI355X path: {}'.format(opt.caption_model))
    model = cls(opt)
    resume = _resume_paths(opt)
    if resume is not None:
        ckpt, infos = resume
        if not os.path.isdir(opt.start_from):
            raise AssertionError('%s must be a path' % opt.start_from)
        if not os.path.isfile(infos):
            raise AssertionError('infos pickle does not exist in path %s' % opt.start_from)
        model.load_state_dict(torch.load(ckpt, map_location='cpu'))
    return model
