"""Model factory: the drop-in boundary of the reference (models.py:14-38).

``setup(opt)`` returns the HIP-backed module for ``caption_model == 'recurrent_fusion_model'`` and resumes
from ``model_<id>.pth`` exactly like the reference (same state_dict keys).  Other caption models are outside
the accelerated path and raise.
"""
import os

import torch

from .fusion_model import RecurrentFusionModel


def setup(opt):
    if opt.caption_model != 'recurrent_fusion_model':
        raise Exception("Caption model not supported by the MI355X path: {}".format(opt.caption_model))
    model = RecurrentFusionModel(opt)
    if vars(opt).get('start_from', None) is not None:
        assert os.path.isdir(opt.start_from), " %s must be a a path" % opt.start_from
        infos = os.path.join(opt.start_from, "infos_" + opt.load_model_id + ".pkl")
        assert os.path.isfile(infos), "infos.pkl file does not exist in path %s" % opt.start_from
        state = torch.load(os.path.join(opt.start_from, 'model_' + opt.load_model_id + '.pth'), map_location='cpu')
        model.load_state_dict(state)
    return model
