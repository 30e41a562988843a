"""MI355X-native recurrent-fusion caption decoder (hot path of cswhjiang/Recurrent_Fusion_Network).

Importing the package loads librfn_hip.so and raises if it is missing: the HIP library is the product,
there is no CPU or PyTorch-op fallback.
"""
from . import _native  # noqa: F401  (fails loudly when the library is absent)
from .fusion_model import RecurrentFusionModel  # noqa: F401
from .criteria import ReviewNetEnsembleCriterion, ReviewNetRewardCriterion, clip_gradient  # noqa: F401
from .optim import FusedClampAdam  # noqa: F401
from .models import setup  # noqa: F401
from .feeder import FeatureFeeder, read_image_features  # noqa: F401
from .eval_shim import eval_step, unique_image_rows  # noqa: F401
from .show_tell import ShowTellModel, LanguageModelCriterion  # noqa: F401
from .graphed import GraphedTrainStep  # noqa: F401

__all__ = ['RecurrentFusionModel', 'ReviewNetEnsembleCriterion', 'ReviewNetRewardCriterion', 'clip_gradient',
           'FusedClampAdam', 'setup', 'FeatureFeeder', 'read_image_features', 'eval_step', 'unique_image_rows', 'ShowTellModel',
           'LanguageModelCriterion', 'GraphedTrainStep']
