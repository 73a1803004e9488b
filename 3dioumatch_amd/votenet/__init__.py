"""Host-side mirror of the VoteNet-IoU callers of the hot path (reference models/*.py), used
by bench.py and the train-step tests.  The reference's own model files also run unchanged on
top of `3dioumatch_amd/dropin`."""
from .backbone import Pointnet2Backbone  # noqa: F401
from .config import DatasetConfig, scannet_config, sunrgbd_config  # noqa: F401
from .detector import VoteNet  # noqa: F401
from .heads import GridConv, ProposalModule, VotingModule  # noqa: F401
from .losses import get_labeled_loss  # noqa: F401
from .data import make_batch, make_semi_batch  # noqa: F401
from .step import (SemiSupervisedStep, SupervisedStep, update_ema_variables, lr_at,  # noqa: F401
                   bn_momentum_at)
from .eval_helper import APCalculator, parse_groundtruths, parse_predictions  # noqa: F401
from .eval_det import eval_det, eval_det_cls, voc_ap  # noqa: F401
