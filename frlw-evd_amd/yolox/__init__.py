"""YOLOX-style detector of the FRLW-EvD hot path.

Module and parameter names equal the reference's (``core/yolox/models/*``, ``core/model.py``) so that
``state_dict`` checkpoints interchange (``core/exp.py:198-210``).  The modules' eager ``forward`` is the
plain-PyTorch fp32 definition of the network (used for training and as the numerics reference); in
eval mode on a ROCm device :class:`frlw_evd_amd.detector.DetectorEngine` runs the same network with the
hand-written gfx950 kernels.
"""
from .network_blocks import BaseConv, Bottleneck, CSPLayer, Focus, SPPBottleneck, SiLU, get_activation  # noqa: F401
from .darknet import CSPDarknet
from .yolo_pafpn import YOLOPAFPN
from .yolo_head import YOLOXHead
from .model import model, build_yolox
from .bfm import Temporal_Active_Focus_connect

__all__ = ["BaseConv", "Bottleneck", "CSPLayer", "Focus", "SPPBottleneck", "SiLU", "get_activation",
           "CSPDarknet", "YOLOPAFPN", "YOLOXHead", "model", "build_yolox", "Temporal_Active_Focus_connect"]
