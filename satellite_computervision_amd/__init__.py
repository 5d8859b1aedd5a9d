"""MI355X-native U-Net / ASPP tile pipeline (drop-in for the hot path of
mjevans26/Satellite_ComputerVision: utils/model_tools.py + utils/prediction_tools.py).

Importing this package loads the HIP C-ABI library (libsatcv.so); there is no CPU fallback.
"""
from . import _lib            # noqa: F401  (fails loudly if the HIP extension is missing)
from . import model_tools, prediction_tools, ops   # noqa: F401

__all__ = ['model_tools', 'prediction_tools', 'ops']
