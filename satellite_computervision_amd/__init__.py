"""MI355X-native U-Net / ASPP tile pipeline (drop-in for the hot path of
mjevans26/Satellite_ComputerVision: utils/model_tools.py + utils/prediction_tools.py).

Importing this package loads the HIP C-ABI library (libsatcv.so); there is no CPU fallback.
(The only exception is `python -m satellite_computervision_amd.build`, which must be able to run
before the library exists or while it is stale.)
"""
import sys as _sys

_BUILDING = len(_sys.argv) > 0 and _sys.argv[0] == '-m' and any('satellite_computervision_amd.build' in a for a in getattr(_sys, 'orig_argv', []))

if not _BUILDING:
    from . import _lib            # noqa: F401  (fails loudly if the HIP extension is missing)
    from . import model_tools, prediction_tools, ops   # noqa: F401

__all__ = ['model_tools', 'prediction_tools', 'ops']
