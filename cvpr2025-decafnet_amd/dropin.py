"""Run the reference's own ``eval.py`` / ``libs.worker_v2.Evaluator`` on this package WITHOUT editing a file of it.

The reference binds the hot path by module path: ``create_model`` does ``from .modeling.model import
PtTransformerEarlyFusionIterative`` (libs/worker_v2.py:191-193), ``libs/modeling/__init__.py`` does ``from .model import
PtTransformer, PtGenerator``, ``libs/nms/__init__.py`` does ``from .nms import batched_nms`` and ``libs/nms/nms.py`` imports
the extension ``nms_1d_cpu_vg``.  ``install()`` puts this package's classes into ``sys.modules`` under exactly those module
paths BEFORE the reference is imported, so the import system hands them out and the reference's loss / optimiser /
data code keeps loading from its own tree:

    import importlib, sys
    importlib.import_module('cvpr2025-decafnet_amd').dropin.install()
    sys.path.insert(0, '/path/to/CVPR2025-DeCafNet')
    import eval            # or: from libs.worker_v2 import Evaluator

(or ``python -c "...dropin.install(); import runpy; runpy.run_path('eval.py', run_name='__main__')" --name ...``).
"""
from __future__ import annotations

import sys
import types

MODEL_MODULE = 'libs.modeling.model'
NMS_MODULE = 'libs.nms.nms'
EXT_MODULE = 'nms_1d_cpu_vg'


def install():
    """Register the MI355X classes under the reference's module paths.  Idempotent; must run before ``libs`` is imported."""
    from . import modeling, nms
    for name in (MODEL_MODULE, NMS_MODULE):
        if name in sys.modules and not getattr(sys.modules[name], '__decafnet_hip__', False):
            raise RuntimeError(f'{name} is already imported from the reference: call dropin.install() first')
    model_mod = types.ModuleType(MODEL_MODULE, 'MI355X drop-in for libs/modeling/model.py (cvpr2025-decafnet_amd.modeling)')
    model_mod.PtTransformerEarlyFusionIterative = modeling.PtTransformerEarlyFusionIterative
    model_mod.PtTransformer = modeling.PtTransformer
    model_mod.PtTransformerEarlyFusion = modeling.PtTransformerEarlyFusion
    model_mod.PtGenerator = modeling.PtGenerator
    model_mod.__decafnet_hip__ = True
    nms_mod = types.ModuleType(NMS_MODULE, 'MI355X drop-in for libs/nms/nms.py (cvpr2025-decafnet_amd.nms)')
    nms_mod.batched_nms = nms.batched_nms
    nms_mod.segment_voting = nms.segment_voting
    nms_mod.__decafnet_hip__ = True
    ext_mod = types.ModuleType(EXT_MODULE, 'MI355X drop-in for the extension module libs/nms/src/nms_cpu.cpp')
    ext_mod.nms = nms.nms
    ext_mod.softnms = nms.softnms
    ext_mod.__decafnet_hip__ = True
    sys.modules[MODEL_MODULE] = model_mod
    sys.modules[NMS_MODULE] = nms_mod
    sys.modules.setdefault(EXT_MODULE, ext_mod)
    return model_mod, nms_mod
