"""MI355X-native implementation of the DeCafNet grounding hot path.

The directory name is not a valid Python identifier; import it with
``importlib.import_module('cvpr2025-decafnet_amd')``.

Sub-modules: ``modeling`` (libs/modeling drop-in), ``nms`` (libs/nms drop-in), ``evaluator``
(the Evaluator hot-path harness), ``config`` / ``synth`` (plain-dict opt tree, synthetic data),
``data`` (feature files, text-CLS table, annotation file -> per-video dicts), ``dropin`` (runs the reference's own
``eval.py`` / ``Evaluator`` on this package without editing it), ``loss`` (libs/modeling/loss.py forward values), ``dist`` (T-sharding over ranks), ``build`` (hipcc driver),
``_lib`` (ctypes binding of the C ABI).
"""
from . import config, synth  # noqa: F401


def __getattr__(name):
    # heavy sub-modules are imported on first use so that `config`/`synth` work without the .so
    if name in ('modeling', 'nms', 'evaluator', 'build', '_lib', 'dist', 'data', 'dropin', 'loss'):
        import importlib
        return importlib.import_module(f'{__name__}.{name}')
    raise AttributeError(name)
