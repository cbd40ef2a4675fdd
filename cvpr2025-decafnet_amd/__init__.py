"""MI355X-native implementation of the DeCafNet grounding hot path.

The directory name is not a valid Python identifier; import it with
``importlib.import_module('cvpr2025-decafnet_amd')``.
"""
from . import config, synth  # noqa: F401
