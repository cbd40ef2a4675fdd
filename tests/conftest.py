import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_pkg():
    return importlib.import_module('cvpr2025-decafnet_amd')


class Golden:
    """npz fixture reader: tensors come back as torch tensors, JSON blobs as python objects."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))

    def __contains__(self, k):
        return k in self.z.files

    def keys(self):
        return self.z.files

    def t(self, k):
        return torch.from_numpy(np.array(self.z[k]))

    def js(self, k):
        return json.loads(bytes(self.z[k]).decode())

    def sub(self, prefix):
        """dict of tensors whose key starts with ``prefix`` (prefix stripped)."""
        return {k[len(prefix):]: self.t(k) for k in self.z.files if k.startswith(prefix)}


@pytest.fixture(scope='session')
def pkg():
    return load_pkg()


def has_gpu():
    try:
        return torch.cuda.is_available()
    except Exception:
        return False
