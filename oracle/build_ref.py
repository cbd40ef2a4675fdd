"""Compile the REFERENCE's own NMS extension into oracle/_ref/ (test infrastructure).

The source is compiled where it lies (/root/reference/libs/nms/src/nms_cpu.cpp); nothing
is copied into this repository and oracle/_ref/ is git-ignored.  The recipe mirrors the
reference's libs/nms/setup_nms.py:7-19 (a torch CppExtension named ``nms_1d_cpu_vg``),
driven through torch.utils.cpp_extension.load so no setup.py of the reference is run.

Used to (a) validate oracle/nms_ref.c and (b) serve as ``cpu_baseline.kind ==
"reference"`` for the NMS stage in bench.py.  On the GPU box /root/reference does not
exist; the prebuilt .so under oracle/_ref/ travels with the snapshot instead.
"""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
SRC = '/root/reference/libs/nms/src/nms_cpu.cpp'


def prebuilt():
    c = sorted(glob.glob(os.path.join(OUT, 'nms_1d_cpu_vg*.so')))
    return c[0] if c else None


def build(verbose=False):
    if prebuilt():
        return prebuilt()
    if not os.path.exists(SRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name='nms_1d_cpu_vg', sources=[SRC], build_directory=OUT, extra_cflags=['-O3'],
         verbose=verbose, is_python_module=True)
    return prebuilt()


def load_module():
    """Import the compiled reference extension under a private name (so it never shadows the
    product's own ``nms_1d_cpu_vg`` package).  Returns None when it is not available."""
    so = prebuilt()
    if so is None:
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location('nms_1d_cpu_vg', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
