"""postproc.hip expf_glibc restates glibc's expf (2.27+: ARM optimized-routines) operation for operation.  This script checks the restatement
-- in exact rational arithmetic for the fused multiply-adds -- against the C library of the machine it runs on: python tools/expf_model_check.py [n]
(build container, 200 000 arguments in (-40, 0]: 0 mismatches with the three multiply-adds fused, 0 without)."""
import ctypes
import random
import struct
import sys
from fractions import Fraction

libm = ctypes.CDLL('libm.so.6')
libm.expf.restype = ctypes.c_float
libm.expf.argtypes = [ctypes.c_float]
TAB = [0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238,
       0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82,
       0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db,
       0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
       0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540]


def f32(x):
    return struct.unpack('<f', struct.pack('<f', x))[0]


def bits64(d):
    return struct.unpack('<Q', struct.pack('<d', d))[0]


def from64(b):
    return struct.unpack('<d', struct.pack('<Q', b & ((1 << 64) - 1)))[0]


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


INV = float.fromhex('0x1.71547652b82fep+0') * 32
SHIFT = float.fromhex('0x1.8p+52')
C0, C1, C2 = (float.fromhex('0x1.c6af84b912394p-5') / 32 / 32 / 32, float.fromhex('0x1.ebfce50fac4f3p-3') / 32 / 32,
              float.fromhex('0x1.62e42ff0c52d6p-1') / 32)


def expf_model(x, fused):
    z = INV * float(x)
    kd = z + SHIFT
    ki = bits64(kd)
    kd -= SHIFT
    r = z - kd
    s = from64(TAB[ki % 32] + ((ki << 47) & ((1 << 64) - 1)))
    if fused:
        y = fma(fma(C0, r, C1), r * r, fma(C2, r, 1.0))
    else:
        y = (C0 * r + C1) * (r * r) + (C2 * r + 1.0)
    return f32(y * s)


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    random.seed(1)
    bad = [0, 0]
    for _ in range(n):
        x = f32(-random.random() * random.choice([0.01, 0.1, 1, 2, 4, 10, 40]))
        ref = libm.expf(x)
        bad[0] += expf_model(x, True) != ref
        bad[1] += expf_model(x, False) != ref
    print(f'{n} arguments: mismatches with libm expf: fused model {bad[0]}, unfused model {bad[1]}')
