"""ISA gate of the shipped objects: no packed-fp32 instruction may route the HIGH dword of a source pair into its LOW lane
(`v_pk_*_f32 ... op_sel:[..1..]`).  That form produced sporadically wrong low lanes in lanes 48 - 63 inside the GEMM's LayerNorm
fold epilogue (profiles/r04_pkfma_hazard.md); `-fno-slp-vectorize` (build.py) keeps the compiler from generating it, this check
keeps it out whatever a future change or compiler does.  The broadcast in the other direction (`op_sel_hi:[..0..]`, the LOW dword
into the high lane) is what the explicit f32x4 arithmetic of the kernels compiles to and is exact.

Second rule (round 5): no instruction may mention M0 except `s_mov_b32 m0, sN` -- the LDS-DMA requests of the chain kernels set it and do not
restore it, which is only sound while nothing else in the objects depends on it.  Round 6 makes the rule prove that: (a) every `s_mov_b32 m0`
must be the first instruction of one of those requests (`s_nop 0` and `global_load_lds_dwordx4` follow it immediately: a write the COMPILER
emitted for a purpose of its own would mean it keeps a value in M0), and (b) no instruction that reads M0 IMPLICITLY -- relative moves,
`s_set_gpr_idx_*`, GWS / append / consume / ordered-count / addtid LDS operations, `s_sendmsg`, interpolation, `buffer_load ... lds` -- may
appear at all (a textual search for `m0` does not see those).

    python tools/isa_gate.py            # every object under cvpr2025-decafnet_amd/build/*.o; exit 1 on a hit
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PK = re.compile(r'(v_pk_[a-z]+_f32)\s+(\S+),\s*(\S+),\s*(\S+?)(?:,\s*(\S+?))?\s+.*?\bop_sel:\[([01,]+)\]')


IMPLICIT_M0 = re.compile(r'\b(v_movrel[a-z_0-9]*|s_movrel[a-z_0-9]*|s_set_gpr_idx_[a-z_0-9]*|ds_gws_[a-z_0-9]*|ds_append|ds_consume|ds_ordered_count|'
                         r'ds_read_addtid_b32|ds_write_addtid_b32|s_sendmsg[a-z]*|v_interp_[a-z_0-9]*)\b|\bbuffer_load_[a-z_0-9]+\b.*\blds\b')


def is_bad(line):
    """a packed fp32 instruction whose op_sel routes the high dword of a VECTOR-register pair into the low lane (a scalar-register
    pair as the selected source is a broadcast of a constant: score.hip has those, exact against the oracle)"""
    m = PK.search(line)
    if not m:
        return False
    srcs = [m.group(3), m.group(4), m.group(5)]
    sel = m.group(6).split(',')
    return any(b == '1' and i < len(srcs) and srcs[i] and srcs[i].startswith('v[') for i, b in enumerate(sel))


def device_disassembly(obj):
    """the gfx950 code object embedded in a host object (-fno-gpu-rdc build) -> text"""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), local)
        subprocess.run([OBJDUMP, '--offloading', local], cwd=tmp, check=True, capture_output=True)
        parts = [f for f in glob.glob(local + '.*') if 'amdgcn' in f]
        if not parts:
            raise RuntimeError(f'no device code object in {obj}')
        return subprocess.run([OBJDUMP, '-d', parts[0]], check=True, capture_output=True, text=True).stdout


def scan(objs=None):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, 'cvpr2025-decafnet_amd', 'build', '*.o')))
    report = {}
    for o in objs:
        txt = device_disassembly(o)
        packed = len(re.findall(r'v_pk_[a-z]+_f32\b', txt))
        hits = [l.strip() for l in txt.split('\n') if 'op_sel:[' in l and is_bad(l)]
        # M0: the chain kernels' LDS-DMA requests write it and do not restore it (glds16), so nothing else may read or expect it
        lines = txt.split('\n')
        hits += [l.strip() for l in lines if re.search(r'\bm0\b', l) and not re.search(r's_mov_b32\s+m0,\s*(s\d+|vcc_lo|vcc_hi)\b', l)]
        hits += [l.strip() for l in lines if IMPLICIT_M0.search(l)]
        code = [l for l in lines if re.match(r'\s+[a-z_0-9]+\b', l)]
        for i, l in enumerate(code):
            if re.search(r's_mov_b32\s+m0,', l):
                nxt = ' '.join(code[i + 1:i + 3])
                if not (re.search(r'\bs_nop\b', nxt) and re.search(r'\bglobal_load_lds_dwordx4\b', nxt)):
                    hits.append('M0 written outside an LDS-DMA request: ' + l.strip())
        report[os.path.relpath(o, ROOT)] = (packed, hits)
    return report


if __name__ == '__main__':
    rep = scan(sys.argv[1:] or None)
    bad = 0
    for o, (packed, hits) in rep.items():
        print(f'{o:58s} packed fp32 instructions {packed:6d}   low lane <- high dword (op_sel): {len(hits)}')
        for h in hits[:3]:
            print('      ', h[:140])
        bad += len(hits)
    sys.exit(1 if bad else 0)
