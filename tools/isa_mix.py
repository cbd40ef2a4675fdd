"""one-letter instruction trace of a range of an ISA listing: python tools/isa_mix.py file.s first_line last_line
M mfma, d ds_read, D ds_write, w s_waitcnt, a v_accvgpr move, T transcendental, v other VALU, g global/LDS-DMA, B barrier, s SALU"""
import sys
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
seq = []
n = {}
for line in open(f).read().split('\n')[a - 1:b]:
    t = line.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.') or t[0].endswith(':'):
        continue
    op = t[0]
    if op.startswith('v_mfma'): c = 'M'
    elif op.startswith('ds_read'): c = 'd'
    elif op.startswith('ds_write'): c = 'D'
    elif op.startswith('s_waitcnt'): c = 'w'
    elif op.startswith('v_accvgpr'): c = 'a'
    elif op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_sqrt')): c = 'T'
    elif op.startswith('v_'): c = 'v'
    elif op.startswith(('global_', 'buffer_')): c = 'g'
    elif op.startswith('s_barrier'): c = 'B'
    elif op.startswith('s_nop'): c = 'n'
    elif op.startswith('s_'): c = 's'
    else: c = '?'
    seq.append(c)
    n[c] = n.get(c, 0) + 1
print(n)
print(''.join(seq))
