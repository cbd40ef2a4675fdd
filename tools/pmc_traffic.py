"""Summarise the two rocprofv3 --pmc passes of tools/pmc_traffic.sh: HBM-side bytes per launch of the GEMM kernels.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request and is doubled
(MI355X_MICROARCH.md, HBM section)."""
import csv, json, os, sys
csv.field_size_limit(1 << 30)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def family(name):
    """gemm_f16x3 / gemm_bf16s (= bf16x6) by the NTERMS template argument of the split kernels, gemm_f32 for the native one"""
    import re
    m = re.search(r'gemm_bf16s_kernel<\s*\d+,\s*\d+,\s*\d+,\s*\d+,\s*\d+,\s*(\d+)', name) or re.search(r'gemm_bf16s_kslice_kernel<\s*(\d+)', name)
    if m:
        return 'gemm_f16x3' if m.group(1) == '16' else 'gemm_bf16s'
    if 'k_ffn_chain<' in name or 'k_head_chain<' in name or 'k_dec_chain<' in name or 'k_enc_qkv' in name or 'k_enc_attn' in name:      # fc + proj (a head's trunk) as one f16x3 kernel: same family as in bench.py
        return 'gemm_f16x3'
    return 'gemm_f32' if 'gemm_f32' in name else None


def collect(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name']
        fam = family(n)
        if fam is None:
            continue
        d = per.setdefault(fam, {'launches': 0, 'kib': 0.0})
        d['launches'] += 1
        d['kib'] += float(r['Counter_Value'])
    return per


def last_forward(path, counter):
    """per-kernel counter sums (KiB) over the LAST forward of the run: dispatches from the last k_text_cls_norm (the first kernel of a
    forward, whether the scores ride on the vid_map GEMM or not) to the end of the trace"""
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    starts = [i for i, r in enumerate(rows) if 'k_text_cls_norm' in r['Kernel_Name']]
    if not starts:
        return {}, 0
    per = {}
    for r in rows[starts[-1]:]:
        import re
        n = re.sub(r'^void ', '', r['Kernel_Name'])
        n = re.sub(r'\(.*$', '', n).replace('dcf::', '')
        d = per.setdefault(n, {'launches': 0, 'kib': 0.0})
        d['launches'] += 1
        d['kib'] += float(r['Counter_Value'])
    return per, len(rows) - starts[-1]


f, w = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for fam in f:
    fl, wl = f[fam]['launches'], w.get(fam, {'launches': 0})['launches']
    fetch = 2.0 * 1024.0 * f[fam]['kib'] / max(fl, 1)
    write = 1024.0 * w.get(fam, {'kib': 0.0})['kib'] / max(wl, 1)
    out[fam] = {'launches_fetch_pass': fl, 'launches_write_pass': wl, 'fetch_bytes_per_launch_corrected_x2': fetch,
                'write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write}
out['note'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 2 '
               '--no-cpu-baseline --no-post`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per '
               '128-B request); counters in KiB; averages over every launch of the kernel family (tile kernel + k-sliced kernel; gemm_f16x3 = the '
               'default two-plane fp16 mode, gemm_bf16s = the bf16x6 instantiations, here the two vid_map products)')
# byte budget of one whole forward (every kernel family), when the run was single-stream (tools/pmc_traffic.sh passes
# --videos 1 --batch 8: eight videos per forward, one forward at a time)
ff, nf = last_forward(sys.argv[1], 'FETCH_SIZE')
fw_, nw = last_forward(sys.argv[2], 'WRITE_SIZE')
if ff and nf == nw:
    by = {}
    for n in ff:
        fetch = 2.0 * 1024.0 * ff[n]['kib']
        write = 1024.0 * fw_.get(n, {'kib': 0.0})['kib']
        by[n] = {'launches': ff[n]['launches'], 'fetch_bytes_x2': fetch, 'write_bytes': write, 'hbm_bytes': fetch + write}
    by = dict(sorted(by.items(), key=lambda kv: -kv[1]['hbm_bytes']))
    videos = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    out['forward'] = {'videos': videos, 'dispatches': nf, 'hbm_bytes': sum(v['hbm_bytes'] for v in by.values()), 'by_kernel': by,
                      'note': 'FETCH_SIZE x2 + WRITE_SIZE of every dispatch of the LAST forward of the run (from its first kernel, k_text_cls_norm, '
                              'to the end of the trace), summed by kernel; HBM / Infinity-Fabric side of L2, i.e. what the kernels miss in L2'}
import bench  # noqa: E402  (csrc_hash: bench.py only quotes this summary for the same kernel sources)
out['csrc_sha16'] = bench.csrc_hash()
print(json.dumps(out, indent=1))
