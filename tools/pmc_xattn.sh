#!/bin/bash
# HBM-side traffic of the cross-attention core at BASELINE config 2 (T=4096, E=1024, 16 heads, Lk=33, 8 queries per launch):
# two separate rocprofv3 --pmc passes over tools/xattn_bench.py -> gpurun_out/pmc_xattn.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcx_$c -- python3 $GRAFT_REPO_ROOT/tools/xattn_bench.py config2 > /tmp/pmcx_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 - <<'PY' > gpurun_out/pmc_xattn.json
import csv, glob, json
csv.field_size_limit(1 << 30)
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'/tmp/pmcx_{c}/*/*counter_collection.csv')[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == c and 'k_xattn_mfma' in r['Kernel_Name']]
    out[c] = {'launches': len(vals), 'kib_per_launch': sum(vals) / max(len(vals), 1)}
alg = 8.0 * 1024 * 8 * 4096 + 2 * 4.0 * 8 * 33 * 1024
fetch = 2.0 * 1024 * out['FETCH_SIZE']['kib_per_launch']
write = 1024.0 * out['WRITE_SIZE']['kib_per_launch']
print(json.dumps({'kernel': 'k_xattn_mfma<4,2,1,1> T=4096 E=1024 heads=16 Lk=33 B=8', 'launches': out['FETCH_SIZE']['launches'],
                  'fetch_bytes_per_launch_corrected_x2': fetch, 'write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write,
                  'algorithmic_bytes_per_launch': alg, 'ratio': (fetch + write) / alg,
                  'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 (gfx950), counters in KiB; cold + warm launches of tools/xattn_bench.py'}, indent=1))
PY
