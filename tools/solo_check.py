"""Reproduce bench.py's roofline.achieved from a rocprofv3 --kernel-trace --stats summary of the SAME command.

    python tools/solo_check.py <kernel_stats.csv> <bench.json>

The CSV aggregates every launch of the process: setup forwards (6), warm-up (W), the timed steps (K) and the K steps
re-run under HIP events, all of them the same forward on one stream.  The algorithmic GEMM flops of one forward come from
the JSON (roofline.alg_flops_per_launch x launches / steps); sum(2MNK) = that x the number of forwards; sum(duration) =
TotalDurationNs of the GEMM kernel rows.  The text encoder's few small GEMM launches (setup only) are in the duration sum
and not in the flop sum (< 0.1 %)."""
import csv
import json
import re
import sys


def is_gemm(name):
    # k_ffn_chain / k_ffn_pair / k_head_chain: fc + proj (resp. a head's two trunk convolutions) as one kernel, profiled as gemm_f16x3<ffn_chain> / <head_chain>
    return re.search(r'gemm_bf16s_(kslice_)?kernel<|gemm_f32_kernel<|k_ffn_chain<|k_ffn_pair<|k_head_chain<|k_dec_chain<|k_enc_qkv|k_enc_attn', name) is not None


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    rf = j['roofline']
    K, W = j['steps'], j['warmup']
    forwards = 6 + W + K * j.get('timed_blocks', 1) + K
    flops_fwd = rf['alg_flops_per_launch'] * rf['launches'] / K
    gem = [r for r in rows if is_gemm(r['Name'])]
    dur_s = sum(int(r['TotalDurationNs']) for r in gem) * 1e-9
    calls = sum(int(r['Calls']) for r in gem)
    all_s = sum(int(r['TotalDurationNs']) for r in rows) * 1e-9
    ach = flops_fwd * forwards / dur_s / 1e12
    print(f'forwards in the process          : {forwards} (6 setup + {W} warm-up + {K * j.get("timed_blocks", 1)} timed + {K} event-profiled)')
    print(f'GEMM rows                        : {len(gem)} kernels, {calls} calls ({calls / forwards:.1f} per forward), {dur_s * 1e3:.1f} ms')
    print(f'sum(2MNK) / sum(duration)        : {ach:.1f} TFLOP/s  = {ach / rf["peak"]:.3f} of {rf["peak"]:.0f}')
    print(f'bench JSON roofline.achieved     : {rf["achieved"]:.1f} TFLOP/s  = {rf["frac"]:.3f}   (HIP events)')
    print(f'ratio rocprof / events           : {ach / rf["achieved"]:.3f}')
    print(f'GEMM share of kernel time (CSV)  : {dur_s / all_s:.3f}   (JSON: {1 - rf["non_gemm_share"]:.3f})')
    print(f'kernel time per forward (CSV)    : {all_s / forwards * 1e3:.3f} ms   (JSON event_ms_per_step {j.get("event_ms_per_step", 0):.3f}, ms_per_step {j["ms_per_step"]:.3f})')
    for r in sorted(gem, key=lambda r: -int(r['TotalDurationNs']))[:12]:
        print(f'  {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:110]}')


if __name__ == '__main__':
    main()
