"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: kernel, VGPRs, AGPRs, scratch, occupancy.
usage: hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
pat = r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)"
seen = set()
for m in re.finditer(pat, t, re.S):
    name = m.group(1)
    if name in seen:
        continue
    seen.add(name)
    try:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dem = name
    dem = dem.replace('dcf::', '').replace('(GemmBatch)', '').replace('void ', '')
    print(dem[:64].ljust(64), 'V', m.group(2).rjust(3), 'A', m.group(3).rjust(3), 'scratch', m.group(4).rjust(4), 'occ', m.group(5))
