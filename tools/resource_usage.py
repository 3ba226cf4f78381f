#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) per kernel."""
import re
import subprocess
import sys

rows, cur = {}, None
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    m = re.search(r'remark: .*?(VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void tq::', '')
    print(f"{name:50s} vgpr={v.get('VGPRs', '?'):>4s} sgpr={v.get('SGPRs', '?'):>4s} "
          f"scratch={v.get('ScratchSize [bytes/lane]', '?'):>4s} occ={v.get('Occupancy [waves/SIMD]', '?'):>2s} "
          f"lds={v.get('LDS Size [bytes/block]', '?')}")
