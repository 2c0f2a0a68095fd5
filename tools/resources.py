#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `make -C scale-letkf_amd resources F=<unit>.hip` output (stdin or file)."""
import re, subprocess, sys
cur = None; rows = []
for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    m = re.search(r':\d+:\d+:(?: remark:)?\s+(.*?) \[-Rpass', l)
    if not m: continue
    s = m.group(1).strip()
    if s.startswith('Function Name') or s.startswith('Name:'):
        cur = {'name': s.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in s:
        a, b = s.split(':', 1); cur[a.strip()] = b.strip()
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip().replace('letkf::', '')
    print(n[:64].ljust(64), 'VGPR', r.get('VGPRs'), 'AGPR', r.get('AGPRs'), 'scratch', r.get('ScratchSize [bytes/lane]'), 'occ', r.get('Occupancy [waves/SIMD]'),
          'sgpr-spill', r.get('SGPRs Spill'), 'vgpr-spill', r.get('VGPRs Spill'), 'LDS', r.get('LDS Size [bytes/block]'))
