#!/usr/bin/env python3
"""VGPR/AGPR/occupancy/scratch per kernel of bde_api.hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
                      '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/_kr.so',
                      os.path.join(repo, 'bde2vid_amd/csrc/bde_api.hip')], capture_output=True, text=True).stderr
cur, d = None, {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); d[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z][A-Za-z /\[\]]*?):\s+(\d+)', line)
    if m and cur:
        d[cur][m.group(1).strip()] = int(m.group(2))
for k, v in d.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void bde::', '')
    print(f"{name:50s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} occ={v.get('Occupancy [waves/SIMD]')} "
          f"scratch={v.get('ScratchSize [bytes/lane]')} sgpr={v.get('TotalSGPRs')} lds={v.get('LDS Size [bytes/block]')}")
