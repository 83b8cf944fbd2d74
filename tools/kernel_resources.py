#!/usr/bin/env python3
"""VGPR/AGPR/occupancy/scratch per kernel of the three translation units (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [name filter]"""
import re, subprocess, sys, os
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = ''
for tu in ('bde_api.hip', 'conv_tu.hip', 'sb_tu.hip'):
    out += subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-DBDE_BUILD', '--cuda-device-only', '-c',
                           '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/_kr.o',
                           os.path.join(repo, 'bde2vid_amd/csrc', tu)], capture_output=True, text=True).stderr
cur, d = None, {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); d[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z][A-Za-z /\[\]]*?):\s+(\d+)', line)
    if m and cur:
        d[cur][m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for k, v in d.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void bde::', '')
    if flt not in name:
        continue
    print(f"{name:50s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} occ={v.get('Occupancy [waves/SIMD]')} "
          f"scratch={v.get('ScratchSize [bytes/lane]')} sgpr={v.get('TotalSGPRs')} lds={v.get('LDS Size [bytes/block]')}")
