#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles for gfx950 without a GPU).  Usage: python tools/kernel_resources.py drl-urban-planning_amd/csrc/edge.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
inc = src.rsplit('/', 1)[0]
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I', 'include', '-I', inc, '-c', src,
                      '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([\w ]+?)(?: \[[^\]]*\])?:\s+(\S+)', line.split('[-Rpass')[0])
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
print('%-5s %-5s %-5s %-8s %-6s %-6s %-4s %-7s %s' % ('VGPR', 'AGPR', 'SGPR', 'scratch', 'Sspill', 'Vspill', 'occ', 'LDS', 'kernel'))
for r in rows:
    short = re.sub(r'\(.*', '', r['name']).replace('upamd::', '')
    if flt and flt not in short:
        continue
    print('%-5s %-5s %-5s %-8s %-6s %-6s %-4s %-7s %s' % (r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', '?'),
                                                         r.get('ScratchSize', '?'), r.get('SGPRs Spill', '?'), r.get('VGPRs Spill', '?'),
                                                         r.get('Occupancy', '?'), r.get('LDS Size', '?'), short))
