"""Print VGPR/AGPR/scratch/occupancy per kernel of one .hip file (compile-time check for spills)."""
import re, subprocess, sys, os
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rsprompter_amd.build import file_flags  # noqa: E402
cmd = ['/opt/rocm/bin/hipcc'] + file_flags(src) + ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
       f'-I{root}/include', '-c', src, '-o', '/tmp/_kr.o', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark: (?:\S+: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
    if not m:
        continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(' ')[0]] = v
for name, r in rows.items():
    print(f"{name[:110]:110s} V={r.get('VGPRs')} A={r.get('AGPRs')} scratch={r.get('ScratchSize')} occ={r.get('Occupancy')} lds={r.get('LDS')}")
