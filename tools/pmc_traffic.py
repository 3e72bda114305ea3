"""profiles/<dir>/report.json (tools/pmc_report.py) -> the `roofline.traffic` record bench.py reads:
memory-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) of the most expensive GEMM shape.
  python tools/pmc_traffic.py profiles/r2_pmc_final/report.json huge > profiles/r2_pmc/gemm_traffic_huge.json"""
import json
import sys

rep = json.load(open(sys.argv[1]))
arch = sys.argv[2] if len(sys.argv) > 2 else 'huge'
SHAPES = {'huge': dict(M=32768, N=5120, K=1280, tile=(256, 256)), 'base': dict(M=32768, N=3072, K=768, tile=(256, 256))}
sh = SHAPES[arch]
grid = -(-sh['M'] // sh['tile'][0]) * -(-sh['N'] // sh['tile'][1]) * 512          # threads: blocks x 512
rows = [r for r in rep if r['kernel'].startswith('gemm_f16x3_dma_kernel<256, 256') and r['grid'] == grid]
if not rows:
    raise SystemExit(f'no 256x256 GEMM dispatch with grid {grid} in {sys.argv[1]}')
d = rows[0]['derived']
alg = 4 * (sh['M'] * sh['K'] + sh['N'] * sh['K'] + sh['M'] * sh['N'])
print(json.dumps(dict(shape=f"lin1 M={sh['M']} N={sh['N']} K={sh['K']} (GELU, plane output), tile 256x256",
                      traffic_bytes_per_launch=int(d['read_bytes'] + d['write_bytes']), read_bytes=int(d['read_bytes']),
                      write_bytes=int(d['write_bytes']), algorithmic_bytes_per_launch=alg,
                      l2_hit_pct=round(d.get('l2_hit_pct', 0), 1), mfma_busy_pct=round(d.get('mfma_busy_pct', 0), 1),
                      source=sys.argv[1]), indent=1))
