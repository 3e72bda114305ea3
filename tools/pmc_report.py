"""Join the rocprofv3 counter CSVs of tools/pmc_round2.sh into one table per kernel instance.

  python tools/pmc_report.py gpurun_out/pmc_<tag> [--json out.json]

Rows are keyed by (kernel short name, grid size); a counter's value is the mean over that key's dispatches.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over SIMDs (32 per 32x32x16 f16 MFMA), SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE are
summed over the 8 XCDs (x 4 SEs for SQ_BUSY_CYCLES), FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE x2 on gfx950)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_]+)(<[^(]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def load(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in sorted(glob.glob(os.path.join(root, '*', '*counter_collection.csv'))):
        for r in csv.DictReader(open(f)):
            n = short(r['Kernel_Name'])
            if not any(t in n for t in ('gemm', 'attn', 'layernorm', 'relpos', 'split', 'sam_', 'upscale', 'rpn_', 'nms_')):
                continue
            key = (n, int(r['Grid_Size']))
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[key] = dict(vgpr=int(r['VGPR_Count']), agpr=int(r['Accum_VGPR_Count']), lds=int(r['LDS_Block_Size']),
                             wg=int(r['Workgroup_Size']))
    out = {}
    for key, cs in agg.items():
        row = {c: sum(v) / len(v) for c, v in cs.items()}
        row.update(meta[key])
        out[key] = row
    return out


def derive(row):
    d = {}
    g = row.get('GRBM_GUI_ACTIVE')
    if g:
        wall = g / 8.0                                     # cycles (one XCD's view of the dispatch)
        d['wall_cycles'] = wall
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in row:
            d['mfma_busy_pct'] = 100.0 * row['SQ_VALU_MFMA_BUSY_CYCLES'] / (wall * 1024)      # 256 CUs x 4 SIMDs
    w = row.get('SQ_WAVE_CYCLES')
    if w:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU',
                  'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM'):
            if c in row:
                d[c.lower() + '_pct_of_wave'] = 100.0 * row[c] / w
    if 'SQ_LDS_IDX_ACTIVE' in row and row['SQ_LDS_IDX_ACTIVE']:
        d['lds_conflict_pct'] = 100.0 * row.get('SQ_LDS_BANK_CONFLICT', 0) / row['SQ_LDS_IDX_ACTIVE']
    if row.get('SQ_INSTS_MFMA'):
        d['valu_insts_per_mfma'] = row.get('SQ_INSTS_VALU', 0.0) / row['SQ_INSTS_MFMA']
    if 'FETCH_SIZE' in row:
        d['read_bytes'] = row['FETCH_SIZE'] * 1024 * 2
    if 'WRITE_SIZE' in row:
        d['write_bytes'] = row['WRITE_SIZE'] * 1024
    if 'TCC_HIT_sum' in row:
        d['l2_hit_pct'] = 100.0 * row['TCC_HIT_sum'] / max(1.0, row['TCC_HIT_sum'] + row.get('TCC_MISS_sum', 0))
    return d


def main():
    root = sys.argv[1]
    rows = load(root)
    res = []
    for (name, grid), row in sorted(rows.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        d = derive(row)
        res.append(dict(kernel=name, grid=grid, counters=row, derived=d))
        print(f'\n{name}  grid={grid} wg={row["wg"]} vgpr={row["vgpr"]}+{row["agpr"]} lds={row["lds"]}')
        for k, v in sorted(row.items()):
            if k not in ('vgpr', 'agpr', 'lds', 'wg'):
                print(f'    {k:32s} {v:16.1f}')
        for k, v in d.items():
            print(f'  * {k:32s} {v:16.2f}')
    if '--json' in sys.argv:
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
