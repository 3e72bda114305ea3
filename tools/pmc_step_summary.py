"""pmc_<name>_report.json (tools/pmc_report.py) -> the one-line-per-kernel table of profiles/r*_pmc_step/summary_per_kernel.txt:
wall cycles of one dispatch, matrix-pipe busy share, and where the waves' cycles went.

    python tools/pmc_step_summary.py gpurun_out/r6/<job>/pmc_step_report.json > profiles/r6_pmc_step/summary_per_kernel.txt"""
import json
import sys


def main():
    rows = json.load(open(sys.argv[1]))
    rows = [r for r in rows if r.get('derived', {}).get('wall_cycles')]
    rows.sort(key=lambda r: -r['derived']['wall_cycles'])
    print('rocprofv3 --pmc (three SQ counter groups, one per pass) over `python bench.py --steps 1 --warmup 1 --no-cpu-baseline` '
          '(ViT-H anchor, 8 tiles):\nper kernel and grid size the mean over its dispatches; wall = GRBM_GUI_ACTIVE / 8 in cycles of ONE '
          'dispatch; percentages of the waves\' cycles.\n')
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
        d = r['derived']
        print(f"{r['kernel'][:78]:78s} grid {r['grid']:9d} wall {d['wall_cycles'] / 1e6:7.3f} Mcycles  MFMA busy {d.get('mfma_busy_pct', 0):5.1f} %  "
              f"at waitcnt/barrier {d.get('sq_wait_any_pct_of_wave', 0):5.1f} %  waiting for issue {d.get('sq_wait_inst_any_pct_of_wave', 0):5.1f} %  "
              f"issuing {d.get('sq_active_inst_any_pct_of_wave', 0):5.1f} % (VALU {d.get('sq_active_inst_valu_pct_of_wave', 0):5.1f} %)  "
              f"vgpr {r['counters'].get('vgpr')}+{r['counters'].get('agpr')} lds {r['counters'].get('lds')}")


if __name__ == '__main__':
    main()
