#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc SQ pass (csv): mean counter values per launch of each (kernel, grid), plus the
derived ratios used in DESIGN.md (units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES count cycles):
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)
usage: tools/pmc_sq_summary.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        k = (r['Kernel_Name'][:88], r['Grid_Size'], r['LDS_Block_Size'], r['VGPR_Count'])
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for k in sorted(agg, key=lambda k: -sum(dur[k])):
        if flt not in k[0]:
            continue
        d = {c: sum(v) / len(v) for c, v in agg[k].items()}
        n = len(next(iter(agg[k].values())))
        line = f'{k[0]:88s} grid={k[1]:>8s} lds={k[2]:>6s} vgpr={k[3]:>3s} n={n:3d} dur_us={sum(dur[k]) / len(dur[k]):8.1f}'
        wc = d.get('SQ_WAVE_CYCLES')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and d.get('SQ_BUSY_CU_CYCLES'):
            line += f" mfma_busy_frac={d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * d['SQ_BUSY_CU_CYCLES']):.3f}"
        for c in sorted(d):
            if c in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CU_CYCLES'):
                continue
            if wc and (c.startswith('SQ_WAIT') or c.startswith('SQ_ACTIVE_INST')):
                line += f' {c[3:].lower()}/wave_cycles={d[c] / wc:.3f}'
            else:
                line += f' {c[3:].lower()}={d[c]:.0f}'
        print(line)


if __name__ == '__main__':
    main()
