#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of bench.py.
usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tiles_per_launch> <out.json> [traffic.json]
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes
for 16 B/lane streams on gfx950; WRITE_SIZE needs no correction (the first conv writes exactly its algorithmic bytes)."""
import collections, csv, json, re, sys


def bench_name(sym: str):
    """kernel symbol -> the name bench.py's layer_accounting uses"""
    m = re.search(r'k_conv_mfmaIDF16[b_]Li(\d)ELi(\d)ELi(\d+)ELi(\d)E', sym)
    if m:
        return 'k_conv_mfma<%sx3x3,NT=%s,%s waves,NSLOT=%s>' % m.groups()
    m = re.search(r'k_conv_mfma<.*?(\d), (\d+), (\d), (\d), (true|false)>', sym)   # partially demangled: <T, KZ?..>
    if m:
        return None          # KZ is lost in this demangling; resolved by the caller from the launch grid
    if 'k_conv_first' in sym:
        return 'k_conv_first'
    if 'k_upconv' in sym:
        return 'upconv'
    return None


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[(r['Kernel_Name'], r['Grid_Size'], r['LDS_Block_Size'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, nf = load(sys.argv[1], 'FETCH_SIZE')
    write, _ = load(sys.argv[2], 'WRITE_SIZE')
    tiles = int(sys.argv[3])
    out = {}
    for k in sorted(fetch):
        f, w = fetch[k], write.get(k, 0.0)
        out['%s grid=%s lds=%s' % k] = {'launches': nf[k], 'read_MB_x2': round(2 * f * 1024 / 1e6, 1),
                                          'write_MB': round(w * 1024 / 1e6, 1),
                                          'hbm_bytes': (2 * f + w) * 1024}
    json.dump({'tiles_per_launch': tiles, 'kernels': out}, open(sys.argv[4], 'w'), indent=1)
    for k, v in out.items():
        print(f"{k[:110]:110s} n={v['launches']:3d} read {v['read_MB_x2']:9.1f} MB  write {v['write_MB']:9.1f} MB")


if __name__ == '__main__':
    main()
