#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of bench.py.
usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tiles_per_launch> <out.json> [traffic.json [arch act]]
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes
for 16 B/lane streams on gfx950; WRITE_SIZE needs no correction (the first conv writes exactly its algorithmic bytes)."""
import collections, csv, json, re, sys


def bench_name(sym: str):
    """kernel symbol as rocprofv3 prints it -> the name the library reports (sd_debug_op_kernel), which bench.py keys its roofline on"""
    m = re.search(r'k_conv_mfmaIDF16([b_])Li(\d)ELi(\d)ELi(\d+)ELi(n?\d)ELi(\d)ELi(\d)E', sym)
    if m:
        t, kz, nt, waves, nslot, mt, mode = m.groups()
        return 'k_conv_mfma<%s,%sx3x3,NT=%s,WAVES=%s,NSLOT=%s,MT=%s,MODE=%s>' % ('bf16' if t == 'b' else 'f16', kz, nt, waves,
                                                                                 nslot.replace('n', '-'), mt, mode)
    # rocprofv3 half-demangles the planar forms of the 16-bit types ("k_conv_mfma<bool _Accum, int, E, NT, WAVES, NSLOT, MT, MODE>"): the
    # storage type and KZ = 1 are swallowed by the bogus "bool _Accum, int, E" (3x3x3 symbols stay mangled and are matched above)
    m = re.search(r'k_conv_mfma<bool _Accum, int, E, (\d), (\d+), (-?\d), (\d), (\d)>', sym)
    if m:
        return 'k_conv_mfma<*,1x3x3,NT=%s,WAVES=%s,NSLOT=%s,MT=%s,MODE=%s>' % m.groups()
    if 'k_dec0' in sym:
        return 'k_dec0'
    if 'k_conv_first' in sym:
        return 'k_conv_first'
    if 'k_upconv_rows' in sym:
        return 'k_upconv_rows'
    if 'k_upconv_mfma' in sym:
        return 'k_upconv_mfma'
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
    if len(sys.argv) > 5:
        # profiles/traffic.json: mean HBM bytes per launch by library kernel name (all launches of a name pooled, weighted by count)
        acc = collections.defaultdict(lambda: [0.0, 0])
        for k in fetch:
            name = bench_name(k[0])
            if name:
                acc[name][0] += ((2 * fetch[k] + write.get(k, 0.0)) * 1024) * nf[k]
                acc[name][1] += nf[k]
        arch, act = (sys.argv[6:8] + ['semseg_spine', 'bf16'])[:2]
        json.dump({'arch': arch, 'tile': 128, 'act': act, 'tiles_per_launch': tiles,
                   'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of bench.py; '
                             'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16 B/lane '
                             'streams on gfx950, WRITE_SIZE needs no correction; produced by tools/pmc_traffic.py',
                   'hbm_bytes_per_launch': {n: v[0] / v[1] for n, v in acc.items()},
                   'note': '(2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes, mean over the launches of that kernel '
                           'symbol; each launch processes %d tiles (sd_forward_batch)' % tiles},
                  open(sys.argv[5], 'w'), indent=1)


if __name__ == '__main__':
    main()
