"""Throughput of the device-side chunk driver of map_subcell_extract_props (syconn_amd/proc/sd_proc.py) on a
synthetic 2048 x 2048 x 512 dataset with a cell segmentation and three organelle segmentations (uint64 label volumes, 512^3 chunks,
all resident in HBM: 69 GB), checked against one pass over a whole sub-volume.

    python tools/propmerge_bench.py [--extent 2048 2048 512] [--chunk 512 512 512] [--reps 3]      -> one JSON line
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth(name, off, size, dev):
    """(x,y,z) int64 labels of the box at `off`: 'sv' = blocks of 96 x 80 x 64 voxels with 3-voxel gaps (cells cross chunk faces),
    organelles = balls on lattices of different pitch (some cut by chunk faces, most purely inside)."""
    ax = [torch.arange(int(off[i]), int(off[i] + size[i]), device=dev, dtype=torch.int64) for i in range(3)]
    X, Y, Z = ax[0][:, None, None], ax[1][None, :, None], ax[2][None, None, :]
    if name == 'sv':
        bx, by, bz = X // 96, Y // 80, Z // 64
        gap = ((X % 96) < 3) | ((Y % 80) < 3) | ((Z % 64) < 3)
        ids = 1 + bx + 64 * by + 4096 * bz
        return torch.where(gap, torch.zeros_like(ids), ids.expand(gap.shape)).contiguous()
    pitch, rad, salt = {'mi': (40, 12, 1), 'vc': (24, 6, 2), 'sj': (32, 5, 3)}[name]
    cx, cy, cz = X // pitch, Y // pitch, Z // pitch
    dx, dy, dz = X % pitch - pitch // 2, Y % pitch - pitch // 2, Z % pitch - pitch // 2
    inside = (dx * dx + dy * dy + dz * dz) < rad * rad
    ids = 1 + cx + 256 * cy + 65536 * cz + (salt << 40)
    return torch.where(inside, ids.expand(inside.shape), torch.zeros_like(inside, dtype=torch.int64)).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--extent', type=int, nargs=3, default=[2048, 2048, 512])
    ap.add_argument('--chunk', type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--min-vx', type=int, default=200)
    args = ap.parse_args()
    import syconn_amd.proc.sd_proc as sp
    from syconn_amd.extraction.find_object_properties import segstats
    dev = torch.device('cuda', 0)
    ext, cs = np.asarray(args.extent), np.asarray(args.chunk)
    names = ['mi', 'vc', 'sj']

    class KD:
        boundary = ext
    sp.kd_factory = lambda p: KD()
    store = {}

    def loader(name, off, size):
        key = (name, tuple(int(v) for v in off))
        if key not in store:
            store[key] = synth(name, off, size, dev)
        return store[key]
    mov = {'sv': 1, 'mi': args.min_vx, 'vc': args.min_vx, 'sj': args.min_vx}
    kw = dict(chunk_size=cs, min_obj_vx=mov, device=dev, as_tables=True, chunk_loader=loader)
    t0 = time.perf_counter()
    res = sp.map_subcell_extract_props('', {n: '' for n in names}, **kw)      # generates the volumes, warms every kernel
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    times = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sp.map_subcell_extract_props('', {n: '' for n in names}, **kw)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t_dicts0 = time.perf_counter()
    cell_d = res[0].as_dicts()
    sub_d = {n: res[1][n].as_dicts() for n in names}
    map_d = {n: res[2][n].as_dict() for n in names}
    t_dicts = time.perf_counter() - t_dicts0
    # check: unfiltered sizes of one organelle over a 2 x 2 x 1-chunk corner == one pass over that whole box
    sub_ext = np.minimum(ext, cs * np.array([2, 2, 1]))
    KD.boundary = sub_ext
    r2 = sp.map_subcell_extract_props('', {'mi': ''}, chunk_size=cs, min_obj_vx={'sv': 1, 'mi': 1}, device=dev, as_tables=True,
                                      chunk_loader=loader)
    whole = segstats(synth('sv', (0, 0, 0), sub_ext, dev), [synth('mi', (0, 0, 0), sub_ext, dev)], device=dev)
    ok = bool(np.array_equal(r2[1]['mi'].ids, whole.sub[0][0]) and np.array_equal(r2[1]['mi'].sizes, whole.sub[0][2]) and
              np.array_equal(r2[0].sizes, whole.cell[2]) and np.array_equal(r2[2]['mi'].counts, whole.pairs[0][2]))
    vox = float(np.prod(ext))
    best = min(times)
    print(json.dumps({
        'what': 'map_subcell_extract_props step 1 (device chunk driver): cell segmentation + 3 organelle segmentations, uint64, resident in HBM',
        'extent_xyz': [int(v) for v in ext], 'chunk': [int(v) for v in cs], 'n_chunks': int(np.prod(-(-ext // cs))),
        'min_obj_vx': mov, 'seconds': times, 'seconds_median': float(np.median(times)), 'first_call_seconds_incl_synthesis': t_first,
        'mvox_per_s': vox / best / 1e6, 'label_bytes_read_GBps': vox * 8 * 4 / best / 1e9,
        'objects': {'sv': len(res[0]), **{n: len(res[1][n]) for n in names}}, 'box_records': {'sv': int(len(res[0].boxes)), **{n: int(len(res[1][n].boxes)) for n in names}},
        'overlap_pairs': {n: len(res[2][n]) for n in names},
        'dict_building_seconds_at_the_api_edge': t_dicts, 'dict_entries': {'sv': len(cell_d[2]), **{n: len(sub_d[n][2]) for n in names}, 'maps': {n: len(map_d[n]) for n in names}},
        'chunked_equals_whole_volume_pass': ok}))
    assert ok


if __name__ == '__main__':
    main()
