"""Rank-0 host work of the chunk pipeline (syconn_amd.parallel.predict_volume_distributed) measured alone: cutting chunk +
halo boxes out of a host volume into pinned staging (pack) and stitching uint8 results back (stitch), in GB/s, for the
config-4 volume (2048 x 2048 x 512) in the 128^3-tile chunk geometry and the reference's.  An 8-GPU strong-scaling run of
config 4 at >= 6x one GPU's ~900 Mvox/s needs the root to sustain ~5.4 Gvox/s = pack (x1.3 .. 3.1 halo) + stitch of 1 byte
per voxel: this prints what one host sustains."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd._lib import host_box_copy, host_zero                      # noqa: E402

vs = np.array((512, 2048, 2048))
vol = torch.empty(tuple(int(v) for v in vs), dtype=torch.uint8).random_(0, 255)
out = torch.empty_like(vol)
for name, cs, ol in (('tile128', (224, 192, 192), (8, 16, 16)), ('reference', (236, 481, 482), (20, 31, 30))):
    cs, ol = np.array(cs), np.array(ol)
    in_shape = cs + 2 * ol
    grid = [int(-(-vs[i] // cs[i])) for i in range(3)]
    ids = [(z, y, x) for z in range(grid[0]) for y in range(grid[1]) for x in range(grid[2])]
    pin_in = torch.empty(tuple(int(v) for v in in_shape), dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(tuple(int(v) for v in in_shape), dtype=torch.uint8)
    res = torch.empty(tuple(int(v) for v in cs), dtype=torch.uint8).random_(0, 255)
    for nthreads in (16, 64):
        t0 = time.perf_counter()
        nbytes = 0
        for cid in ids:
            lo = np.array(cid) * cs - ol
            hi = lo + in_shape
            a, b = np.maximum(lo, 0), np.minimum(hi, vs)
            if np.any(a > lo) or np.any(b < hi):
                host_zero(pin_in, nthreads)
            host_box_copy(pin_in[a[0] - lo[0]:b[0] - lo[0], a[1] - lo[1]:b[1] - lo[1], a[2] - lo[2]:b[2] - lo[2]],
                          vol[a[0]:b[0], a[1]:b[1], a[2]:b[2]], nthreads)
            nbytes += int(np.prod(in_shape))
        t_pack = time.perf_counter() - t0
        t0 = time.perf_counter()
        for cid in ids:
            lo = np.array(cid) * cs
            n = np.minimum(cs, vs - lo)
            host_box_copy(out[lo[0]:lo[0] + n[0], lo[1]:lo[1] + n[1], lo[2]:lo[2] + n[2]], res[:n[0], :n[1], :n[2]], nthreads)
        t_st = time.perf_counter() - t0
        nv = float(np.prod(vs))
        print(f'{name} geometry, {len(ids)} chunks, {nthreads} host threads: pack {nbytes / t_pack / 1e9:.1f} GB/s of payload '
              f'({t_pack:.2f} s), stitch {nv / t_st / 1e9:.1f} GB/s ({t_st:.2f} s) -> root host sustains '
              f'{nv / (t_pack + t_st) / 1e9:.2f} Gvox/s of volume (one thread doing both in turn)')
