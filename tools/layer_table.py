"""Per-layer table of one launch set (GPU box): time per tile, algorithmic GFLOP and MB per tile, achieved PFLOP/s and TB/s --
the table DESIGN.md section 5 quotes for the three judged models.  usage: layer_table.py <arch> <act> [tiles=8] [edge=128 | DxHxW]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd import _lib as L                                        # noqa: E402
from syconn_amd.cnn import random_state_dict                            # noqa: E402
from syconn_amd.engine import DenseModel                                # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
act = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
shape = tuple(int(v) for v in sys.argv[4].split('x')) if len(sys.argv) > 4 and 'x' in sys.argv[4] else (int(sys.argv[4]) if len(sys.argv) > 4 else 128,) * 3
dm = DenseModel(random_state_dict(arch, seed=0, final_scale=8.0), act, torch.device('cuda', 0))
x = torch.randint(0, 256, (B, *shape), dtype=torch.uint8, device='cuda')
ids, thr = list(range(1, dm.out_channels)), [127.5] * (dm.out_channels - 1)
# LAYER_ROI="z0,y0,x0:z1,y1,x1": output box of interest (sd_model_set_roi), e.g. the core a tile keeps
ROI = tuple(tuple(int(v) for v in part.split(',')) for part in os.environ['LAYER_ROI'].split(':')) if os.environ.get('LAYER_ROI') else None
for _ in range(3):
    dm.forward_labels_batch(x, ids, thr, roi=ROI)
torch.cuda.synchronize()
dm.profile(5)
for _ in range(5):
    dm.forward_labels_batch(x, ids, thr, roi=ROI)
us = sum(dm.profile_read(k) for k in range(5)) / 5 / B * 1e3
# shapes per buffer
dims = {0: shape}
shape_s = 'x'.join(map(str, shape))
chans = {0: 1}
rows = []
names = {1: 'conv', 2: 'pool', 3: 'upconv', 4: 'groupnorm', 5: 'final'}
w = 2
for i, o in enumerate(dm.ops):
    k = int(o.kind)
    a = dims.get(o.src0)
    if k == 1:
        out = dims[o.src1] if o.src1 >= 0 else a
        vox = np.prod(out)
        cin = o.cin0 + max(o.cin1, 0)
        fl = 2.0 * o.kz * 9 * cin * o.cout * vox
        by = (1 if o.src0 == 0 else w * o.cin0) * vox + (w * o.cin1 * vox if o.src1 >= 0 else 0) + w * o.cout * vox
        dims[o.dst], chans[o.dst] = out, o.cout
        what = f'conv {o.kz}x3x3 {cin} -> {o.cout} @ {out[0]}x{out[1]}x{out[2]}'
    elif k == 2:
        out = ((a[0] + 1) // 2 if o.kz == 2 else a[0], (a[1] + 1) // 2, (a[2] + 1) // 2)
        dims[o.dst], chans[o.dst] = out, chans[o.src0]
        fl, by = 0.0, w * chans[o.src0] * (np.prod(a) + np.prod(out))
        what = f'pool @ {a[0]}x{a[1]}x{a[2]}'
    elif k == 3:
        out = (a[0] * o.kz, a[1] * 2, a[2] * 2)
        dims[o.dst], chans[o.dst] = out, o.cout
        fl = 2.0 * o.cin0 * o.cout * o.kz * 4 * np.prod(a)
        by = w * (o.cin0 * np.prod(a) + o.cout * np.prod(out))
        what = f'up-conv {o.cin0} -> {o.cout} to {out[0]}x{out[1]}x{out[2]}'
    elif k == 4:
        fl, by = 0.0, 2 * w * chans[o.src0] * np.prod(a)
        what = f'GroupNorm {chans[o.src0]} ch @ {a[0]}x{a[1]}x{a[2]}'
    else:
        fl = 2.0 * o.cin0 * o.cout * np.prod(a)
        by = w * o.cin0 * np.prod(a) + np.prod(a)
        what = f'final {o.cin0} -> {o.cout} + softmax + labels'
    rows.append((i, what, us[i], fl / 1e9, by / 1e6))
print(f'| op | layer ({arch}, {act}, {B} x {shape_s} per launch set) | us / tile | GFLOP | PFLOP/s | MB (algorithmic) | TB/s |')
print('|---|---|---|---|---|---|---|')
for i, what, t, gf, mb in rows:
    if t < 1.0:
        print(f'| {i} | {what} | fused | {gf:.1f} | | {mb:.0f} | |')
    else:
        print(f'| {i} | {what} | {t:.1f} | {gf:.1f} | {gf / t:.2f} | {mb:.0f} | {mb / t:.2f} |')
tt = float(us.sum())
print(f'| | **sum** | **{tt:.0f}** | {sum(r[3] for r in rows):.0f} | {sum(r[3] for r in rows) / tt:.2f} | | |')
