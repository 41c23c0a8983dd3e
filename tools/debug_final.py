import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.unet_ref import build_unet
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
os.environ['SD_KEEP_ALL'] = '1'
model = build_unet('myelin', seed=13, n_blocks=2, start_filts=32)
g = torch.Generator().manual_seed(17)
raw = torch.randint(0, 256, (40, 200, 240), generator=g, dtype=torch.uint8)
dm = DenseModel(model, 'bf16', torch.device('cuda', 0))
out = dm.forward(raw.cuda(), L.SD_OUT_LOGITS_F32).cpu()
x = dm.read_buffer(8).cpu()                       # (32, D, H, W) activations as stored by the same kernel
w = model.conv_final.weight.detach().reshape(2, 32)
b = model.conv_final.bias.detach()
exp = torch.einsum('oc,czyx->ozyx', w, x) + b[:, None, None, None]
d = out - exp
bad = (d.abs() > 1e-3).nonzero()
print('bad entries', len(bad), 'bias', b.tolist())
# candidate missing terms per channel quad-group: lane half h owns channels 4h + 8q + e
parts = {}
for h in (0, 1):
    idx = [4 * h + 8 * q + e for q in range(4) for e in range(4)]
    parts[h] = torch.einsum('oc,czyx->ozyx', w[:, idx], x[idx])
for k in range(min(12, len(bad))):
    co, z, y, xx = bad[k].tolist()
    print((co, z, y, xx), 'diff %.5f' % float(d[co, z, y, xx]), 'bias %.5f' % float(b[co]),
          'part_lower %.5f part_upper %.5f' % (float(parts[0][co, z, y, xx]), float(parts[1][co, z, y, xx])),
          'got %.5f exp %.5f' % (float(out[co, z, y, xx]), float(exp[co, z, y, xx])),
          'tile0-voxel (y-2) parts: %.5f %.5f' % (float(parts[0][co, z, y - 2, xx]), float(parts[1][co, z, y - 2, xx])))
import collections
print('bad per class', collections.Counter(bad[:, 0].tolist()))
