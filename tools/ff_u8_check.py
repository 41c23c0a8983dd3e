"""GPU box: the uint8 first convolution on the bf16 pipe (k_conv_mfma MODE 5 / k_conv_first) against the exact-f32 chain
(SD_NO_FIRST_U8=1) and against its own unfused form; per-op times of both.  usage: ff_u8_check.py [arch] [act]"""
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
dev = torch.device('cuda', 0)
sd = random_state_dict(arch, seed=0, final_scale=8.0)
dm = DenseModel(sd, act, dev)
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 256, (8, 128, 128, 128), dtype=torch.uint8, generator=g).to(dev)
xs = x[:2, :24, :100, :90].contiguous()
ids, thr = list(range(1, dm.out_channels)), [127.5] * (dm.out_channels - 1)


def run(inp, kind):
    return dm.forward_batch(inp, kind).cpu()


os.environ.pop('SD_NO_FIRST_U8', None)
new = run(xs, L.SD_OUT_LOGITS_F32)
os.environ['SD_NO_FIRST_U8'] = '1'
old = run(xs, L.SD_OUT_LOGITS_F32)
flt = run(xs.float() / 255., L.SD_OUT_LOGITS_F32)
os.environ.pop('SD_NO_FIRST_U8')
print('old uint8 == float32 input:', torch.equal(old, flt))
sc = float(old.abs().max())
print(f'new vs old: max |diff| / max|logit| = {float((new - old).abs().max()) / sc:.3e}, differing values {float((new != old).float().mean()):.3e}')
os.environ['SD_NO_FIRST_FUSE'] = '1'
plain = DenseModel(sd, act, dev)
os.environ.pop('SD_NO_FIRST_FUSE')
pn = plain.forward_batch(xs, L.SD_OUT_LOGITS_F32).cpu()
print('fused == unfused (new arithmetic):', torch.equal(pn, new))
# first layer output itself against fp64 arithmetic
os.environ['SD_KEEP_ALL'] = '1'
dk = DenseModel(sd, act, dev)
os.environ.pop('SD_KEEP_ALL')
dk.forward_batch(xs[:1], L.SD_OUT_LOGITS_F32)
b_new = dk.read_buffer(1).cpu().double()
os.environ['SD_NO_FIRST_U8'] = '1'
dk.forward_batch(xs[:1], L.SD_OUT_LOGITS_F32)
b_old = dk.read_buffer(1).cpu().double()
os.environ.pop('SD_NO_FIRST_U8')
print(f'first-layer buffer: new vs old max diff {float((b_new - b_old).abs().max()):.3e} on max {float(b_old.abs().max()):.3e}, '
      f'differing {float((b_new != b_old).double().mean()):.3e}')

for tag, env in (('new', None), ('old', '1')):
    if env:
        os.environ['SD_NO_FIRST_U8'] = env
    for _ in range(3):
        dm.forward_labels_batch(x, ids, thr)
    torch.cuda.synchronize()
    dm.profile(10)
    for _ in range(10):
        dm.forward_labels_batch(x, ids, thr)
    us = sum(dm.profile_read(k) for k in range(10)) / 10 / 8 * 1e3
    dm.profile(0)
    print(tag, 'us per tile:', ' '.join(f'{i}:{u:.1f}' for i, u in enumerate(us) if u > 0), 'sum', f'{us.sum():.1f}')
    os.environ.pop('SD_NO_FIRST_U8', None)
