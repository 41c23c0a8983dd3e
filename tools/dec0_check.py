# dev tool: fused level-0 decoder (sd_dec0.hip) against the layer-by-layer plan (SD_NO_DEC0=1): exact comparison of
# logits / uint8 probabilities / labels on ragged shapes, then timing of the headline launch set.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel

def models(arch, act, seed=0):
    sd = random_state_dict(arch, seed=seed, final_scale=8.0)
    os.environ.pop('SD_NO_DEC0', None)
    a = DenseModel(sd, act, torch.device('cuda', 0))
    os.environ['SD_NO_DEC0'] = '1'
    b = DenseModel(sd, act, torch.device('cuda', 0))
    os.environ.pop('SD_NO_DEC0', None)
    return a, b

bad = 0
shapes = [(1, (4, 32, 32)), (2, (3, 50, 70)), (1, (5, 33, 129)), (1, (2, 131, 64)), (3, (8, 64, 64)), (1, (1, 7, 5)), (1, (6, 96, 200))]
if len(sys.argv) > 1 and sys.argv[1] == 'quick': shapes = shapes[:2]
if len(sys.argv) > 1 and sys.argv[1] == 'time': shapes = []
for arch in ('semseg_spine', 'myelin', 'syntype'):
    for act in ('bf16', 'f16'):
        a, b = models(arch, act)
        for nb, shape in shapes:
            x = torch.randint(0, 256, (nb, *shape), dtype=torch.uint8, device='cuda')
            for kind in (L.SD_OUT_LOGITS_F32, L.SD_OUT_PROBS_U8):
                ya, yb = a.forward_batch(x, kind), b.forward_batch(x, kind)
                torch.cuda.synchronize()
                same = bool((ya == yb).all())
                d = float((ya.float() - yb.float()).abs().max())
                if not same: bad += 1
                print(arch, act, nb, shape, 'logits' if kind == L.SD_OUT_LOGITS_F32 else 'probs_u8', 'identical' if same else f'DIFF max {d:.3e} n={(ya != yb).sum().item()}')
            ids = list(range(1, a.out_channels)); thr = [100.0] * len(ids)
            la, lb = a.forward_labels_batch(x, ids, thr), b.forward_labels_batch(x, ids, thr)
            if not bool((la == lb).all()): bad += 1; print('  labels DIFF', (la != lb).sum().item())
print('mismatching cases:', bad)
a, b = models('semseg_spine', 'bf16')
x = torch.randint(0, 256, (8, 128, 128, 128), dtype=torch.uint8, device='cuda')
ids = list(range(1, a.out_channels)); thr = [127.5] * len(ids)
for name, m in (('fused', a), ('layers', b)):
    for _ in range(3): m.forward_labels_batch(x, ids, thr)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): m.forward_labels_batch(x, ids, thr)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    m.profile(5)
    for _ in range(5): m.forward_labels_batch(x, ids, thr)
    acc = sum(m.profile_read(k) for k in range(5)) / 5 / 8
    m.profile(0)
    print(name, f'{dt * 1e3 / 8:.4f} ms/tile', 'last ops us/tile:', [round(v * 1e3, 1) for v in acc[-5:]])
