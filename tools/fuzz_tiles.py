# dev fuzz: Predictor tiling with clipped windows + output boxes (clip_tiles=True) against whole windows (clip_tiles=False) for
# random tile / overlap / volume shapes and valid boxes: equal inside the box, zero outside.  usage: fuzz_tiles.py [seconds]
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd.handler.prediction import Predictor
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
dev = torch.device('cuda', 0)
sds = {a: random_state_dict(a, seed=5, final_scale=4.0) for a in ('myelin', 'semseg_axon', 'syntype')}
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    arch = str(rng.choice(list(sds)))
    act = str(rng.choice(['bf16', 'f16', 'f16x2']))
    tile = np.array([int(rng.integers(6, 24)), int(rng.integers(24, 72)), int(rng.integers(24, 80))])
    ol = np.array([int(rng.integers(0, 10)), int(rng.integers(0, 40)), int(rng.integers(0, 40))])
    nt = np.array([int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.integers(1, 3))])
    shape = tile * nt
    nc = sds[arch]['conv_final.bias'].numel()
    kw = dict(strict_shapes=True, tile_shape=tuple(tile), out_shape=(nc, *shape), overlap_shape=tuple(ol), apply_softmax=True, act_dtype=act)
    a, b = Predictor(sds[arch], device=dev, **kw), Predictor(sds[arch], device=dev, clip_tiles=False, **kw)
    vlo = np.array([int(rng.integers(0, s // 2 + 1)) for s in shape])
    vhi = np.array([int(rng.integers(l + 1, s + 1)) for l, s in zip(vlo, shape)])
    box = (tuple(int(v) for v in vlo), tuple(int(v) for v in vhi)) if rng.random() < 0.8 else None
    x = torch.randint(0, 256, tuple(int(s) for s in shape), dtype=torch.uint8, device=dev)
    pa, pb = a.predict_proba_u8_device(x, valid_box=box), b.predict_proba_u8_device(x, valid_box=box)
    sl = (slice(None),) + (tuple(slice(l, h) for l, h in zip(vlo, vhi)) if box is not None else (slice(None),) * 3)
    if not torch.equal(pa[sl], pb[sl]):
        print('MISMATCH', arch, act, tile, ol, nt, box); sys.exit(1)
    ids, thr = list(range(1, nc)), [float(rng.integers(60, 200))] * (nc - 1)
    la, lb = a.predict_labels_u8_device(x, ids, thr, valid_box=box), b.predict_labels_u8_device(x, ids, thr, valid_box=box)
    if not torch.equal(la[sl[1:]], lb[sl[1:]]):
        print('LABEL MISMATCH', arch, act, tile, ol, nt, box); sys.exit(1)
    n += 1
print(f'fuzz_tiles: {n} cases ok in {time.time() - t0:.0f} s')
