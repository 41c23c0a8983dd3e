# dev fuzz: random architectures / storage types / tile shapes / batch sizes / output boxes -- the values inside the box
# (sd_model_set_roi) and inside clipped windows must equal the whole-tile pass bit for bit.  usage: fuzz_roi.py [seconds]
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd import _lib as L
from syconn_amd.cnn import random_state_dict
from syconn_amd.engine import DenseModel
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
dev = torch.device('cuda', 0)
models = {}
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    arch = str(rng.choice(['myelin', 'semseg_spine', 'semseg_axon', 'er', 'syntype', 'syntype_enh']))
    act = str(rng.choice(['bf16', 'f16', 'f16x2']))
    if (arch, act) not in models:
        models[(arch, act)] = DenseModel(random_state_dict(arch, seed=3, final_scale=4.0), act, dev)
    dm = models[(arch, act)]
    shape = (int(rng.integers(4, 60)), int(rng.integers(9, 150)), int(rng.integers(9, 170)))
    N = int(rng.choice([1, 1, 2, 3]))
    x = torch.randint(0, 256, (N, *shape), dtype=torch.uint8, device=dev)
    lo = [int(rng.integers(0, s)) for s in shape]
    hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, shape)]
    kind = int(rng.choice([L.SD_OUT_PROBS_U8, L.SD_OUT_LOGITS_F32]))
    full = dm.forward_batch(x, kind)
    got = dm.forward_batch(x, kind, roi=(tuple(lo), tuple(hi)))
    sl = (slice(None), slice(None)) + tuple(slice(a, b) for a, b in zip(lo, hi))
    if not torch.equal(full[sl], got[sl]):
        d = (full[sl].float() - got[sl].float()).abs()
        print('MISMATCH', arch, act, shape, N, lo, hi, kind, float(d.max()), int((d > 0).sum()))
        sys.exit(1)
    if dm.overflowed():
        print('overflow flag', arch, act, shape); sys.exit(1)
    n += 1
print(f'fuzz_roi: {n} cases ok in {time.time() - t0:.0f} s')
