# dev helper: per-layer timing of one launch set of B n^3 tiles (usage: layer_times.py [arch] [n] [B]); times are per tile
import os, sys, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = random_state_dict(arch, seed=0)
dm = DenseModel(m, os.environ.get("SD_ACT", "bf16"), torch.device("cuda", 0))
x = torch.randint(0, 256, (B, n, n, n), dtype=torch.uint8, device='cuda')
out = torch.empty((B, dm.out_channels, n, n, n), dtype=torch.uint8, device='cuda')
for _ in range(3): dm.forward_batch(x, L.SD_OUT_PROBS_U8, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): dm.forward_batch(x, L.SD_OUT_PROBS_U8, out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 / B
print(f'{arch} {n}^3 x{B}: {t:.3f} ms/tile -> {n**3 / t / 1e3:.1f} Mvox/s')
dm.profile(5)
acc = np.zeros(dm.n_ops)
for _ in range(5):
    dm.forward_batch(x, L.SD_OUT_PROBS_U8, out)
for k in range(5):
    acc += dm.profile_read(k)
acc /= 5 * B
names = {1: 'conv', 2: 'pool', 3: 'upconv', 4: 'gn', 5: 'final'}
for i, (o, ms) in enumerate(zip(dm.ops, acc)):
    print(f'  op{i:2d} {names[o.kind]:6s} k{o.kz} cin {o.cin0}+{max(o.cin1,0)} -> {o.cout}: {ms*1e3:8.1f} us')
print('  sum', acc.sum())
