# dev helper: per-layer timing of sd_forward_labels_batch (the bench's launch set) vs sd_forward_batch(PROBS_U8)
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
B = 8
dm = DenseModel(random_state_dict(arch, seed=0, final_scale=8.0), os.environ.get('SD_ACT', 'bf16'), torch.device('cuda', 0))
x = torch.randint(0, 256, (B, 128, 128, 128), dtype=torch.uint8, device='cuda')
ids = list(range(1, dm.out_channels)); thr = [127.5] * len(ids)
for mode in ('labels', 'probs_u8'):
    run = (lambda: dm.forward_labels_batch(x, ids, thr)) if mode == 'labels' else (lambda: dm.forward_batch(x, L.SD_OUT_PROBS_U8))
    for _ in range(3): run()
    dm.profile(5)
    for _ in range(5): run()
    acc = sum(dm.profile_read(k) for k in range(5)) / 5 / B
    dm.profile(0)
    print(mode, 'last conv op us/tile:', [round(v * 1e3, 1) for v in acc[-3:]], 'sum ms', round(float(acc.sum()), 4))
