"""GPU box: per-op times (us per tile, 8 x 128^3 per launch set) under a list of environment settings, interleaved repetitions.
usage: layer_times_env.py <arch> <act> "NAME=VAL,NAME2=VAL2" "..." (an empty string = no switch)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict                            # noqa: E402
from syconn_amd.engine import DenseModel                                # noqa: E402

arch, act = sys.argv[1], sys.argv[2]
envs = sys.argv[3:] or ['']
dev = torch.device('cuda', 0)
dm = DenseModel(random_state_dict(arch, seed=0, final_scale=8.0), act, dev)
shape = tuple(int(v) for v in os.environ.get('LAYER_SHAPE', '8,128,128,128').split(','))      # tiles, z, y, x
x = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev)
ids, thr = list(range(1, dm.out_channels)), [127.5] * (dm.out_channels - 1)
res = {e: [] for e in envs}
for rep in range(3):
    for e in envs:
        kv = [p.split('=') for p in e.split(',') if p]
        for k, v in kv:
            os.environ[k] = v
        for _ in range(2):
            dm.forward_labels_batch(x, ids, thr)
        torch.cuda.synchronize()
        dm.profile(8)
        for _ in range(8):
            dm.forward_labels_batch(x, ids, thr)
        res[e].append(sum(dm.profile_read(k) for k in range(8)) / 8 / shape[0] * 1e3)
        dm.profile(0)
        for k, v in kv:
            os.environ.pop(k)
for e in envs:
    us = np.mean(res[e], axis=0)
    print(f'{e or "(default)":28s}', ' '.join(f'{i}:{u:.1f}' for i, u in enumerate(us) if u > 0), f'sum {us.sum():.1f}',
          'reps', ' '.join(f'{r.sum():.1f}' for r in res[e]))
