"""GPU box: kernel time per tile (us, 8 x 128^3 per launch set) of a model PLANNED under a list of environment settings (switches that
sd_model_create reads: SD_NO_GN_DEFER, SD_NO_GN_FUSE, SD_NO_DEC0, ...), interleaved repetitions; per-op times of every plan.
usage: plan_times_env.py <arch> <act> "NAME=VAL,NAME2=VAL2" "..." (an empty string = no switch)"""
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
x = torch.randint(0, 256, (8, 128, 128, 128), dtype=torch.uint8, device=dev)
sd = random_state_dict(arch, seed=0, final_scale=8.0)
models = {}
for e in envs:
    kv = [p.split('=') for p in e.split(',') if p]
    for k, v in kv:
        os.environ[k] = v
    models[e] = DenseModel(sd, act, dev)
    for k, v in kv:
        os.environ.pop(k)
ids = list(range(1, models[envs[0]].out_channels))
thr = [127.5] * len(ids)
res = {e: [] for e in envs}
ref = None
for rep in range(3):
    for e in envs:
        dm = models[e]
        kv = [p.split('=') for p in e.split(',') if p]
        for k, v in kv:                      # (launch-time switches are read per launch)
            os.environ[k] = v
        for _ in range(2):
            lab = dm.forward_labels_batch(x, ids, thr)
        torch.cuda.synchronize()
        if rep == 0:
            if ref is None:
                ref = lab.clone()
            else:
                print(f'{e or "(default)":28s} labels differ from the first plan on {(lab != ref).float().mean().item():.2e} of the voxels')
        dm.profile(8)
        for _ in range(8):
            dm.forward_labels_batch(x, ids, thr)
        res[e].append(sum(dm.profile_read(k) for k in range(8)) / 8 / 8 * 1e3)
        dm.profile(0)
        for k, v in kv:
            os.environ.pop(k)
for e in envs:
    us = np.mean(res[e], axis=0)
    print(f'{e or "(default)":28s}', ' '.join(f'{i}:{u:.1f}' for i, u in enumerate(us) if u > 0), f'sum {us.sum():.1f}',
          'reps', ' '.join(f'{r.sum():.1f}' for r in res[e]))
