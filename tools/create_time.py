# dev probe: wall time of DenseModel creation (plan + BatchNorm fold + weight packing + upload) per storage type
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd.engine import DenseModel
dev = torch.device('cuda', 0)
for arch in ('myelin', 'mivcsj'):
    sd = random_state_dict(arch, seed=0)
    for act in ('bf16', 'f16x2', 'bf16', 'f16x2'):
        t = time.perf_counter(); dm = DenseModel(sd, act, dev); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f'{arch} {act}: {dt * 1e3:.1f} ms')
        del dm
