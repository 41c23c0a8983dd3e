# dev tool: fused level-0 decoder vs the separate layers on tile shapes other than the headline's (time of the last plan ops)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
sd = random_state_dict('myelin', seed=0, final_scale=8.0)
a = DenseModel(sd, 'bf16', torch.device('cuda', 0))
os.environ['SD_NO_DEC0'] = '1'
b = DenseModel(sd, 'bf16', torch.device('cuda', 0))
os.environ.pop('SD_NO_DEC0')
for nb, shape in ((1, (178, 243, 331)), (2, (178, 243, 331)), (4, (178, 243, 331)), (8, (128, 128, 128)), (1, (128, 128, 128)), (4, (64, 200, 200)), (2, (100, 130, 66))):
    x = torch.randint(0, 256, (nb, *shape), dtype=torch.uint8, device='cuda')
    res = []
    for m in (a, b):
        for _ in range(2): m.forward_batch(x, L.SD_OUT_PROBS_U8)
        m.profile(3)
        for _ in range(3): m.forward_batch(x, L.SD_OUT_PROBS_U8)
        acc = sum(m.profile_read(k) for k in range(3)) / 3
        m.profile(0)
        res.append((float(acc.sum()), float(acc[-4:].sum())))
    print(nb, shape, 'fused: total %.2f ms, decoder-0 ops %.3f ms | layers: total %.2f ms, decoder-0 ops %.3f ms' % (res[0] + res[1]))
