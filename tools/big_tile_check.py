import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet   # seeded random weights (no trained models exist)
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
dm = DenseModel(build_unet('myelin', seed=1, final_scale=6.0), 'bf16', torch.device('cuda', 0))
x = torch.randint(0, 256, (256, 640, 640), dtype=torch.uint8, device='cuda')
print('workspace GiB', dm.workspace_bytes(x.shape) / 2**30)
a = dm.forward(x, L.SD_OUT_PROBS_U8); torch.cuda.synchronize()
t = time.perf_counter(); b = dm.forward(x, L.SD_OUT_PROBS_U8).clone(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print('105 Mvox tile:', dt * 1e3, 'ms', x.numel() / dt / 1e6, 'Mvox/s', torch.equal(a, b))
s = a.to(torch.int32).sum(0); print(int(s.min()), int(s.max()))
# interior block equals the same region predicted as its own tile (RF 44 in y/x, 20 in z for this net -> margin 48)
sub = x[64:192, 200:456, 200:456].contiguous()
c = dm.forward(sub, L.SD_OUT_PROBS_U8)
print('interior equal:', torch.equal(c[:, 48:80, 48:208, 48:208], a[:, 112:144, 248:408, 248:408]))
