# dev helper (needs -DSD_TIMING -DSD_STAGES): duration of every stage of the probed block of one conv op
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet   # seeded random weights (no trained models exist)
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
op = int(sys.argv[1]); os.environ['SD_TIMING_OP'] = str(op)
waves = int(sys.argv[2]); nwg = int(sys.argv[3]); nst = int(sys.argv[4])
dm = DenseModel(build_unet('semseg_spine', seed=0), 'bf16', torch.device('cuda', 0))
x = torch.randint(0, 256, (128, 128, 128), dtype=torch.uint8, device='cuda')
out = torch.empty((5, 128, 128, 128), dtype=torch.uint8, device='cuda')
for _ in range(200): dm.forward(x, L.SD_OUT_PROBS_U8, out)
torch.cuda.synchronize()
base = 65536 + (1 << 20) * 8
raw = dm._ws[base:base + nwg * waves * 128].view(torch.int64).cpu().numpy().reshape(nwg, waves, 16)
start = raw[:, :, 14]
ends = raw[:, :, :nst]
prev = start
for s in range(nst):
    d = (ends[:, :, s] - prev).astype(np.float64)
    print(f'stage {s:2d}: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}')
    prev = ends[:, :, s]
d = (raw[:, :, 15] - prev).astype(np.float64)
print(f'epilogue: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}')
print('block total median', np.median(raw[:, :, 15] - start))
