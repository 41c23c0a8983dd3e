# dev helper: one chunk in the reference's hard-coded geometry (prediction.py:672-677): chunk 482x481x236 + halo
# (30,31,20) -> (276,543,542) zyx, 12 tiles of 138x181x271 + halo = model input 178x243x331
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet   # seeded random weights (no trained models exist)
from syconn_amd.handler.prediction import Predictor
model = build_unet('myelin', seed=0, final_scale=8.0)
p = Predictor(model, strict_shapes=True, tile_shape=(138, 181, 271), out_shape=(2, 276, 543, 542),
              overlap_shape=(20, 31, 30), apply_softmax=True)
raw = torch.randint(0, 256, (276, 543, 542), dtype=torch.uint8, device='cuda')
out = p.predict_proba_u8_device(raw); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): out = p.predict_proba_u8_device(raw)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
conv = 12 * 178 * 243 * 331
print(f'chunk {tuple(raw.shape)}: {dt*1e3:.1f} ms -> {raw.numel()/dt/1e6:.0f} Mvox/s of chunk voxels, {conv/dt/1e6:.0f} Mvox/s convolved, '
      f'useful {482*481*236/dt/1e6:.0f} Mvox/s; workspace {p._dm._ws.numel()/2**30:.2f} GiB; finite {bool(out.float().mean() > 0)}')
