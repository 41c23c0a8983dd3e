# dev helper (needs a -DSD_TIMING build): per-wave cycle stamps of one conv op, second block of each workgroup
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.unet_ref import build_unet
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
op = int(sys.argv[1]); os.environ['SD_TIMING_OP'] = str(op)
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nwg = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dm = DenseModel(build_unet('semseg_spine', seed=0), 'bf16', torch.device('cuda', 0))
x = torch.randint(0, 256, (128, 128, 128), dtype=torch.uint8, device='cuda')
out = torch.empty((5, 128, 128, 128), dtype=torch.uint8, device='cuda')
for _ in range(3): dm.forward(x, L.SD_OUT_PROBS_U8, out)
torch.cuda.synchronize()
raw = dm._ws[65536:65536 + nwg * waves * 64].view(torch.int64).cpu().numpy().reshape(nwg, waves, 8)
d = np.diff(raw[:, :, :7], axis=2).astype(np.float64)
names = ['dma issue', 'stage-0 MFMA loop', 'stage-0 wait+barrier', 'remaining stages', 'main store', 'pool+final']
ok = (d >= 0).all(axis=2) & (d < 1e7).all(axis=2)
print('valid waves', ok.sum(), 'of', ok.size)
for i, n in enumerate(names):
    v = d[:, :, i][ok]
    print(f'{n:24s} median {np.median(v):9.0f}  mean {v.mean():9.0f}  p90 {np.percentile(v, 90):9.0f} cycles')
tot = (raw[:, :, 6] - raw[:, :, 0])[ok]
print('block total median', np.median(tot), 'cycles  (s_memtime counts at 100 MHz? check scale)')
