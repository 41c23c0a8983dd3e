# dev helper (needs a -DSD_TIMING build): per-wave cycle stamps of one conv op, second block of each workgroup
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet   # seeded random weights (no trained models exist)
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
op = int(sys.argv[1]); os.environ['SD_TIMING_OP'] = str(op)
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nwg = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dm = DenseModel(build_unet('semseg_spine', seed=0), 'bf16', torch.device('cuda', 0))
x = torch.randint(0, 256, (128, 128, 128), dtype=torch.uint8, device='cuda')
out = torch.empty((5, 128, 128, 128), dtype=torch.uint8, device='cuda')
for _ in range(int(os.environ.get("SD_PROBE_ITERS", "3"))): dm.forward(x, L.SD_OUT_PROBS_U8, out)
torch.cuda.synchronize()
raw = dm._ws[65536:65536 + nwg * waves * 64].view(torch.int64).cpu().numpy().reshape(nwg, waves, 8)
ok = np.ones(raw.shape[:2], bool)
def seg(a, b, name):
    v = (raw[:, :, b] - raw[:, :, a]).astype(np.float64)
    print(f'{name:34s} median {np.median(v):9.0f}  mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}')
seg(0, 7, 'block start -> probed stage start')
seg(7, 1, 'probed stage: DMA issue')
seg(1, 2, 'probed stage: MFMA loop')
seg(2, 3, 'probed stage: wait + barrier')
seg(7, 3, 'probed stage: total')
seg(3, 4, 'after probed stage -> stages done')
seg(4, 5, 'main store')
seg(5, 6, 'pool + final')
seg(0, 6, 'block total')
# skew: per workgroup, spread of the waves' arrival at the barrier of the probed stage
arr = raw[:, :, 2]; print('barrier arrival spread per WG (max-min), median', np.median(arr.max(1) - arr.min(1)))
st = raw[:, :, 1]; print('loop start spread per WG (max-min), median', np.median(st.max(1) - st.min(1)))
