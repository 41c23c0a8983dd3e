# dev helper (needs a -DSD_TIMING -DSD_RT build): absolute start/end of every workgroup of one conv op + the CU it ran on
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet   # seeded random weights (no trained models exist)
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
op = int(sys.argv[1]); os.environ['SD_TIMING_OP'] = str(op)
waves = int(sys.argv[2]); nwg = int(sys.argv[3])
dm = DenseModel(build_unet('semseg_spine', seed=0), 'bf16', torch.device('cuda', 0))
x = torch.randint(0, 256, (128, 128, 128), dtype=torch.uint8, device='cuda')
out = torch.empty((5, 128, 128, 128), dtype=torch.uint8, device='cuda')
for _ in range(int(os.environ.get('SD_PROBE_ITERS', '300'))): dm.forward(x, L.SD_OUT_PROBS_U8, out)
torch.cuda.synchronize()
raw = dm._ws[65536:65536 + nwg * waves * 64].view(torch.int64).cpu().numpy().reshape(nwg, waves, 8)
st = raw[:, 0, 4]; en = raw[:, :, 5].max(1); hw = raw[:, 0, 6]
xcc = (hw >> 32) & 0xf; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 16 + cu
t0 = st.min()
print('workgroups', nwg, 'distinct CUs', len(np.unique(key)), 'kernel span (first start -> last end) %.1f us' % ((en.max() - t0) / 100.))
dur = (en - st) / 100.
print('block duration us: median %.2f p10 %.2f p90 %.2f' % (np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90)))
cnt = np.array([np.sum(key == k) for k in np.unique(key)])
print('blocks per CU: min %d max %d' % (cnt.min(), cnt.max()))
gaps = []
for k in np.unique(key):
    m = key == k; o = np.argsort(st[m]); s_, e_ = st[m][o], en[m][o]
    gaps += list((s_[1:] - e_[:-1]) / 100.)
gaps = np.array(gaps)
print('gap between consecutive blocks on a CU us: median %.2f p90 %.2f max %.2f' % (np.median(gaps), np.percentile(gaps, 90), gaps.max()))
print('first-start spread us: p50 %.2f p99 %.2f' % (np.percentile((st - t0) / 100., 50), np.percentile(np.sort((st - t0) / 100.)[:len(np.unique(key))], 99)))
print('busy fraction of CU-time inside the span: %.3f' % (dur.sum() / (len(np.unique(key)) * (en.max() - t0) / 100.)))
