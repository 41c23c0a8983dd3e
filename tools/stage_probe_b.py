# dev helper (needs -DSD_TIMING -DSD_STAGES): duration of every stage of the probed block of one conv op, batched launch
# usage: stage_probe_b.py <op> <waves> <nwg> <nstages> [batch]   (SD_NO_WS_REUSE=1 keeps the stamp area unshared)
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SD_NO_WS_REUSE', '1')
from syconn_amd.cnn import random_state_dict as build_unet
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
op = int(sys.argv[1]); os.environ['SD_TIMING_OP'] = str(op)
waves = int(sys.argv[2]); nwg = int(sys.argv[3]); nst = int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dm = DenseModel(build_unet(os.environ.get('SD_ARCH', 'semseg_spine'), seed=0), os.environ.get('SD_ACT', 'bf16'), torch.device('cuda', 0))
x = torch.randint(0, 256, (B, 128, 128, 128), dtype=torch.uint8, device='cuda')
out = torch.empty((B, dm.out_channels, 128, 128, 128), dtype=torch.uint8, device='cuda')
for _ in range(30): dm.forward_batch(x, L.SD_OUT_PROBS_U8, out)
torch.cuda.synchronize()
base = 65536 + (1 << 20) * 8
raw = dm._ws[base:base + nwg * waves * 128].view(torch.int64).cpu().numpy().reshape(nwg, waves, 16)
t8 = dm._ws[65536:65536 + nwg * waves * 64].view(torch.int64).cpu().numpy().reshape(nwg, waves, 8)
start = raw[:, :, 14]
ends = raw[:, :, :min(nst, 14)]
prev = start
for s in range(min(nst, 14)):
    d = (ends[:, :, s] - prev).astype(np.float64)
    print(f'stage {s:2d}: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}')
    prev = ends[:, :, s]
if nst <= 14:
    d = (raw[:, :, 15] - prev).astype(np.float64)
    print(f'epilogue: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}')
print('block total median', np.median(raw[:, :, 15] - start))
# tstamp: 0 block start, 7 start of stage SD_TS, 1 after DMA issue, 2 after MFMAs, 3 after barrier, 4 all stages done, 5 main store done, 6 epilogue done
for a, b, nm in ((0, 4, 'stages'), (4, 5, 'pack+store'), (5, 6, 'rest of epilogue'), (7, 1, 'dma issue (stage TS)'), (1, 2, 'mfma (stage TS)'), (2, 3, 'wait+barrier (stage TS)')):
    d = (t8[:, :, b] - t8[:, :, a]).astype(np.float64)
    print(f'{nm:28s}: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}')

# per-group timeline (waves 0..w/2-1 = "late DMA / early epilogue" group A, the rest = group B), relative to the earliest block start of the workgroup
t0 = raw[:, :, 14].min(axis=1, keepdims=True)
for g, sl in (('A', slice(0, waves // 2)), ('B', slice(waves // 2, waves))):
    rel = (raw[:, sl, :] - t0[:, :, None]).astype(np.float64)
    line = ' '.join(f'{np.median(rel[:, :, k]):7.0f}' for k in [14] + list(range(min(nst, 14))) + [15])
    print(f'group {g}: start, stage ends..., epilogue end: {line}')

# phases of the probed stage (SD_TS) per group, relative to the stage start of the earliest wave of the workgroup
t7 = t8[:, :, 7].min(axis=1, keepdims=True)
for g, sl in (('A', slice(0, waves // 2)), ('B', slice(waves // 2, waves))):
    rel = (t8[:, sl, :] - t7[:, :, None]).astype(np.float64)
    print(f'group {g}: stage start {np.median(rel[:, :, 7]):6.0f}  dma issued {np.median(rel[:, :, 1]):6.0f}  mfma done {np.median(rel[:, :, 2]):6.0f}  barrier passed {np.median(rel[:, :, 3]):6.0f}')
