"""Whole watershed branch (object_segmentation_first_stage, device resident) on the two 512^3 volumes of tools/segbench.py, timed
back to back; arguments are passed on as SD_WS_NT (only honoured by builds that still have that switch) -- used for the A/B runs of
the flood kernel's workgroup size.  usage: python3 tools/ws_ab.py 1024 1024"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy import ndimage
from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
n = 512
rng = np.random.default_rng(0)
prob = ndimage.gaussian_filter(rng.random((n // 4,) * 3).astype(np.float32), 1.5)
prob = np.kron(prob, np.ones((4, 4, 4), np.float32))[:n, :n, :n]
prob = ((prob - prob.min()) / (prob.max() - prob.min()) * 255).astype(np.uint8)
thr = float(np.quantile(prob[::4, ::4, ::4], 0.9))
sph = np.zeros((n, n, n), np.uint8)
r2 = np.random.default_rng(5)
for _ in range(3000):
    c = r2.integers(16, n - 16, 3); r = r2.integers(5, 15)
    if r2.random() < 0.35:
        c2 = np.clip(c + r2.integers(-r, r + 1, 3) * 1.4, 16, n - 17).astype(int); centres = (c, c2)
    else: centres = (c,)
    for cc in centres:
        lo, hi = np.maximum(cc - r, 0), np.minimum(cc + r + 1, n)
        g = np.ogrid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        sph[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]][((g[0] - cc[0]) ** 2 + (g[1] - cc[1]) ** 2 + ((g[2] - cc[2]) * 1.6) ** 2) <= r * r] = 255
ops = ['binary_opening', 'binary_closing', 'binary_erosion']
vols = {'field': (torch.from_numpy(prob).cuda(), thr), 'organelle': (torch.from_numpy(sph).cuda(), 127.5)}
ref = {}
for nt in sys.argv[1:]:
    os.environ['SD_WS_NT'] = nt
    for name, (p, t) in vols.items():
        f = lambda: object_segmentation_first_stage(p, t, ops, return_device=True, min_seed_vx=10)
        lab = f()[0]; torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if name not in ref: ref[name] = lab.clone()
        print(f'NT={nt} {name}: {dt*1e3:.2f} ms equal={bool(torch.equal(lab, ref[name]))}', flush=True)
