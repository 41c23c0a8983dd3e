"""Where the time of the level-synchronous marker flood goes (csrc/sd_objseg.hip::k_ws_flood): wall-clock stamps of thread 0 per
workgroup and phase, from a -DSD_WS_TIMING build of the library:
    cd syconn_amd/csrc && hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -DSD_WS_TIMING -c sd_objseg.hip -o /tmp/t.o \
      && hipcc --offload-arch=gfx950 -shared -fPIC -pthread sd_kernels.o sd_dec0.o sd_f32.o sd_api.o sd_segstats.o /tmp/t.o \
         sd_snappy.o sd_host.o -o ../libsd_wst.so
usage (GPU box): python3 tools/ws_timing.py [field]      (default: the organelle-like volume of tools/segbench.py)"""
import os, sys, ctypes as C, numpy as np, torch
os.environ['SD_LIB_NAME'] = 'libsd_wst.so'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd import _lib as L
from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
n = 512
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
thr = 127.5
if len(sys.argv) > 1 and sys.argv[1] == 'field':
    from scipy import ndimage
    rng = np.random.default_rng(0)
    prob = ndimage.gaussian_filter(rng.random((n // 4,) * 3).astype(np.float32), 1.5)
    prob = np.kron(prob, np.ones((4, 4, 4), np.float32))[:n, :n, :n]
    sph = ((prob - prob.min()) / (prob.max() - prob.min()) * 255).astype(np.uint8)
    thr = float(np.quantile(sph[::4, ::4, ::4], 0.9))
p = torch.from_numpy(sph).cuda()
object_segmentation_first_stage(p, thr, ops, return_device=True, min_seed_vx=10)
lib = L.load()
buf = np.zeros((2048, 16), np.uint64)
lib.sd_debug_ws_timing.argtypes = [C.c_void_p, C.c_int]
lib.sd_debug_ws_timing(None, 1)
object_segmentation_first_stage(p, thr, ops, return_device=True, min_seed_vx=10)
lib.sd_debug_ws_timing(buf.ctypes.data, 0)
names = ['far_rescan', 'level_split+sort', 'phase1', 'cascade', 'phase3', 'gen_sort', 'n_rescan', 'n_levels', 'n_gen', 'sum_nA', 'n_gen_casc', 'sum_nCL', 'p1_loads', 'p1_atomics', 'p1_slots', 'x']
tot = buf[:, :6].sum(1) + buf[:, 12:15].sum(1)
order = np.argsort(-tot.astype(np.int64))[:4]
print('wall clock ticks are 100 MHz: 100 ticks = 1 us')
for w in order:
    print('WG', w, 'total %.2f ms' % (tot[w] / 1e5), {k: int(v) for k, v in zip(names, buf[w])})
print('all WGs:', {k: int(v) for k, v in zip(names, buf.sum(0))})
