# dev fuzz: parallel.predict_volume_distributed on the device path (volume resident in HBM, sd_tile_gather / sd_tile_scatter, slab
# downloads) with random volume / chunk / halo shapes and an identity-like predict_fn: the result must be the volume itself (n_out
# copies with a per-channel offset), whatever the geometry, pipelined or not, pinned or pageable input.  usage: fuzz_volume.py [seconds]
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd import parallel as par
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
dev = torch.device('cuda', 0)
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    vs = tuple(int(v) for v in rng.integers(1, 90, 3))
    cs = tuple(int(v) for v in rng.integers(1, 60, 3))
    ol = tuple(int(v) for v in rng.integers(0, 9, 3))
    n_out = int(rng.integers(1, 4))
    vol = torch.from_numpy(rng.integers(0, 256, vs, dtype=np.uint8))
    if rng.random() < 0.5:
        vol = vol.pin_memory()
    elif rng.random() < 0.3:
        vol = vol.to(dev)

    def predict_fn(ch, valid_box=None):
        core = ch[ol[0]:ch.shape[0] - ol[0], ol[1]:ch.shape[1] - ol[1], ol[2]:ch.shape[2] - ol[2]]
        # (the halo must hold the real neighbours: fold a neighbour voxel in where there is a halo)
        out = torch.stack([core + c for c in range(n_out)])
        if ol[2] > 0:
            out[0] = core ^ ch[ol[0]:ch.shape[0] - ol[0], ol[1]:ch.shape[1] - ol[1], ol[2] - 1:ch.shape[2] - ol[2] - 1]
        return out
    pipelined = bool(rng.random() < 0.7)
    got = par.predict_volume_distributed(vol, vs, cs, ol, predict_fn, n_out=n_out, device=dev, pipelined=pipelined)
    v = vol.cpu()
    want = torch.stack([v + c for c in range(n_out)])
    if ol[2] > 0:
        left = torch.zeros_like(v); left[:, :, 1:] = v[:, :, :-1]
        want[0] = v ^ left
    if not torch.equal(got, want):
        print('MISMATCH', vs, cs, ol, n_out, pipelined); sys.exit(1)
    n += 1
assert par.HOST_BOX_COPIES == 0
print(f'fuzz_volume: {n} cases ok in {budget:.0f} s, host box copies {par.HOST_BOX_COPIES}')
