# dev tool: randomized parity sweep -- random architectures / tile shapes / batch sizes / dtypes against the fp32 oracle
# (tolerances of tests/test_gpu_unet.py) and against the unfused launch sequence (same rounded activations; the fused final
# layer evaluates the 1x1x1 conv as hi+lo MFMAs instead of fp32 FMAs -> logits agree to ~1e-5 relative, not bitwise; nets whose
# level-0 decoder runs as the streaming kernel sd_dec0.hip sum in another order -> roundings of stored activations flip).
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.unet_ref import ARCHS, build_unet
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {'bf16': (5e-2, 2e-2), 'f16': (8e-3, 4e-3)}
bad = 0
for k in range(n):
    arch = list(ARCHS)[rng.integers(len(ARCHS))]
    act = 'bf16' if rng.random() < 0.7 else 'f16'
    nb = 1 + int(rng.integers(3))
    if os.environ.get('SD_FUZZ_BIG'):      # shapes with enough blocks for the persistent / 4-tile kernel forms
        shape = (int(rng.choice([16, 24, 32, 40, 48, 64, 100])), int(rng.integers(90, 200)), int(rng.integers(90, 200)))
        nb = 1 + int(rng.integers(2))
    else:
        shape = (int(rng.integers(1, 40)), int(rng.integers(1, 150)), int(rng.integers(1, 150)))
    net = build_unet(arch, seed=int(rng.integers(1000)), final_scale=4.0)
    g = torch.Generator().manual_seed(k)
    x = torch.randint(0, 256, (nb, *shape), dtype=torch.uint8, generator=g)
    with torch.no_grad():
        ref = net((x.float() / 255.)[:, None])
    os.environ.pop('SD_NO_FUSE', None)
    dm = DenseModel(net, act, torch.device('cuda', 0))
    got = dm.forward_batch(x.cuda(), L.SD_OUT_LOGITS_F32).cpu()
    os.environ['SD_NO_FUSE'] = '1'
    dm2 = DenseModel(net, act, torch.device('cuda', 0))
    got2 = dm2.forward_batch(x.cuda(), L.SD_OUT_LOGITS_F32).cpu()
    os.environ.pop('SD_NO_FUSE', None)
    scale = float(ref.abs().max()) + 1e-12
    e_max = float((got - ref).abs().max()) / scale
    e_rms = float(((got - ref) ** 2).mean().sqrt() / ((ref ** 2).mean().sqrt() + 1e-12))
    d12 = float((got - got2).abs().max()) / scale
    # GroupNorm: statistics summed in another order -> scale / shift differ in the last bit -> roundings of stored activations flip
    loose = ARCHS[arch].get('normalization') == 'group8' or (ARCHS[arch]['start_filts'] <= 32 and shape[1] >= 8)
    same = d12 < ((2e-2 if act == 'bf16' else 3e-3) if loose else 1e-4)
    ok = e_max <= TOL[act][0] * 1.5 and e_rms <= TOL[act][1] * 1.5 and same and bool(torch.isfinite(got).all())
    bad += not ok
    print(f'{k:3d} {arch:13s} {act} N={nb} {str(shape):16s} max {e_max:.2e} rms {e_rms:.2e} fused-unfused {d12:.1e} {"ok" if ok else "FAIL"}')
print('failures:', bad)
