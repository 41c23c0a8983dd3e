# dev helper: layer-wise comparison of the HIP path against the reduced-precision oracle emulation
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.unet_ref import build_unet, unet_forward_emulated
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
shape = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (40, 200, 240)
kw = dict(n_blocks=2, start_filts=32) if len(sys.argv) < 3 else eval(sys.argv[2])
arch = sys.argv[3] if len(sys.argv) > 3 else 'myelin'
model = build_unet(arch, seed=13, **kw)
g = torch.Generator().manual_seed(17)
raw = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
col = []
with torch.no_grad():
    emu = unet_forward_emulated(model, (raw.float() / 255.)[None, None], collect=col)[0]
dm = DenseModel(model, 'bf16', torch.device('cuda', 0))
out = dm.forward(raw.cuda(), L.SD_OUT_LOGITS_F32).cpu()
print('logits rel err', float((out - emu).abs().max() / emu.abs().max()))
for i, t in enumerate(col):
    got = dm.read_buffer(i + 1).cpu()
    t = t[0]
    gcrop = got[:, :t.shape[1], :t.shape[2], :t.shape[3]]
    d = (gcrop - t).abs()
    err = float(d.max()) / max(float(t.abs().max()), 1e-6)
    bad = (d > 0.05 * t.abs().max()).nonzero()
    msg = f'buffer {i+1} {tuple(t.shape)} rel err {err:.2e} nbad {len(bad)}'
    if len(bad):
        msg += f' first bad (c,z,y,x) {bad[0].tolist()} last {bad[-1].tolist()}'
        zs, ys, xs = bad[:, 1].unique(), bad[:, 2].unique(), bad[:, 3].unique()
        msg += f' z {zs[:6].tolist()}..{zs[-3:].tolist()} y {ys[:8].tolist()}..{ys[-3:].tolist()} x {xs[:8].tolist()}..{xs[-3:].tolist()}'
    print(msg)
d = (out - emu).abs().max(0).values
bad = (d > 0.05 * emu.abs().max()).nonzero()
print('logit bad voxels', len(bad), 'of', d.numel())
if len(bad):
    zs, ys, xs = bad[:, 0].unique(), bad[:, 1].unique(), bad[:, 2].unique()
    print(' z', zs[:10].tolist(), '..', zs[-3:].tolist(), len(zs))
    print(' y', ys[:40].tolist(), '..', ys[-3:].tolist(), len(ys))
    print(' x', xs[:40].tolist(), '..', xs[-3:].tolist(), len(xs))
    print(' first', bad[:5].tolist())
if len(bad):
    import collections
    print(' x%16', sorted(collections.Counter((bad[:, 2] % 16).tolist()).items()))
    print(' y%32', sorted(collections.Counter((bad[:, 1] % 32).tolist()).items()))
    print(' x//16', sorted(collections.Counter((bad[:, 2] // 16).tolist()).items()))
    print(' y//32', sorted(collections.Counter((bad[:, 1] // 32).tolist()).items()))
    out2 = dm.forward(raw.cuda(), L.SD_OUT_LOGITS_F32).cpu()
    print(' rerun identical:', bool(torch.equal(out, out2)), 'n different', int((out != out2).any(0).sum()))
    b0 = bad[0].tolist()
    print(' sample got', out[:, b0[0], b0[1], b0[2]].tolist(), 'want', emu[:, b0[0], b0[1], b0[2]].tolist())
