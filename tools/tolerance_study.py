"""GPU box: measured max |logit error| / max |oracle logit| of the fast plans on several tiles AND several weight seeds, next to the stated
a-priori tolerance (oracle/label_margin.py::stated_tolerance) -- how much room the rule has.  usage: tolerance_study.py [arch] [act]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import BENCH_FINAL_SCALE, synthetic_em_tiles
from oracle.label_margin import stated_tolerance
from oracle.unet_ref import ARCHS, UNet
from syconn_amd import _lib as L
from syconn_amd.cnn import random_state_dict
from syconn_amd.engine import DenseModel

arch = sys.argv[1] if len(sys.argv) > 1 else 'mivcsj'
act = sys.argv[2] if len(sys.argv) > 2 else 'f16'
dev = torch.device('cuda', 0)
tol = stated_tolerance(arch, act)
rows = []
for wseed, tseed in ((0, 1), (0, 2), (0, 3), (1, 1), (2, 1), (3, 4)):
    sd = random_state_dict(arch, seed=wseed, final_scale=BENCH_FINAL_SCALE)
    ref = UNet(in_channels=1, **ARCHS[arch]).eval(); ref.load_state_dict(sd)
    x = torch.from_numpy(synthetic_em_tiles(1, 128, seed=tseed))
    with torch.no_grad():
        want = ref((x[0].float() / 255.)[None, None])[0]
    got = DenseModel(sd, act_dtype=act, device=dev).forward_batch(x.to(dev), L.SD_OUT_LOGITS_F32)[0].cpu()
    err = float((got - want).abs().max()) / float(want.abs().max())
    rows.append(dict(weights_seed=wseed, tile_seed=tseed, err=err, frac_of_stated=err / tol))
    print(f'{arch} {act} weights seed {wseed} tile seed {tseed}: {err:.3e} = {err / tol:.2f} of the stated {tol:.3e}', flush=True)
print(json.dumps(dict(arch=arch, act=act, stated=tol, max=max(r['err'] for r in rows), rows=rows)))
