"""Split-fp16 plan (act_dtype='f16x2') against the fp32 oracle: logit error of all architectures on small tiles, then the
headline tile (semseg_spine 128^3): error, label agreement and time per tile next to f16.  Run on the GPU box."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle.unet_ref import ARCHS, build_unet, UNet
from oracle.label_margin import label_split
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel

gpu = torch.device('cuda:0')
cases = [('myelin', (8, 32, 48)), ('myelin', (13, 27, 29)), ('er', (8, 40, 72)), ('syntype', (16, 33, 47)),
         ('syntype_enh', (8, 24, 40)), ('mivcsj', (9, 35, 52)), ('semseg_spine', (6, 70, 130)), ('semseg_axon', (5, 17, 50)),
         ('golgi', (8, 16, 16))]
for arch, shape in cases:
    model = build_unet(arch, seed=5, final_scale=4.0)
    g = torch.Generator().manual_seed(3)
    raw = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    with torch.no_grad():
        ref = model((raw.float() / 255.)[None, None])[0]
    res = {}
    for act in ('f16x2', 'f32', 'f16'):
        dm = DenseModel(model, act_dtype=act, device=gpu)
        out = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
        res[act] = float((out - ref).abs().max()) / float(ref.abs().max())
        if act == 'f16x2':
            assert not dm.overflowed()
    print(f'{arch:13s} {str(shape):15s} ' + '  '.join(f'{k} {v:.2e}' for k, v in res.items()), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == 'small':
    sys.exit(0)
from bench import BENCH_FINAL_SCALE, synthetic_em_tiles
from syconn_amd.cnn import random_state_dict
for arch in ('semseg_spine',) + tuple(sys.argv[1:]):
    sd = random_state_dict(arch, seed=0, final_scale=BENCH_FINAL_SCALE)
    ref_net = UNet(in_channels=1, **ARCHS[arch]).eval()
    ref_net.load_state_dict(sd)
    raw = torch.from_numpy(synthetic_em_tiles(8, 128, seed=1))
    t0 = time.time()
    with torch.no_grad():
        ref_logits = ref_net((raw[0].float() / 255.)[None, None])[0]
    print(f'{arch}: oracle {time.time() - t0:.1f} s', flush=True)
    x = raw.to(gpu)
    for act in ('f16x2', 'f16', 'bf16'):
        dm = DenseModel(sd, act_dtype=act, device=gpu)
        ids = list(range(1, dm.out_channels))
        lg = dm.forward_batch(x[:1], L.SD_OUT_LOGITS_F32)[0].cpu()
        pr = dm.forward_batch(x[:1], L.SD_OUT_PROBS_F32)[0].cpu()
        lab = dm.forward_labels_batch(x[:1], ids, [127.5] * len(ids))[0].cpu()
        r = label_split(ref_logits, lg, pr, lab, ids, [None] * dm.out_channels, 2e-5 if act == 'f16x2' else 1e-2)
        for _ in range(2):
            dm.forward_labels_batch(x, ids, [127.5] * len(ids))
        torch.cuda.synchronize()
        t0 = time.time()
        n = 5
        for _ in range(n):
            dm.forward_labels_batch(x, ids, [127.5] * len(ids))
        torch.cuda.synchronize()
        ms = (time.time() - t0) / n / 8 * 1e3
        print(f'[{arch} {act}] {ms:.3f} ms per tile, overflow {dm.overflowed()}, logit_err_max_rel {r["logit_err_max_rel"]:.2e} '
              f'argmax_agreement {r["argmax_agreement"]:.7f} label_agreement {r["label_agreement"]:.7f}', flush=True)
        if act == 'f16x2':
            dm.profile(1)
            dm.forward_labels_batch(x, ids, [127.5] * len(ids))
            ms_op = dm.profile_read(0)
            for i, (k, t) in enumerate(zip(dm.op_kinds, ms_op)):
                print(f'   op {i:2d} kind {k} {t / 8 * 1e3:8.1f} us per tile')
