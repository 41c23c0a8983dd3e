"""GPU box: the fused level-0 decoder with 56-column strips (PW = 60) against 64-column strips (PW = 68) and against the separate layers:
bit-identical uint8 probabilities / labels on widths that pick either, and the decoder's time on the reference tile 178 x 243 x 331."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd import _lib as L                                        # noqa: E402
from syconn_amd.cnn import random_state_dict                            # noqa: E402
from syconn_amd.engine import DenseModel                                # noqa: E402

dev = torch.device('cuda', 0)
sd = random_state_dict('semseg_spine', seed=0, final_scale=8.0)
fused = DenseModel(sd, 'bf16', dev)
os.environ['SD_NO_DEC0'] = '1'
layers = DenseModel(sd, 'bf16', dev)
os.environ.pop('SD_NO_DEC0')
ids, thr = [1, 2, 3, 4], [127.5] * 4
bad = 0
for shape in ((6, 40, 331), (5, 33, 166), (4, 24, 112), (3, 50, 57), (2, 18, 200), (7, 26, 280), (3, 21, 390), (4, 16, 128), (2, 30, 56), (3, 9, 120)):
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randint(0, 256, (2, *shape), dtype=torch.uint8, generator=g).to(dev)
    ref = layers.forward_batch(x, L.SD_OUT_PROBS_U8).clone()
    ref_l = layers.forward_labels_batch(x, ids, thr).clone()
    res = {}
    for pw in ('', '68', '60'):
        if pw:
            os.environ['SD_DEC0_PW'] = pw
        res[pw] = (fused.forward_batch(x, L.SD_OUT_PROBS_U8).clone(), fused.forward_labels_batch(x, ids, thr).clone(),
                   fused.forward_batch(x, L.SD_OUT_LOGITS_F32).clone())
        os.environ.pop('SD_DEC0_PW', None)
    ok = all(torch.equal(r[0], ref) and torch.equal(r[1], ref_l) for r in res.values()) and torch.equal(res['68'][2], res['60'][2])
    bad += not ok
    print(shape, 'kernel:', fused.op_kernels()[-1][1], '|', layers.op_kernels()[-1][1], 'OK' if ok else 'MISMATCH')
print('mismatching shapes:', bad)
x = torch.randint(0, 256, (3, 178, 243, 331), dtype=torch.uint8, device=dev)
for rep in range(2):
    for pw in ('68', '60', ''):
        if pw:
            os.environ['SD_DEC0_PW'] = pw
        for _ in range(2):
            fused.forward_labels_batch(x, ids, thr)
        torch.cuda.synchronize()
        fused.profile(4)
        for _ in range(4):
            fused.forward_labels_batch(x, ids, thr)
        us = sum(fused.profile_read(k) for k in range(4)) / 4 / 3 * 1e3
        fused.profile(0)
        os.environ.pop('SD_DEC0_PW', None)
        print(f'178x243x331, 3 tiles, SD_DEC0_PW={pw or "(auto)"}: tile {us.sum():.1f} us, last op (k_dec0) {us[us > 0][-1]:.1f} us')
