# dev probe: time of a launch set of 8 x 128^3 tiles with and without the output box of the 128^3-tile geometry (core 112 x 96 x 96)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd import _lib as L
from syconn_amd.cnn import random_state_dict
from syconn_amd.engine import DenseModel
dev = torch.device('cuda', 0)
x = torch.randint(0, 256, (8, 128, 128, 128), dtype=torch.uint8, device=dev)
roi = ((8, 16, 16), (120, 112, 112))
def t(dm, r):
    for _ in range(2): dm.forward_batch(x, L.SD_OUT_PROBS_U8, roi=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): dm.forward_batch(x, L.SD_OUT_PROBS_U8, roi=r)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / 8
for arch in ('myelin', 'semseg_spine', 'semseg_axon', 'er', 'syntype', 'syntype_enh', 'mivcsj'):
    for act in ('bf16', 'f16x2'):
        dm = DenseModel(random_state_dict(arch, seed=0), act, dev)
        a, b = t(dm, None), t(dm, roi)
        print(f'{arch:13s} {act:6s} whole {a:.3f} ms  box {b:.3f} ms  ({b / a:.2f})')
