# dev tool: cycle stamps of one step of the fused level-0 decoder (library built with -DSD_DEC0_TIMING=<step>, SD_DEC0_DBG=1)
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SD_DEC0_DBG'] = '1'
from syconn_amd.cnn import random_state_dict
from syconn_amd.engine import DenseModel
dm = DenseModel(random_state_dict('semseg_spine', seed=0, final_scale=8.0), 'bf16', torch.device('cuda', 0))
x = torch.randint(0, 256, (8, 128, 128, 128), dtype=torch.uint8, device='cuda')
ids = list(range(1, dm.out_channels)); thr = [127.5] * len(ids)
for _ in range(3): dm.forward_labels_batch(x, ids, thr)
torch.cuda.synchronize()
ws = dm._ws
st = ws[:64 * 8 * 8 * 8].view(torch.int64).cpu().numpy().reshape(64, 8, 8)
names = ['start', 'dma/up done', 'mfma done', 'epilogue done', 'waitcnt done', 'barrier done', 'logits ready', 'softmax done']
for wg in (0, 1, 17, 40):
    print('workgroup', wg)
    t0 = st[wg, :, 0].min()
    for w in range(8):
        print('  wave', w, 'merge' if w < 4 else 'tail ', [int(st[wg, w, i] - t0) for i in range(8)])
d = st[:, :, 1:6] - st[:, :, 0:5]
print('mean phase cycles merge:', dict(zip(names[1:], d[:, :4].mean((0, 1)).round())))
print('mean phase cycles tail :', dict(zip(names[1:], d[:, 4:].mean((0, 1)).round())))
print('step length (start -> barrier done):', (st[:, :, 5] - st[:, :, 0]).mean().round())
