# dev tool: GroupNorm network (mivcsj), default fusions vs SD_NO_FUSE: logits and (with SD_KEEP_ALL) every activation buffer
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
net = build_unet('mivcsj', seed=3, final_scale=4.0)
x = torch.randint(0, 256, (2, 37, 63, 33), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
for keep in (False, True):
    if keep: os.environ['SD_KEEP_ALL'] = '1'
    os.environ['SD_NO_FUSE'] = '1'
    a = DenseModel(net, 'bf16', torch.device('cuda', 0))
    del os.environ['SD_NO_FUSE']
    b = DenseModel(net, 'bf16', torch.device('cuda', 0))
    for n in (1, 2):
        ya = a.forward_batch(x[:n], L.SD_OUT_LOGITS_F32); yb = b.forward_batch(x[:n], L.SD_OUT_LOGITS_F32, slot=1)
        print('keep_all', keep, 'N', n, 'logit max abs diff %.3e of max %.2f' % (float((ya - yb).abs().max()), float(ya.abs().max())))
    if keep:
        ya = a.forward(x[0], L.SD_OUT_LOGITS_F32); yb = b.forward(x[0], L.SD_OUT_LOGITS_F32)
        for buf in range(1, a.info['n_buffers']):
            ta, tb = a.read_buffer(buf), b.read_buffer(buf)
            d = (ta - tb).abs()
            if float(d.max()) > 0: print(' buffer', buf, tuple(ta.shape), 'max abs diff %.3e' % float(d.max()), 'frac differing %.2e' % float((d > 0).float().mean()))
