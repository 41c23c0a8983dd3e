# dev probe: throughput of alternating tiles over S HIP streams (S workspaces) vs one stream
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict as build_unet   # seeded random weights (no trained models exist)
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel
arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
net = build_unet(arch, seed=0)
for S in (1, 2, 3):
    dms = [DenseModel(net, 'bf16', torch.device('cuda', 0)) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    x = torch.randint(0, 256, (128, 128, 128), dtype=torch.uint8, device='cuda')
    outs = [torch.empty((dms[0].out_channels, 128, 128, 128), dtype=torch.uint8, device='cuda') for _ in range(S)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                dms[i % S].forward(x, L.SD_OUT_PROBS_U8, outs[i % S])
    run(16); torch.cuda.synchronize()
    t = time.perf_counter(); run(160); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f'{arch}: {S} stream(s): {dt / 160 * 1e3:.3f} ms/tile -> {128**3 / (dt / 160) / 1e6:.1f} Mvox/s')
    del dms
