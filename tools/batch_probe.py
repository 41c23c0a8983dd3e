# dev probe: tile throughput of sd_forward_batch vs batch size (and optionally over 2 streams)
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd import _lib as L
from syconn_amd.engine import DenseModel, StreamRing
arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
net = random_state_dict(arch, seed=0)
dm = DenseModel(net, 'bf16', torch.device('cuda', 0))
for N in [int(v) for v in os.environ.get('SD_PROBE_BATCHES', '1,2,4,8').split(',')]:
    for ns in (1, 2):
        ring = StreamRing(torch.device('cuda', 0), ns)
        x = torch.randint(0, 256, (N, S, S, S), dtype=torch.uint8, device='cuda')
        outs = [torch.empty((N, dm.out_channels, S, S, S), dtype=torch.uint8, device='cuda') for _ in range(ns)]
        def run(n):
            with ring:
                for i in range(n):
                    with ring.stream(i):
                        dm.forward_batch(x, L.SD_OUT_PROBS_U8, outs[ring.slot(i)], slot=ring.slot(i))
        reps = max(8, 96 // N)
        run(4); torch.cuda.synchronize()
        t = time.perf_counter(); run(reps); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / (reps * N)
        print(f'{arch} {S}^3: batch {N}, {ns} stream(s): {dt * 1e3:.3f} ms/tile -> {S**3 / dt / 1e6:.1f} Mvox/s')
