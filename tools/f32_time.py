"""Time per 128^3 tile of the reference-precision mode (act_dtype='f32') next to bf16 / f16, semseg_spine (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_em_tiles                                    # noqa: E402
from syconn_amd import _lib as L                                        # noqa: E402
from syconn_amd.cnn import random_state_dict                            # noqa: E402
from syconn_amd.engine import DenseModel                                # noqa: E402

dev = torch.device('cuda', 0)
arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
x = torch.from_numpy(synthetic_em_tiles(2, 128, seed=1)).to(dev)
sd = random_state_dict(arch, seed=0, final_scale=8.0)
for act in ('bf16', 'f16', 'f32'):
    dm = DenseModel(sd, act_dtype=act, device=dev)
    dm.forward_batch(x, L.SD_OUT_PROBS_U8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        dm.forward_batch(x, L.SD_OUT_PROBS_U8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n / x.shape[0]
    print(f'{arch} {act}: {dt * 1e3:.2f} ms per 128^3 tile, workspace {dm.workspace_bytes((128, 128, 128)) / 2**30:.2f} GiB')
    if act == 'f32':
        dm.profile(1)
        dm.forward_batch(x[:1], L.SD_OUT_PROBS_U8)
        ms = dm.profile_read(0)
        print('  per op ms:', ' '.join(f'{v:.2f}' for v in ms))
    del dm
