# dev probe: workspace bytes per tile with and without lifetime-based buffer reuse (run twice: SD_NO_WS_REUSE=1 / unset)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syconn_amd.cnn import random_state_dict
from syconn_amd.engine import DenseModel
for arch in ('myelin', 'semseg_axon', 'mivcsj'):
    dm = DenseModel(random_state_dict(arch, 0), 'bf16', torch.device('cuda', 0))
    for shape in ((128, 128, 128), (178, 243, 331)):
        print(f'{arch:12s} {shape}: {dm.workspace_bytes(shape) / 2**30:6.2f} GiB  (reuse {"off" if os.environ.get("SD_NO_WS_REUSE") else "on"})')
