"""Label-exactness study at the headline configuration (GPU box): HIP path in bf16 and fp16 storage against the fp32
oracle on bench.py's tiles and weights, for several scales of the final 1x1x1 weights.  Prints one JSON line per case
(margin-safe / margin-unsafe split of oracle/label_margin.py) and writes them to gpurun_out/label_study.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_em_tiles                                    # noqa: E402
from oracle.label_margin import TOL_LOGIT_REL, label_split, merge_splits              # noqa: E402
from oracle.unet_ref import ARCHS, UNet                                 # noqa: E402
from syconn_amd import _lib as L                                        # noqa: E402
from syconn_amd.cnn import random_state_dict                            # noqa: E402
from syconn_amd.engine import DenseModel                                # noqa: E402


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'semseg_spine'
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device('cuda', 0)
    tiles = torch.from_numpy(synthetic_em_tiles(ntiles, 128, seed=1))
    out = []
    for scale in (4.0, 8.0, 16.0):
        sd = random_state_dict(arch, seed=0, final_scale=scale)
        ref = UNet(in_channels=1, **ARCHS[arch]).eval()
        ref.load_state_dict(sd)
        with torch.no_grad():
            ref_logits = [ref((tiles[i].float() / 255.)[None, None])[0] for i in range(ntiles)]
        for act in ('bf16', 'f16', 'f32'):
            dm = DenseModel(sd, act_dtype=act, device=dev)
            ids = list(range(1, dm.out_channels))
            parts = []
            for i in range(ntiles):
                x = tiles[i:i + 1].to(dev)
                parts.append(label_split(ref_logits[i], dm.forward_batch(x, L.SD_OUT_LOGITS_F32)[0].cpu(),
                                         dm.forward_batch(x, L.SD_OUT_PROBS_F32)[0].cpu(),
                                         dm.forward_labels_batch(x, ids, [127.5] * len(ids))[0].cpu(), ids,
                                         [None] * dm.out_channels, TOL_LOGIT_REL[act]))
            r = dict(arch=arch, act=act, final_scale=scale, tiles=ntiles, **merge_splits(parts))
            print(json.dumps(r), flush=True)
            out.append(r)
            del dm
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/label_study.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
