"""Label exactness at FULL tile size for the models of BASELINE.json's GPU configs against the fp32 oracle:
configs[1] semseg_spine (bf16 and fp16), configs[2] semseg_axon (bf16, one 128^3 tile of its chunk geometry) and
configs[4] mivcsj (fp16, GroupNorm) -- the weights and synthetic EM tiles bench.py uses.

north_star: "argmax labels bit-exact vs reference"; the reference's own rule is the threshold on
``uint8(floor(255*softmax))`` (/root/reference/syconn/handler/prediction.py:813-833, 864-865).  With bf16 / fp16 storage
bit-exactness can only hold where the oracle's decision margin exceeds the numeric error of the storage type, so every
disagreement is split into margin-safe (asserted: none) and margin-unsafe (counted and printed, bounded).  The margin is A
PRIORI: ``TOL_LOGIT_REL[act] * max|oracle logit|`` with constants fixed in oracle/label_margin.py -- the measured error is
asserted to stay below that tolerance separately, so "no safe mismatch" is a statement about the kernels, not a tautology.
(The reference-precision mode act_dtype='f32' has its own full-size test in tests/test_gpu_f32.py.)
"""
import pytest
import torch

from bench import BENCH_FINAL_SCALE, synthetic_em_tiles
from oracle.label_margin import TOL_LOGIT_REL, label_split
from oracle.unet_ref import ARCHS, UNet

pytestmark = pytest.mark.gpu

# bounds on the fraction of voxels whose oracle value lies inside the STATED tolerance of a decision boundary (these may
# legitimately differ; measured with the tolerance measured on the tensors in round 2: threshold rule 5.3e-2 / 6.6e-3, argmax
# 1.2e-2 / 1.3e-3 -- the a-priori tolerance is ~1.4x wider) and on the labels that really differ (measured: 4.5e-3 / 5.8e-4
# of the voxels for the threshold rule, 5e-4 / 6e-5 for argmax)
MAX_UNSAFE_FRAC = {'bf16': (0.12, 0.03), 'f16': (0.02, 0.005)}
MIN_AGREEMENT = {'bf16': (0.993, 0.999), 'f16': (0.999, 0.9998)}
# median top-2 logit margin of the oracle / stated tolerance (scale-invariant for a ReLU network)
MIN_MARGIN_OVER_TOL = {'bf16': 8.0, 'f16': 60.0}


@pytest.mark.parametrize('arch,act', [('semseg_spine', 'bf16'), ('semseg_spine', 'f16'), ('semseg_axon', 'bf16'), ('mivcsj', 'f16')])
def test_full_size_tile_labels_vs_fp32_oracle(gpu, arch, act):
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    sd = random_state_dict(arch, seed=0, final_scale=BENCH_FINAL_SCALE)
    ref_net = UNet(in_channels=1, **ARCHS[arch]).eval()
    ref_net.load_state_dict(sd)
    raw = torch.from_numpy(synthetic_em_tiles(1, 128, seed=1))            # bench.py's rank-0 tiles
    with torch.no_grad():
        ref_logits = ref_net((raw[0].float() / 255.)[None, None])[0]
    dm = DenseModel(sd, act_dtype=act, device=gpu)
    ids = list(range(1, dm.out_channels))
    thr_u8 = [127.5] * len(ids)
    x = raw.to(gpu)
    lg = dm.forward_batch(x, L.SD_OUT_LOGITS_F32)[0].cpu()
    pr = dm.forward_batch(x, L.SD_OUT_PROBS_F32)[0].cpu()
    lab = dm.forward_labels_batch(x, ids, thr_u8)[0].cpu()
    r = label_split(ref_logits, lg, pr, lab, ids, [None] * dm.out_channels, TOL_LOGIT_REL[act])
    print(f'\n[{arch} {act}] 128^3 vs fp32 oracle: ' + ', '.join(f'{k}={v:.4g}' for k, v in r.items()))
    assert r['logit_err_max_rel'] <= TOL_LOGIT_REL[act], r
    assert r['label_mismatch_safe'] == 0, r          # threshold rule of the reference: exact wherever it can be
    assert r['argmax_mismatch_safe'] == 0, r         # argmax: exact wherever the fp32 margin exceeds the stated tolerance
    assert r['label_unsafe_frac'] <= MAX_UNSAFE_FRAC[act][0] and r['argmax_unsafe_frac'] <= MAX_UNSAFE_FRAC[act][1], r
    assert r['label_agreement'] >= MIN_AGREEMENT[act][0] and r['argmax_agreement'] >= MIN_AGREEMENT[act][1], r
    # the workload is meaningful: several classes are really predicted, and most voxels carry a decisive margin
    assert len(torch.unique(lab)) >= 3
    assert r['median_top2_margin_over_tol'] >= MIN_MARGIN_OVER_TOL[act], r
