"""Split-fp16 reference-precision plan (act_dtype='f16x2', syconn_amd/csrc/sd_split.hip + k_conv_mfma MODE 3) against the
fp32 torch-CPU oracle.

The reference runs the U-Net in fp32 (/root/reference/syconn/handler/prediction.py:777-779: `float16` is never set).  In this plan
every activation and weight is two fp16 numbers hi + lo (22+ mantissa bits) and every product three fp16 MFMA passes with fp32
accumulation, so HIP path and oracle differ by fp32-level rounding only.  Stated tolerance: max |logit error| <= 1e-5 of the
largest |logit| (oracle/label_margin.py; measured 0.7 - 4.6e-6); labels equal except where the oracle's own value sits within that
tolerance of a decision boundary."""
import os

import numpy as np
import pytest
import torch

from oracle.label_margin import TOL_LOGIT_REL, label_split
from oracle.unet_ref import ARCHS, build_cnn3, build_unet

pytestmark = pytest.mark.gpu

TOL = TOL_LOGIT_REL['f16x2']


def _input(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)


@pytest.mark.parametrize('arch,shape', [('myelin', (8, 32, 48)), ('myelin', (13, 27, 29)), ('er', (8, 40, 72)),
                                        ('syntype', (16, 33, 47)), ('syntype_enh', (8, 24, 40)), ('mivcsj', (9, 35, 52)),
                                        ('semseg_spine', (6, 70, 130)), ('semseg_axon', (5, 17, 50)), ('golgi', (8, 16, 16)),
                                        ('myelin', (16, 64, 96))])
def test_split_logits_match_fp32_oracle(gpu, arch, shape):
    """All 8 architectures of the path, even and odd tiles (ceil-mode pooling + autocrop), BatchNorm and GroupNorm families."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_unet(arch, seed=5, final_scale=4.0)
    raw = _input(shape, 3)
    with torch.no_grad():
        ref = model((raw.float() / 255.)[None, None])[0]
    dm = DenseModel(model, act_dtype='f16x2', device=gpu)
    out = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    print(f'{arch} {shape}: max logit err / max logit = {err:.2e}')
    assert err <= TOL, (arch, shape, err)
    assert not dm.overflowed()
    # float32 input == uint8 input (normalisation float32(v)/255, prediction.py:808), bit for bit
    out_f = dm.forward((raw.float() / 255.).to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    assert torch.equal(out, out_f)
    pr = dm.forward(raw.to(gpu), L.SD_OUT_PROBS_F32).cpu()
    assert float((pr - ref.softmax(0)).abs().max()) <= TOL * float(ref.abs().max()) + 1e-6     # |dp| <= |dlogit|
    u8 = dm.forward(raw.to(gpu), L.SD_OUT_PROBS_U8).cpu()
    ref_u8 = torch.from_numpy((ref.softmax(0).numpy() * 255).astype(np.uint8))          # prediction.py:864-865
    d = (u8.int() - ref_u8.int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3, (int(d.max()), float((d > 0).float().mean()))


def test_split_layerwise_buffers_match_oracle(gpu, monkeypatch):
    """Every activation buffer of the split plan (hi + lo planes read back as their exact sum) against the oracle's tensor of the
    same layer, GroupNorm network included (statistics over the autocropped region of up-convolution outputs)."""
    from oracle.unet_ref import unet_forward_emulated
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    monkeypatch.setenv('SD_KEEP_ALL', '1')
    for arch, shape in (('mivcsj', (4, 35, 38)), ('myelin', (6, 21, 45))):
        model = build_unet(arch, seed=2, final_scale=2.0)
        raw = _input(shape, 9)
        col = []
        with torch.no_grad():
            unet_forward_emulated(model, (raw.float() / 255.)[None, None], dtype=torch.float32, collect=col)
        dm = DenseModel(model, act_dtype='f16x2', device=gpu)
        dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32)
        for i, t in enumerate(col):
            got = dm.read_buffer(i + 1).cpu()
            t = t[0]
            g = got[:, :t.shape[1], :t.shape[2], :t.shape[3]]
            err = float((g - t).abs().max()) / max(float(t.abs().max()), 1e-6)
            assert err <= 1e-5, (arch, i + 1, tuple(t.shape), err)


def test_split_batch_and_labels(gpu):
    """Batched launch == single forwards (bit for bit); label output == the label rule on the uint8 probabilities; the
    3-layer CNN of BASELINE configs[0] reproduces the oracle's uint8 output up to single-level flips at truncation boundaries."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel, postproc_labels
    model = build_unet('semseg_spine', seed=1, final_scale=6.0)
    dm = DenseModel(model, act_dtype='f16x2', device=gpu)
    x = _input((3, 6, 40, 70), 5).to(gpu)
    a = dm.forward_batch(x, L.SD_OUT_LOGITS_F32)
    for i in range(3):
        assert torch.equal(a[i], dm.forward(x[i], L.SD_OUT_LOGITS_F32, slot=1))
    probs = dm.forward_batch(x, L.SD_OUT_PROBS_U8)
    ids, thr = (4, 2, 1, 3), (40.0, 60.5, 30.0, 80.0)
    want = torch.stack([postproc_labels(probs[i], list(ids), list(thr)) for i in range(3)])
    assert torch.equal(dm.forward_labels_batch(x, ids, thr), want)
    cnn = build_cnn3(seed=0)
    raw = _input((16, 40, 40), 0)
    with torch.no_grad():
        ref = cnn((raw.float() / 255.)[None, None])[0].softmax(0)
    u8 = DenseModel(cnn, act_dtype='f16x2', device=gpu).forward(raw.to(gpu), L.SD_OUT_PROBS_U8).cpu()
    d = (u8.int() - torch.from_numpy((ref.numpy() * 255).astype(np.uint8)).int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3


@pytest.mark.parametrize('arch', ['semseg_spine', 'semseg_axon', 'mivcsj'])
def test_split_full_size_tile_labels(gpu, arch):
    """The models of BASELINE configs[1], [2], [4] at 128^3 (bench.py's weights and tile) in the reference-precision plan: logit
    error within the stated 1e-5, argmax agreement with the fp32 oracle >= 0.99999 and threshold-rule agreement >= 0.9999
    (VERDICT r3 item 1), no mismatch outside the a-priori margin -- for EVERY model, no per-architecture tolerance."""
    from bench import BENCH_FINAL_SCALE, synthetic_em_tiles
    from oracle.unet_ref import UNet
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    sd = random_state_dict(arch, seed=0, final_scale=BENCH_FINAL_SCALE)
    ref_net = UNet(in_channels=1, **ARCHS[arch]).eval()
    ref_net.load_state_dict(sd)
    raw = torch.from_numpy(synthetic_em_tiles(1, 128, seed=1))
    with torch.no_grad():
        ref_logits = ref_net((raw[0].float() / 255.)[None, None])[0]
    dm = DenseModel(sd, act_dtype='f16x2', device=gpu)
    ids = list(range(1, dm.out_channels))
    x = raw.to(gpu)
    lg = dm.forward_batch(x, L.SD_OUT_LOGITS_F32)[0].cpu()
    pr = dm.forward_batch(x, L.SD_OUT_PROBS_F32)[0].cpu()
    lab = dm.forward_labels_batch(x, ids, [127.5] * len(ids))[0].cpu()
    assert not dm.overflowed()
    r = label_split(ref_logits, lg, pr, lab, ids, [None] * dm.out_channels, TOL)
    print(f'\n[f16x2] 128^3 {arch} vs fp32 oracle: ' + ', '.join(f'{k}={v:.4g}' for k, v in r.items()))
    assert r['logit_err_max_rel'] <= TOL, r
    assert r['argmax_agreement'] >= 0.99999 and r['label_agreement'] >= 0.9999, r
    assert r['argmax_mismatch_safe'] == 0 and r['label_mismatch_safe'] == 0, r
    assert r['label_unsafe_frac'] <= 1e-3 and r['argmax_unsafe_frac'] <= 1e-3, r      # (bf16: 17 % of the tile is margin-unsafe)


def _blow_up(model, factor=2e5):
    with torch.no_grad():
        model.down_convs[1].conv2.weight.mul_(factor)
    return model


@pytest.mark.parametrize('arch,shape', [('myelin', (4, 40, 64)), ('mivcsj', (4, 35, 38))])
def test_split_range_guard(gpu, arch, shape):
    """The hi planes are fp16: a value above 65504 raises the device flag in the split plan as in the fp16 plan."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    x = _input(shape, 1).to(gpu)
    ok = DenseModel(build_unet(arch, seed=3, final_scale=4.0), act_dtype='f16x2', device=gpu)
    ok.forward(x, L.SD_OUT_PROBS_U8)
    assert not ok.overflowed()
    bad = DenseModel(_blow_up(build_unet(arch, seed=3, final_scale=4.0)), act_dtype='f16x2', device=gpu)
    for kind in (L.SD_OUT_LOGITS_F32, L.SD_OUT_PROBS_U8):
        bad.forward(x, kind)
        assert bad.overflowed(), (arch, shape, kind)
        assert not bad.overflowed()                      # cleared by the read


def test_predictor_float16_flag_selects_the_plan_and_overflow_falls_back(gpu):
    """`float16=False` (what SyConn passes, prediction.py:777-779) = the reference-precision plan 'f16x2', `float16=True`
    (elektronn3: model.half()) = 'f16'; an overflow of the default plans repeats the prediction in the plan with fp32's exponent
    range ('f16x2' -> 'f32', 'f16' -> 'bf16', bit-identical to a Predictor built with that type); an explicitly requested
    storage type raises ActivationOverflowError unless `overflow_fallback=True` (dense_predictor's setting)."""
    from syconn_amd import _lib as L
    from syconn_amd.handler.prediction import Predictor
    good = build_unet('myelin', seed=3, final_scale=4.0)
    kw = dict(tile_shape=(4, 16, 32), overlap_shape=(2, 4, 4), out_shape=(2, 8, 32, 64), strict_shapes=True, apply_softmax=True)
    assert Predictor(good, **kw).act_dtype == 'f16x2'
    assert Predictor(good, float16=True, **kw).act_dtype == 'f16'
    assert Predictor(good, float16=True, act_dtype='bf16', **kw).act_dtype == 'bf16'
    model = _blow_up(build_unet('myelin', seed=3, final_scale=4.0))
    raw = _input((8, 32, 64), 2)
    want32 = Predictor(model, act_dtype='f32', **kw).predict_proba_u8_device(raw.to(gpu))
    p = Predictor(model, **kw)
    got = p.predict_proba_u8_device(raw.to(gpu))
    assert p.act_dtype == 'f32' and torch.equal(got, want32)
    want16 = Predictor(model, act_dtype='bf16', **kw).predict_proba_u8_device(raw.to(gpu))
    p = Predictor(model, float16=True, **kw)
    assert torch.equal(p.predict_proba_u8_device(raw.to(gpu)), want16) and p.act_dtype == 'bf16'
    with pytest.raises(L.ActivationOverflowError):
        Predictor(model, act_dtype='f16x2', **kw).predict((raw.float() / 255.)[None, None].numpy())
    p = Predictor(model, act_dtype='f16x2', overflow_fallback=True, **kw)
    assert torch.equal(p.predict_proba_u8_device(raw.to(gpu)), want32) and p.act_dtype == 'f32'


def test_non_sticky_overflow_fallback_repeats_one_prediction_only(gpu):
    """`sticky_fallback=False` (dense_predictor's default, `dense_prediction.sticky_overflow_fallback: false`): the prediction that
    overflowed is repeated in the plan with fp32's exponent range -- bit-identical to that plan -- and the NEXT prediction runs in the
    configured plan again; `n_fallbacks` counts the repeats."""
    from syconn_amd.handler.prediction import Predictor
    kw = dict(tile_shape=(4, 16, 32), overlap_shape=(2, 4, 4), out_shape=(2, 8, 32, 64), strict_shapes=True, apply_softmax=True)
    model = _blow_up(build_unet('myelin', seed=3, final_scale=4.0))
    raw = _input((8, 32, 64), 2).to(gpu)
    want32 = Predictor(model, act_dtype='f32', **kw).predict_proba_u8_device(raw)
    p = Predictor(model, sticky_fallback=False, **kw)
    assert p.act_dtype == 'f16x2' and p.n_fallbacks == 0
    for n in (1, 2):
        got = p.predict_proba_u8_device(raw)
        assert torch.equal(got, want32), 'the repeated prediction is not the f32 plan\'s'
        assert p.act_dtype == 'f16x2' and p.n_fallbacks == n          # back in the configured plan, one more repeat counted


@pytest.mark.parametrize('arch', sorted(ARCHS))
def test_split_plan_against_committed_golden_logits(gpu, arch):
    """the split plan against the COMMITTED fp32 logits of all 8 architectures (tests/golden/g4_unet_logits.npz: even and odd
    tiles, generated by the oracle in the build container -- a fixture that travels)"""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g4_unet_logits.npz'))
    dm = DenseModel(build_unet(arch, seed=100), act_dtype='f16x2', device=gpu)
    for tag in ('even', 'odd'):
        x = torch.from_numpy(g[f'{arch}_{tag}_in'])
        ref = torch.from_numpy(g[f'{arch}_{tag}_logits'])
        out = dm.forward(x.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        assert out.shape == ref.shape and err <= TOL, (arch, tag, err)
        assert torch.equal(out.argmax(0), ref.argmax(0)) or float((out.argmax(0) != ref.argmax(0)).float().mean()) < 1e-4


def test_config1_cnn3_through_the_default_predictor(gpu):
    """BASELINE configs[0] (64^3, 3-layer CNN, Predictor tiling with overlap) through the DEFAULT Predictor (float16=False ->
    'f16x2'): the committed oracle output is reproduced up to single-level flips at truncation boundaries."""
    from syconn_amd.handler.prediction import Predictor, dense_predicton_helper
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g6_config1.npz'))
    raw = np.random.default_rng(0).integers(0, 256, (64, 64, 64), dtype=np.uint8)      # BASELINE configs[0]
    p = Predictor(build_cnn3(seed=0), tile_shape=(32, 32, 32), overlap_shape=(8, 8, 8), out_shape=(2, 64, 64, 64),
                  strict_shapes=True, apply_softmax=True)
    assert p.act_dtype == 'f16x2'
    out = dense_predicton_helper(raw.astype(np.float32) / 255., p, is_zyx=True, return_zyx=True)
    d = np.abs(out.astype(np.int16) - g['out_u8'].astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 1e-4, (int(d.max()), float((d > 0).mean()))
    assert np.array_equal(out.argmax(0)[d.max(0) == 0], g['out_u8'].argmax(0)[d.max(0) == 0])


def test_split_fusions_agree_with_the_layer_wise_split_plan(gpu, monkeypatch):
    """The split plan's epilogue fusions -- MaxPool3d in the producing convolution (exact: the maximum of the same fp32 values)
    and the final 1x1x1 as three fp16 MFMA products of scaled hi / lo weight fragments instead of an fp32 FMA chain -- against
    the same plan with every op as its own launch (SD_NO_FUSE): pooled buffers bit-identical, logits within 2e-6 of the logit
    range, uint8 probabilities within one level on a vanishing fraction of voxels."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    for arch, shape in (('myelin', (6, 44, 70)), ('semseg_axon', (4, 33, 47)), ('mivcsj', (5, 36, 40))):
        model = build_unet(arch, seed=4, final_scale=5.0)
        raw = _input(shape, 6).to(gpu)
        fused = DenseModel(model, act_dtype='f16x2', device=gpu)
        a = fused.forward(raw, L.SD_OUT_LOGITS_F32).cpu()
        au8 = fused.forward(raw, L.SD_OUT_PROBS_U8).cpu()
        n_fused = fused.last_launch_count()
        monkeypatch.setenv('SD_NO_FUSE', '1')
        plain = DenseModel(model, act_dtype='f16x2', device=gpu)
        monkeypatch.delenv('SD_NO_FUSE')
        b = plain.forward(raw, L.SD_OUT_LOGITS_F32).cpu()
        bu8 = plain.forward(raw, L.SD_OUT_PROBS_U8).cpu()
        # (GroupNorm nets: a GroupNorm sits between every convolution and its pooling / the final layer -- nothing to fuse there)
        assert n_fused < plain.last_launch_count() or arch == 'mivcsj', (arch, n_fused, plain.last_launch_count())
        err = float((a - b).abs().max()) / float(b.abs().max())
        d = (au8.int() - bu8.int()).abs()
        # (GroupNorm nets: the fused statistics sum in another order than the separate pass -- both within 1e-5 of the oracle)
        assert err <= (5e-6 if arch == 'mivcsj' else 2e-6), (arch, err)
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3, (arch, int(d.max()))


@pytest.mark.parametrize('arch,act,ntiles,shape,switches', [
    ('semseg_axon', 'f16x2', 1, (128, 128, 128), ('SD_SPLIT_ROWS96_L2', 'SD_PLANAR_NT3_BIG')),
    ('mivcsj', 'f16x2', 4, (128, 128, 128), ('SD_SPLIT_ROWS96_L2', 'SD_PLANAR_NT3_BIG')),
    ('semseg_spine', 'f16x2', 4, (8, 96, 128), ('SD_SPLIT_ROWS32_NO_WL', 'SD_SPLIT_UPCONV128_MFMA')),
    ('semseg_spine', 'bf16', 4, (8, 96, 128), ('SD_UPCONV32_NO_WL', 'SD_UPCONV128_MFMA')),
    ('myelin', 'bf16', 1, (178, 243, 331), ('SD_UPCONV32_NO_WL', 'SD_UPCONV128_MFMA', 'SD_PLANAR4_H_RULE', 'SD_MT4_D_RULE')),      # the reference's tile
    ('semseg_axon', 'bf16', 6, (128, 128, 128), ('SD_UPCONV192_MFMA', 'SD_WRES_CAP_KB')),      # (the grouped forms need >= 6 such tiles per launch set)
    ('myelin', 'f16x2', 1, (178, 243, 331), ('SD_SPLIT_UPCONV128_MFMA', 'SD_SPLIT_ROWS32_NO_WL', 'SD_PLANAR4_H_RULE', 'SD_MT4_D_RULE')),
])
def test_round5_up_convolution_and_workgroup_forms_are_bit_identical(gpu, monkeypatch, arch, act, ntiles, shape, switches):
    """The forms picked in round 5 -- up-convolutions with LDS-resident weights (64 -> 32 channels; 256 -> 128, 384 -> 192 and the split
    plan's 192 -> 96 / 256 -> 128 in channel groups, k_upconv_rows<G>), planar 96-column layers as 4-wave workgroups in the split plan
    (sizes at which the 8-wave form used to be picked), the relaxed height / depth rules of the four-tile forms -- against the forms
    they replaced (launch-time switches): same logits, bit for bit."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    from syconn_amd.cnn import random_state_dict
    monkeypatch.setenv('SD_NO_DEC0', '1')            # (plain plan: the level-0 up-convolution as its own launch)
    dm = DenseModel(random_state_dict(arch, seed=2, final_scale=4.0), act_dtype=act, device=gpu)
    raw = torch.stack([_input(shape, 11 + k) for k in range(ntiles)]).to(gpu)
    new = dm.forward_batch(raw, L.SD_OUT_LOGITS_F32).clone()
    for sw in switches:
        monkeypatch.setenv(sw, '96' if sw == 'SD_WRES_CAP_KB' else '1')
    old = dm.forward_batch(raw, L.SD_OUT_LOGITS_F32)
    assert torch.equal(new, old)
    assert float(new.abs().max()) > 0 and not dm.overflowed()
