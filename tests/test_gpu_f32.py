"""Reference-precision mode (act_dtype='f32', syconn_amd/csrc/sd_f32.hip) against the fp32 torch-CPU oracle.

The reference runs the U-Net in fp32 (/root/reference/syconn/handler/prediction.py:777-779: `float16` is never set), so in
this mode HIP path and oracle differ by fp32 summation order only.  Stated tolerance: max |logit error| <= 2e-5 of the
largest |logit| (measured ~1e-6); uint8 probabilities within one level on a vanishing fraction of voxels; labels equal
except where the oracle's own value sits within that tolerance of a decision boundary (a-priori split of
oracle/label_margin.py)."""
import numpy as np
import pytest
import torch

from oracle.label_margin import TOL_LOGIT_REL, label_split
from oracle.unet_ref import ARCHS, build_cnn3, build_unet

pytestmark = pytest.mark.gpu

TOL_F32 = TOL_LOGIT_REL['f32']


def _input(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)


@pytest.mark.parametrize('arch,shape', [('myelin', (8, 32, 48)), ('myelin', (13, 27, 29)), ('er', (8, 40, 72)),
                                        ('syntype', (16, 33, 47)), ('syntype_enh', (8, 24, 40)), ('mivcsj', (9, 35, 52)),
                                        ('semseg_spine', (6, 70, 130)), ('semseg_axon', (5, 17, 50)), ('golgi', (8, 16, 16))])
def test_f32_logits_match_fp32_oracle(gpu, arch, shape):
    """All 8 architectures of the path, even and odd tiles (ceil-mode pooling + autocrop), every width class of the fp32
    convolution kernel (64 / 32 / 16 column blocks)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_unet(arch, seed=5, final_scale=4.0)
    raw = _input(shape, 3)
    with torch.no_grad():
        ref = model((raw.float() / 255.)[None, None])[0]
    dm = DenseModel(model, act_dtype='f32', device=gpu)
    out = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    print(f'{arch} {shape}: max logit err / max logit = {err:.2e}')
    assert err <= TOL_F32, (arch, shape, err)
    # float32 input == uint8 input (normalisation float32(v)/255, prediction.py:808), bit for bit
    out_f = dm.forward((raw.float() / 255.).to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    assert torch.equal(out, out_f)
    # softmax / uint8 kinds
    pr = dm.forward(raw.to(gpu), L.SD_OUT_PROBS_F32).cpu()
    assert float((pr - ref.softmax(0)).abs().max()) <= TOL_F32 * float(ref.abs().max()) + 1e-6     # |dp| <= |dlogit|
    u8 = dm.forward(raw.to(gpu), L.SD_OUT_PROBS_U8).cpu()
    ref_u8 = torch.from_numpy((ref.softmax(0).numpy() * 255).astype(np.uint8))          # prediction.py:864-865
    d = (u8.int() - ref_u8.int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3, (int(d.max()), float((d > 0).float().mean()))


def test_f32_layerwise_buffers_match_oracle(gpu, monkeypatch):
    """Every activation buffer of the fp32 plan (planar, real channel count = torch's layout) against the oracle's tensor of
    the same layer, GroupNorm network included (statistics over the autocropped region of up-convolution outputs)."""
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
        dm = DenseModel(model, act_dtype='f32', device=gpu)
        dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32)
        for i, t in enumerate(col):
            got = dm.read_buffer(i + 1).cpu()
            t = t[0]
            g = got[:, :t.shape[1], :t.shape[2], :t.shape[3]]
            err = float((g - t).abs().max()) / max(float(t.abs().max()), 1e-6)
            assert err <= 2e-5, (arch, i + 1, tuple(t.shape), err)


def test_f32_batch_and_labels(gpu):
    """Batched launch == single forwards (bit for bit); fused label output == the label rule on the uint8 probabilities; the
    3-layer CNN of BASELINE configs[0] reproduces the oracle's uint8 output up to single-level flips at truncation boundaries."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel, postproc_labels
    model = build_unet('semseg_spine', seed=1, final_scale=6.0)
    dm = DenseModel(model, act_dtype='f32', device=gpu)
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
    u8 = DenseModel(cnn, act_dtype='f32', device=gpu).forward(raw.to(gpu), L.SD_OUT_PROBS_U8).cpu()
    d = (u8.int() - torch.from_numpy((ref.numpy() * 255).astype(np.uint8)).int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3


def test_f32_headline_tile_argmax_is_exact_where_it_can_be(gpu):
    """BASELINE configs[1] tile (semseg_spine 128^3, bench.py's weights and tile) in the reference-precision mode: argmax
    agreement with the fp32 oracle >= 0.99999 and no mismatch outside the a-priori margin."""
    from bench import BENCH_FINAL_SCALE, synthetic_em_tiles
    from oracle.unet_ref import UNet
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    sd = random_state_dict('semseg_spine', seed=0, final_scale=BENCH_FINAL_SCALE)
    ref_net = UNet(in_channels=1, **ARCHS['semseg_spine']).eval()
    ref_net.load_state_dict(sd)
    raw = torch.from_numpy(synthetic_em_tiles(1, 128, seed=1))
    with torch.no_grad():
        ref_logits = ref_net((raw[0].float() / 255.)[None, None])[0]
    dm = DenseModel(sd, act_dtype='f32', device=gpu)
    ids = list(range(1, dm.out_channels))
    x = raw.to(gpu)
    lg = dm.forward_batch(x, L.SD_OUT_LOGITS_F32)[0].cpu()
    pr = dm.forward_batch(x, L.SD_OUT_PROBS_F32)[0].cpu()
    lab = dm.forward_labels_batch(x, ids, [127.5] * len(ids))[0].cpu()
    r = label_split(ref_logits, lg, pr, lab, ids, [None] * dm.out_channels, TOL_F32)
    print('\n[f32] 128^3 semseg_spine vs fp32 oracle: ' + ', '.join(f'{k}={v:.4g}' for k, v in r.items()))
    assert r['logit_err_max_rel'] <= TOL_F32, r
    assert r['argmax_agreement'] >= 0.99999 and r['label_agreement'] >= 0.9999, r
    assert r['argmax_mismatch_safe'] == 0 and r['label_mismatch_safe'] == 0, r


def _blow_up(model, factor=2e5):
    """Scale one mid-network convolution so that its outputs leave fp16's range (> 65504) but stay far inside fp32's."""
    with torch.no_grad():
        model.down_convs[1].conv2.weight.mul_(factor)
    return model


@pytest.mark.parametrize('arch,shape', [('myelin', (3, 6, 50)), ('myelin', (4, 40, 64)), ('mivcsj', (4, 35, 38)), ('er', (4, 24, 40))])
def test_fp16_range_guard(gpu, arch, shape):
    """fp16 storage overflows above 65504; the final-layer kernels flag it (sd_model_overflow) in every plan that can serve
    a shape: conv epilogue with fused final layer (H < 8), the streaming level-0 decoder, the MFMA final layer of the
    GroupNorm nets, the 48-filter family -- for logits and for probabilities.  bf16 / f32 never flag; a healthy fp16 model
    does not flag; the flag is cleared by reading it."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    x = _input(shape, 1).to(gpu)
    ok = DenseModel(build_unet(arch, seed=3, final_scale=4.0), act_dtype='f16', device=gpu)
    ok.forward(x, L.SD_OUT_PROBS_U8)
    assert not ok.overflowed()
    bad_model = _blow_up(build_unet(arch, seed=3, final_scale=4.0))
    bad = DenseModel(bad_model, act_dtype='f16', device=gpu)
    for kind in (L.SD_OUT_LOGITS_F32, L.SD_OUT_PROBS_F32, L.SD_OUT_PROBS_U8):
        bad.forward(x, kind)
        assert bad.overflowed(), (arch, shape, kind)
        assert not bad.overflowed()                      # cleared by the read
    bad.forward_labels_batch(x[None], [1], [127.5])
    with pytest.raises(L.ActivationOverflowError):
        bad.check_overflow()
    for act in ('bf16', 'f32'):
        other = DenseModel(bad_model, act_dtype=act, device=gpu)
        out = other.forward(x, L.SD_OUT_LOGITS_F32)
        assert not other.overflowed() and bool(torch.isfinite(out).all())


def test_predictor_falls_back_to_bf16_on_fp16_overflow(gpu):
    """Predictor(float16=True) (default storage type 'f16') repeats an overflowed prediction in bf16 (== a bf16 Predictor, bit
    for bit) and stays there; an explicit act_dtype='f16' raises ActivationOverflowError (a RuntimeError).  (float16=False and its
    'f16x2' -> 'f32' fallback: tests/test_gpu_split.py.)"""
    from syconn_amd import _lib as L
    from syconn_amd.handler.prediction import Predictor
    model = _blow_up(build_unet('myelin', seed=3, final_scale=4.0))
    raw = _input((8, 32, 64), 2)
    kw = dict(tile_shape=(4, 16, 32), overlap_shape=(2, 4, 4), out_shape=(2, 8, 32, 64), strict_shapes=True, apply_softmax=True)
    want = Predictor(model, act_dtype='bf16', **kw).predict_proba_u8_device(raw.to(gpu))
    p = Predictor(model, float16=True, **kw)
    assert p.act_dtype == 'f16'
    got = p.predict_proba_u8_device(raw.to(gpu))
    assert p.act_dtype == 'bf16' and torch.equal(got, want)
    assert torch.equal(p.predict_proba_u8_device(raw.to(gpu)), want)
    with pytest.raises(L.ActivationOverflowError):
        Predictor(model, act_dtype='f16', **kw).predict((raw.float() / 255.)[None, None].numpy())


@pytest.mark.parametrize('arch', sorted(ARCHS))
def test_f32_mode_against_committed_golden_logits(gpu, arch):
    """the reference-precision mode against the COMMITTED fp32 logits of all 8 architectures (tests/golden/g4_unet_logits.npz:
    even and odd tiles, generated by the oracle in the build container -- a fixture that travels, independent of the CPU and
    oneDNN kernel of the box the test runs on); bf16 / fp16 are checked against the same fixture with their tolerances in
    tests/test_gpu_unet.py"""
    import os
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g4_unet_logits.npz'))
    dm = DenseModel(build_unet(arch, seed=100), act_dtype='f32', device=gpu)
    for tag in ('even', 'odd'):
        x = torch.from_numpy(g[f'{arch}_{tag}_in'])
        ref = torch.from_numpy(g[f'{arch}_{tag}_logits'])
        out = dm.forward(x.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        assert out.shape == ref.shape and err <= TOL_F32, (arch, tag, err)
        assert torch.equal(out.argmax(0), ref.argmax(0)) or float((out.argmax(0) != ref.argmax(0)).float().mean()) < 1e-4
