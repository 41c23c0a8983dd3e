"""GPU parity of the HIP U-Net forward (through the C ABI) against the torch-CPU oracle.

Two references per case:
* the fp32 oracle (``oracle.unet_ref.UNet``) -- tolerance `TOL_FP32` states what bf16/fp16 storage costs;
* the reduced-precision emulation of the same graph (``unet_forward_emulated``: weights + stored activations
  rounded to the activation dtype, fp32 accumulate) -- differs from the HIP path only by summation order, so the
  tolerance is tight and catches indexing bugs a loose bf16 tolerance would hide.
"""
import os

import numpy as np
import pytest
import torch

from oracle.unet_ref import build_cnn3, build_unet, unet_forward_emulated

pytestmark = pytest.mark.gpu

# stated tolerances: max |diff| relative to max|logit| of the case, and RMS diff relative to RMS logit.
# One bf16 ulp is 2^-8 = 3.9e-3 relative: a different summation order flips roundings of stored activations, so even
# the same-precision emulation differs by a few ulp at the logits after ~20 layers; the RMS bound is the sharp one.
TOL_EMU = {'bf16': 2e-2, 'f16': 3e-3}        # HIP vs same-precision emulation (max; measured over all cases: 1.45e-2 / 2.0e-3)
TOL_EMU_RMS = {'bf16': 1.2e-2, 'f16': 2.5e-3}   # (measured: 9.0e-3 / 1.7e-3)
TOL_FP32 = {'bf16': 5e-2, 'f16': 8e-3}       # HIP vs fp32 oracle (max)
TOL_FP32_RMS = {'bf16': 2e-2, 'f16': 4e-3}


def _input(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)


def _run_case(gpu, model, shape, act, seed=0, check_layers=True):
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    raw = _input(shape, seed)
    x = (raw.to(torch.float32) / 255.)[None, None]
    with torch.no_grad():
        ref = model(x)[0]
        col = []
        dt = torch.bfloat16 if act == 'bf16' else torch.float16
        emu = unet_forward_emulated(model, x, dtype=dt, collect=col)[0]
    dm = DenseModel(model, act_dtype=act, device=gpu)
    out = dm.forward(raw.to(gpu), out_kind=L.SD_OUT_LOGITS_F32).cpu()
    scale = float(ref.abs().max())
    rms = float(ref.pow(2).mean().sqrt())
    e_emu = float((out - emu).abs().max()) / scale
    e_ref = float((out - ref).abs().max()) / scale
    r_emu = float((out - emu).pow(2).mean().sqrt()) / rms
    r_ref = float((out - ref).pow(2).mean().sqrt()) / rms
    msg = (f'act={act} shape={shape}: vs emulation max {e_emu:.2e} rms {r_emu:.2e}; '
           f'vs fp32 oracle max {e_ref:.2e} rms {r_ref:.2e}')
    if check_layers and (e_emu > TOL_EMU[act] or r_emu > TOL_EMU_RMS[act]):
        for i, t in enumerate(col):
            got = dm.read_buffer(i + 1).cpu()
            t = t[0]
            g = got[:, :t.shape[1], :t.shape[2], :t.shape[3]]
            err = float((g - t).abs().max()) / max(float(t.abs().max()), 1e-6)
            msg += f'\n  buffer {i + 1} {tuple(t.shape)}: rel err {err:.2e}'
    print(msg)
    assert e_emu <= TOL_EMU[act] and r_emu <= TOL_EMU_RMS[act], msg
    assert e_ref <= TOL_FP32[act] and r_ref <= TOL_FP32_RMS[act], msg
    # same input as float32 (Predictor.predict's path) against the uint8 fast path.  A planar first convolution takes uint8 input
    # through two bf16 MFMAs on the exact uint8 values (weights / 255 split three ways: fp32-level arithmetic, but not the bit pattern
    # of the float32(v) / 255 chain): a rounding of a stored activation flips here and there, so the two paths agree like two summation
    # orders do (the emulation tolerance).  With the chain forced for uint8 too (SD_NO_FIRST_U8) they are equal bit for bit.
    xf = (raw.to(torch.float32) / 255.).to(gpu)
    out_f = dm.forward(xf, out_kind=L.SD_OUT_LOGITS_F32).cpu()
    assert float((out_f - out).abs().max()) / scale <= TOL_EMU[act], 'uint8 and float32 input paths differ'
    os.environ['SD_NO_FIRST_U8'] = '1'
    try:
        out_c = dm.forward(raw.to(gpu), out_kind=L.SD_OUT_LOGITS_F32).cpu()
    finally:
        del os.environ['SD_NO_FIRST_U8']
    assert torch.equal(out_f, out_c), 'uint8 (exact-f32 chain) and float32 input paths differ'
    return out, ref, emu


@pytest.mark.parametrize('act', ['bf16', 'f16'])
def test_small_unet_even(gpu, act):
    model = build_unet('myelin', seed=1, n_blocks=3, start_filts=16)
    _run_case(gpu, model, (8, 32, 32), act)


@pytest.mark.parametrize('shape', [(13, 27, 29), (5, 17, 50), (4, 16, 16)])
def test_small_unet_odd_shapes(gpu, shape):
    """ceil-mode pooling + autocrop and partial workgroup tiles."""
    model = build_unet('myelin', seed=2, n_blocks=3, start_filts=16)
    _run_case(gpu, model, shape, 'bf16')


@pytest.mark.parametrize('arch', ['myelin', 'er', 'syntype', 'syntype_enh', 'semseg_spine', 'semseg_axon'])
def test_syconn_architectures(gpu, arch):
    """Every BatchNorm architecture SyConn instantiates (full size) on a small odd tile."""
    model = build_unet(arch, seed=3)
    _run_case(gpu, model, (10, 37, 43), 'bf16', check_layers=True)


def test_mivcsj_groupnorm(gpu):
    """GroupNorm(8) U-Net with 5 blocks (cnn_cellorganelles.py:69-77): runtime statistics per tile."""
    model = build_unet('mivcsj', seed=4)
    _run_case(gpu, model, (12, 35, 41), 'bf16')


@pytest.mark.parametrize('defer', [True, False])
def test_groupnorm_net_with_64_channel_groups_large(gpu, monkeypatch, defer):
    """A GroupNorm variant of the 32-filter family on tiles large enough for the 4-tile 3x3x3 form: that form with FUSED
    GroupNorm statistics (symmetric epilogue: the statistics reduction needs all waves) and the deferred-apply plan, against
    the fp32 oracle (no architecture of the reference combines the two; the dispatch allows it)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    if not defer:
        monkeypatch.setenv('SD_NO_GN_DEFER', '1')       # separate apply passes: plain conv kernels with fused statistics
    model = build_unet('myelin', seed=6, normalization='group8')
    raw = _input((64, 128, 144), 23)
    with torch.no_grad():
        ref = model((raw.float() / 255.)[None, None])[0]
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    out = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    assert torch.isfinite(out).all()
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    rms = float((out - ref).pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
    print(f'myelin/group8 (64, 128, 144): rel err max {err:.2e} rms {rms:.2e}')
    assert err < TOL_FP32['bf16'] and rms < TOL_FP32_RMS['bf16']


def test_mivcsj_fp16(gpu):
    model = build_unet('mivcsj', seed=5)
    _run_case(gpu, model, (8, 32, 48), 'f16')


def test_cnn3_config1(gpu):
    """BASELINE.json config 1: 3-layer CNN (8 channels -> padded to one 16-channel chunk)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_cnn3(0)
    raw = _input((20, 33, 30), 7)
    with torch.no_grad():
        ref = model((raw.float() / 255.)[None, None])[0]
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    out = dm.forward(raw.to(gpu), out_kind=L.SD_OUT_LOGITS_F32).cpu()
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    print('cnn3 rel err', err)
    assert err < 2e-2


def test_outputs_softmax_u8_and_argmax(gpu):
    """softmax / floor(255 p) epilogues and label agreement where the fp32 margin exceeds the stated tolerance."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_unet('semseg_spine', seed=6, final_scale=16.0)
    shape = (12, 40, 48)
    raw = _input(shape, 11)
    with torch.no_grad():
        logits = model((raw.float() / 255.)[None, None])[0]
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    lg = dm.forward(raw.to(gpu), out_kind=L.SD_OUT_LOGITS_F32).cpu()
    pr = dm.forward(raw.to(gpu), out_kind=L.SD_OUT_PROBS_F32).cpu()
    u8 = dm.forward(raw.to(gpu), out_kind=L.SD_OUT_PROBS_U8).cpu()
    assert torch.allclose(pr, lg.softmax(0), atol=2e-6)
    assert torch.allclose(pr.sum(0), torch.ones(shape), atol=1e-5)
    # uint8 = truncation of 255*p computed from the SAME device probabilities (prediction.py:864-865)
    assert torch.equal(u8, torch.from_numpy((pr.numpy() * 255).astype(np.uint8)))
    tol = TOL_FP32['bf16'] * float(logits.abs().max())
    top2 = logits.topk(2, dim=0).values
    safe = (top2[0] - top2[1]) > 2 * tol
    agree = lg.argmax(0) == logits.argmax(0)
    print(f'argmax: {int(safe.sum())}/{safe.numel()} voxels have margin > 2*tol; '
          f'mismatches inside margin-safe set: {int((~agree & safe).sum())}, outside: {int((~agree & ~safe).sum())}')
    assert bool((agree | ~safe).all()), 'argmax label differs on a voxel whose fp32 margin exceeds the tolerance'
    assert float(safe.float().mean()) > 0.3, 'test input has too few margin-safe voxels to be meaningful'


def test_epilogue_fusions_match_unfused(gpu, monkeypatch):
    """MaxPool and conv_final+softmax+uint8 fused into the producing convolution's epilogue must reproduce the
    separate-launch path: pooled activations bit-exact (max of the same rounded values), class probabilities up to
    fp32 summation order."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    for arch, shape in (('semseg_spine', (9, 35, 37)), ('syntype', (6, 20, 36)), ('semseg_axon', (4, 18, 34))):
        model = build_unet(arch, seed=9, final_scale=4.0)
        raw = _input(shape, 5).to(gpu)
        monkeypatch.setenv('SD_NO_FUSE', '1')
        plain = DenseModel(model, act_dtype='bf16', device=gpu)
        monkeypatch.delenv('SD_NO_FUSE')
        monkeypatch.setenv('SD_KEEP_ALL', '1')
        fused = DenseModel(model, act_dtype='bf16', device=gpu)
        monkeypatch.delenv('SD_KEEP_ALL')
        a = plain.forward(raw, L.SD_OUT_PROBS_F32).cpu()
        b = fused.forward(raw, L.SD_OUT_PROBS_F32).cpu()
        assert float((a - b).abs().max()) < 1e-5, arch
        for buf in range(1, plain.info['n_buffers']):
            assert torch.equal(plain.read_buffer(buf), fused.read_buffer(buf)), (arch, buf)
        a8 = plain.forward(raw, L.SD_OUT_PROBS_U8).cpu().to(torch.int16)
        b8 = fused.forward(raw, L.SD_OUT_PROBS_U8).cpu().to(torch.int16)
        assert int((a8 - b8).abs().max()) <= 1 and float(((a8 - b8) != 0).float().mean()) < 1e-3
        al = plain.forward(raw, L.SD_OUT_LOGITS_F32).cpu()
        bl = fused.forward(raw, L.SD_OUT_LOGITS_F32).cpu()
        assert float((al - bl).abs().max()) < 1e-4 * float(al.abs().max())


def test_large_tile_many_workgroups(gpu):
    """A tile with far more output blocks than resident workgroups (persistent / multi-round scheduling) against the
    same network evaluated in pieces: the interior of the big result must equal independently computed sub-tiles
    wherever their receptive fields do not touch a cut (checks that EVERY block is computed exactly once)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_unet('myelin', seed=13, n_blocks=2, start_filts=32)      # receptive field radius (2, 8, 8)
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    raw = _input((40, 200, 240), 17).to(gpu)
    big = dm.forward(raw, L.SD_OUT_LOGITS_F32).cpu()
    assert bool(torch.isfinite(big).all())
    with torch.no_grad():
        ref = model((raw.cpu().float() / 255.)[None, None])[0]
    err = float((big - ref).abs().max()) / float(ref.abs().max())
    print('large tile rel err vs fp32 oracle', err)
    assert err < TOL_FP32['bf16']


@pytest.mark.parametrize('shape', [(1, 1, 1), (2, 3, 5), (3, 16, 2), (1, 40, 33)])
def test_tiny_and_degenerate_tiles(gpu, shape):
    """Tiles smaller than one workgroup block / one pooling window (every voxel is a border voxel)."""
    model = build_unet('myelin', seed=23, n_blocks=3, start_filts=16)
    _run_case(gpu, model, shape, 'bf16')


def test_uint8_normalisation_is_exact_for_all_values(gpu, monkeypatch):
    """In-kernel float32(v)/255 vs numpy's raw.astype(np.float32)/255. for all 256 input values: an identity-like
    first layer (single centre tap = 1) exposes the normalised value itself (bf16 rounding applies to both paths
    equally, so compare the uint8 path with the float32-input path bit for bit on every value)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_unet('myelin', seed=3, n_blocks=2, start_filts=16)
    raw = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16).repeat(2, 1, 1).contiguous()
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    fast = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()      # (the default uint8 form: bf16 MFMAs on the exact uint8 values)
    monkeypatch.setenv('SD_NO_FIRST_U8', '1')                       # the exact-f32 chain on float32(v) / 255 for uint8 input too
    a = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    assert float((fast - a).abs().max()) <= TOL_EMU['bf16'] * float(a.abs().max())
    b = dm.forward(torch.from_numpy(raw.numpy().astype(np.float32) / 255.).to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    assert torch.equal(a, b)
    # ... and on the first layer's own output (SD_KEEP_ALL: every buffer materialised in its own range, no fusion skips it)
    monkeypatch.setenv('SD_KEEP_ALL', '1')
    dk = DenseModel(model, act_dtype='bf16', device=gpu)
    dk.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32)
    first_u8 = dk.read_buffer(1).cpu().clone()
    dk.forward(torch.from_numpy(raw.numpy().astype(np.float32) / 255.).to(gpu), L.SD_OUT_LOGITS_F32)
    assert torch.equal(first_u8, dk.read_buffer(1).cpu()) and float(first_u8.abs().max()) > 0
    # ... and the default uint8 form gives the same first-layer tensor up to one rounding step of a stored value here and there
    monkeypatch.delenv('SD_NO_FIRST_U8')
    dk.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32)
    d = (dk.read_buffer(1).cpu() - first_u8).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(first_u8.abs().max()) and float((d > 0).float().mean()) < 1e-2


@pytest.mark.parametrize('arch,shape', [('er', (20, 150, 170)), ('syntype', (18, 140, 150)), ('mivcsj', (16, 100, 120))])
def test_full_architectures_many_workgroups(gpu, arch, shape):
    """Full-size production architectures on tiles with thousands of workgroups per layer (multi-round persistent
    scheduling, 48/28-channel padding, GroupNorm statistics over large tensors) against the fp32 oracle."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    model = build_unet(arch, seed=41)
    raw = _input(shape, 19)
    with torch.no_grad():
        ref = model((raw.float() / 255.)[None, None])[0]
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    out = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    out2 = dm.forward(raw.to(gpu), L.SD_OUT_LOGITS_F32).cpu()
    assert torch.equal(out, out2), 'forward is not deterministic'
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    rms = float((out - ref).pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
    print(f'{arch} {shape}: rel err max {err:.2e} rms {rms:.2e}')
    assert err < TOL_FP32['bf16'] and rms < TOL_FP32_RMS['bf16']


@pytest.mark.gpu
def test_ring_form_is_bit_identical(gpu, monkeypatch):
    """Planar layers of the 48-filter family with streamed weights run through 3-deep halo / weight rings (RING in k_conv_mfma,
    DMA groups issued under the SIMD partner's MFMAs) where the grid is large enough; same per-output summation order as the
    double-buffered form (SD_NO_RING) -> bit-identical logits, for even and ragged extents and for batches."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    shapes = ((2, 96, 160, 160), (3, 64, 150, 170))
    sd = random_state_dict('semseg_axon', seed=12, final_scale=6.0)
    monkeypatch.setenv('SD_NO_RING', '1')
    plain = DenseModel(sd, act_dtype='bf16', device=gpu)
    a = [plain.forward_batch(_input(sh, 5).to(gpu), L.SD_OUT_LOGITS_F32).clone() for sh in shapes]
    monkeypatch.delenv('SD_NO_RING')                    # (the switch is read at every launch)
    ring = DenseModel(sd, act_dtype='bf16', device=gpu)
    for sh, ref in zip(shapes, a):
        b = ring.forward_batch(_input(sh, 5).to(gpu), L.SD_OUT_LOGITS_F32, slot=1)
        assert torch.isfinite(b).all()
        assert torch.equal(ref, b), sh


@pytest.mark.gpu
@pytest.mark.parametrize('arch,shape', [('semseg_spine', (24, 40, 48)), ('mivcsj', (16, 32, 48)), ('syntype', (13, 27, 29))])
def test_forward_batch_equals_single_forwards(gpu, arch, shape):
    """sd_forward_batch (N tiles, one set of launches) is bit-identical to N sd_forward calls."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    net = build_unet(arch, seed=5, final_scale=4.0)
    dm = DenseModel(net, 'bf16', gpu)
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 256, (3, *shape), dtype=torch.uint8, generator=g).to(gpu)
    for kind in (L.SD_OUT_PROBS_U8, L.SD_OUT_LOGITS_F32):
        single = torch.stack([dm.forward(x[i], kind).clone() for i in range(3)])
        batched = dm.forward_batch(x, kind, slot=1)
        assert batched.shape == single.shape
        assert torch.equal(batched, single)


def test_full_size_tile_properties(gpu):
    """BASELINE configs[1] at full size (semseg_spine, 128^3, bf16), checked through size-independent properties:
    bit-identical repeat runs, batched == single launches, probabilities of a voxel sum to 255 up to the truncation
    (each of the C classes loses < 1), and translation consistency: the centre of a tile does not depend on what lies
    beyond the network's receptive field."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    net = build_unet('semseg_spine', seed=2, final_scale=6.0)
    dm = DenseModel(net, 'bf16', gpu)
    x = _input((2, 128, 128, 128), 5).to(gpu)
    a = dm.forward(x[0], L.SD_OUT_PROBS_U8).clone()
    b = dm.forward(x[0], L.SD_OUT_PROBS_U8).clone()
    assert torch.equal(a, b)
    both = dm.forward_batch(x, L.SD_OUT_PROBS_U8, slot=1)
    assert torch.equal(both[0], a) and torch.equal(both[1], dm.forward(x[1], L.SD_OUT_PROBS_U8))
    s = a.to(torch.int32).sum(0)
    assert int(s.max()) <= 255 and int(s.min()) > 255 - a.shape[0]
    # same data, but everything further than 48 voxels (> receptive field 44 of this net along y/x; planar levels do
    # not look along z at all levels, 3D ones reach 20) from the central 16^3 block replaced: the block must not change
    y = x[0].clone()
    y[:8] = 255 - y[:8]
    y[-8:] = 255 - y[-8:]
    y[:, :8] = 0
    y[:, :, -8:] = 17
    c = dm.forward(y, L.SD_OUT_PROBS_U8)
    assert torch.equal(c[:, 56:72, 56:72, 56:72], a[:, 56:72, 56:72, 56:72])


@pytest.mark.parametrize('arch,shape,ids,thr', [('semseg_spine', (24, 40, 48), (1, 2, 3, 4), (60.5, 50.0, 55.5, 45.0)),
                                                ('mivcsj', (16, 32, 48), (1, 2, 3), (90.5, 76.5, 60.0)),
                                                ('syntype', (13, 27, 29), (2, 1), (0.0, 254.5))])
def test_fused_label_rule_equals_probs_then_labels(gpu, arch, shape, ids, thr):
    """sd_forward_labels_batch == sd_postproc_labels(sd_forward_batch(PROBS_U8)) bit for bit (fused and unfused final
    layer, different id orders and thresholds)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel, postproc_labels
    net = build_unet(arch, seed=6, final_scale=5.0)
    dm = DenseModel(net, 'bf16', gpu)
    x = _input((2, *shape), 13).to(gpu)
    probs = dm.forward_batch(x, L.SD_OUT_PROBS_U8)
    want = torch.stack([postproc_labels(probs[i], list(ids), list(thr)) for i in range(2)])
    got = dm.forward_labels_batch(x, ids, thr)
    assert got.shape == want.shape and torch.equal(got, want)
    if arch != 'syntype':
        assert len(torch.unique(got)) >= 2       # the rule really selects between labels


def test_fused_first_conv_is_bit_identical(gpu, monkeypatch):
    """Level-0 pair first conv (1->32, 1x3x3) -> conv (1x3x3): the second conv computes its input halo from the uint8 /
    float tile itself (k_conv_mfma<FF>), the 32-channel tensor in between is never written.  Same arithmetic and rounding
    points -> every later activation buffer and the output must be bit-identical to the separate launches; odd extents
    exercise the zero padding of both convolutions, two tiles the batched path."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    # (semseg_axon / er: the 48-filter family -- a two-tile first convolution into three resident halo slots, round 4)
    for arch, shape in (('semseg_spine', (12, 150, 170)), ('myelin', (9, 131, 77)), ('syntype', (16, 128, 144)),
                        ('semseg_axon', (10, 140, 150)), ('er', (12, 150, 170))):
        model = build_unet(arch, seed=31, final_scale=4.0)
        raw = _input((2, *shape), 8).to(gpu)
        monkeypatch.setenv('SD_NO_FIRST_FUSE', '1')
        plain = DenseModel(model, act_dtype='bf16', device=gpu)
        monkeypatch.delenv('SD_NO_FIRST_FUSE')
        fused = DenseModel(model, act_dtype='bf16', device=gpu)
        for kind in (L.SD_OUT_LOGITS_F32, L.SD_OUT_PROBS_U8):
            a = plain.forward_batch(raw, kind)
            b = fused.forward_batch(raw, kind, slot=1)
            assert torch.equal(a, b), (arch, kind)
        af = plain.forward_batch(raw.float() / 255., L.SD_OUT_LOGITS_F32)          # float32 input path
        bf = fused.forward_batch(raw.float() / 255., L.SD_OUT_LOGITS_F32, slot=1)
        assert torch.equal(af, bf), arch


def test_large_single_tile_locality(gpu):
    """A 50-Mvoxel single tile (27 GiB of activations, > 2^31 bytes per chunk plane): deterministic, and an interior block
    equals the same region predicted as its own tile once the margin exceeds the receptive field (44 voxels in y/x, 20 in
    z for this network) -- pins 64-bit addressing of the channel-blocked planes and block-position independence."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    dm = DenseModel(build_unet('myelin', seed=1, final_scale=6.0), 'bf16', gpu)
    x = _input((192, 512, 512), 21).to(gpu)
    a = dm.forward(x, L.SD_OUT_PROBS_U8).clone()
    assert torch.equal(a, dm.forward(x, L.SD_OUT_PROBS_U8))
    sub = x[32:160, 128:384, 128:384].contiguous()
    c = dm.forward(sub, L.SD_OUT_PROBS_U8, slot=1)
    assert torch.equal(c[:, 48:80, 48:208, 48:208], a[:, 80:112, 176:336, 176:336])


@pytest.mark.parametrize('act', ['f16', 'bf16'])
def test_deferred_groupnorm_apply_is_bit_identical(gpu, monkeypatch, act):
    """GroupNorm networks: the apply (+ReLU) pass of every GroupNorm is deferred to the readers of the raw tensor
    (convolutions rewrite their LDS halo pieces, up-convolutions / the final layer their register fragments, a fused pooling
    writes the normalised pooled tensor only).  Same arithmetic and rounding point as the separate in-place pass -> logits,
    probabilities and labels must be bit-identical to the plan that runs every apply pass (SD_NO_GN_DEFER), for even and
    odd extents (autocrop regions), batches, and a tile with many workgroups per layer."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    sd = random_state_dict('mivcsj', seed=7, final_scale=5.0)
    monkeypatch.setenv('SD_NO_GN_DEFER', '1')
    plain = DenseModel(sd, act_dtype=act, device=gpu)
    monkeypatch.delenv('SD_NO_GN_DEFER')
    fused = DenseModel(sd, act_dtype=act, device=gpu)
    assert fused.workspace_bytes((32, 48, 48)) >= 0
    for shape in ((2, 16, 32, 48), (3, 13, 37, 43), (1, 32, 112, 144)):
        x = _input(shape, 3).to(gpu)
        for kind in (L.SD_OUT_LOGITS_F32, L.SD_OUT_PROBS_U8):
            a = plain.forward_batch(x, kind)
            b = fused.forward_batch(x, kind, slot=1)
            assert torch.equal(a, b), (act, shape, kind)
        la = plain.forward_labels_batch(x, (1, 2, 3), (127.5, 127.5, 127.5))
        lb = fused.forward_labels_batch(x, (1, 2, 3), (127.5, 127.5, 127.5), slot=1)
        assert torch.equal(la, lb)


def test_four_tile_form_is_bit_identical(gpu, monkeypatch):
    """3x3x3 layers with 64-channel groups run with 4 z-stacked voxel tiles per wave (8x8x16 blocks) where the grid is large
    enough; same per-output summation order as the 2-tile form (SD_MT2) -> bit-identical logits and labels, for z extents that
    are multiples of 8, ragged ones (partial top blocks, odd y / x with ceil-mode pooling and autocrop) and batches."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    for arch, shapes in (('semseg_spine', ((2, 64, 128, 144), (1, 100, 150, 170), (3, 32, 97, 131))),
                         ('syntype', ((2, 48, 112, 128),))):
        sd = random_state_dict(arch, seed=11, final_scale=6.0)
        monkeypatch.setenv('SD_MT2', '1')
        two = DenseModel(sd, act_dtype='bf16', device=gpu)
        a = [two.forward_batch(_input(sh, 4).to(gpu), L.SD_OUT_LOGITS_F32).clone() for sh in shapes]
        monkeypatch.delenv('SD_MT2')                    # (the switch is read at every launch)
        four = DenseModel(sd, act_dtype='bf16', device=gpu)
        for sh, ref in zip(shapes, a):
            b = four.forward_batch(_input(sh, 4).to(gpu), L.SD_OUT_LOGITS_F32, slot=1)
            assert torch.equal(ref, b), (arch, sh)


def test_planar_four_tile_form_is_bit_identical(gpu, monkeypatch):
    """Planar 64 / 128-filter layers (level 2 of the 32-filter nets) run with 4 y-stacked voxel tiles per wave in 4-wave workgroups
    of 1 x 32 x 16 voxels where the row count allows (round 4); same per-output summation order as the 2-tile form
    (SD_NO_PLANAR4) -> bit-identical logits, incl. the fused planar pooling of the encoder's second convolution, in the bf16
    plan and in the split plan."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    for act in ('bf16', 'f16x2'):
        for arch, shapes in (('semseg_spine', ((8, 64, 128, 128), (2, 32, 128, 144), (1, 16, 512, 200))), ('syntype', ((4, 48, 128, 128),))):
            sd = random_state_dict(arch, seed=13, final_scale=6.0)
            monkeypatch.setenv('SD_NO_PLANAR4', '1')
            two = DenseModel(sd, act_dtype=act, device=gpu)
            a = [two.forward_batch(_input(sh, 4).to(gpu), L.SD_OUT_LOGITS_F32).clone() for sh in shapes]
            monkeypatch.delenv('SD_NO_PLANAR4')                    # (the switch is read at every launch)
            four = DenseModel(sd, act_dtype=act, device=gpu)
            for sh, ref in zip(shapes, a):
                b = four.forward_batch(_input(sh, 4).to(gpu), L.SD_OUT_LOGITS_F32, slot=1)
                assert torch.equal(ref, b), (act, arch, sh)


@pytest.mark.parametrize('arch,act', [('semseg_spine', 'bf16'), ('semseg_spine', 'f16'), ('myelin', 'bf16'), ('syntype', 'f16')])
def test_fused_level0_decoder_matches_separate_layers(gpu, monkeypatch, arch, act):
    """sd_dec0.hip (up-convolution + merge conv + conv + final layer of the planar top level in ONE streaming launch) against the
    layer-by-layer plan (SD_NO_DEC0=1): same rounded weights, same rounding points and the same summation order (bias first, chunks
    in concat order, taps 0..8 -- the up-convolution kernels start from the bias as well since round 3), so EVERY output kind is
    bit-identical: which plan serves a shape (`dec0_shape_ok`, e.g. after the OOM tile-halving loop changed the tile) must not
    change results.  Shapes: odd extents (crop of the up-convolved tensor), 1 / 2 / 3 / 4 x-strips of 64 columns, rows that do not
    divide the 128-position steps, batches; and shapes the streaming kernel is not used for (H < 8, or a width that fills less
    than 70 % of its strips): sd_debug_last_launch_count tells which plan served a shape."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    net = build_unet(arch, seed=21, final_scale=6.0)
    dm = DenseModel(net, act, gpu)
    monkeypatch.setenv('SD_NO_DEC0', '1')
    layers = DenseModel(net, act, gpu)
    monkeypatch.delenv('SD_NO_DEC0')
    n_fused = 0
    for nb, shape in ((1, (3, 9, 47)), (2, (2, 33, 121)), (1, (5, 64, 64)), (3, (4, 50, 183)), (1, (2, 131, 200)), (1, (3, 6, 50)),
                      (1, (3, 20, 70))):
        x = _input((nb, *shape), 31 + shape[2]).to(gpu)
        a, b = dm.forward_batch(x, L.SD_OUT_LOGITS_F32).cpu(), layers.forward_batch(x, L.SD_OUT_LOGITS_F32).cpu()
        served = shape[1] >= 8 and shape[2] * 10 >= -(-shape[2] // 64) * 64 * 7
        assert dm.last_launch_count() == layers.last_launch_count() - (2 if served else 0), shape
        n_fused += int(served)
        assert torch.equal(a, b), (arch, act, shape)
        for kind in (L.SD_OUT_PROBS_U8, L.SD_OUT_PROBS_F32):
            assert torch.equal(dm.forward_batch(x, kind).cpu(), layers.forward_batch(x, kind).cpu()), (arch, act, shape, kind)
    assert n_fused == 5


def test_fused_level0_decoder_label_rules(gpu):
    """Both label-rule forms of the streaming decoder kernel -- the per-class table (distinct ids) and the generic list (an
    id listed twice: the later entry overrides) -- equal sd_postproc_labels applied to the kernel's own uint8 probabilities."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel, postproc_labels
    net = build_unet('semseg_spine', seed=4, final_scale=6.0)
    dm = DenseModel(net, 'bf16', gpu)
    x = _input((2, 6, 70, 90), 77).to(gpu)
    probs = dm.forward_batch(x, L.SD_OUT_PROBS_U8)
    for ids, thr in (((4, 2, 1, 3), (40.0, 60.5, 30.0, 80.0)), ((1, 2, 1), (20.0, 50.0, 120.5)), ((3,), (-1.0,)), ((2, 2), (254.5, 10.0))):
        want = torch.stack([postproc_labels(probs[i], list(ids), list(thr)) for i in range(2)])
        got = dm.forward_labels_batch(x, ids, thr)
        assert torch.equal(got, want), (ids, thr)
    assert len(torch.unique(dm.forward_labels_batch(x, (4, 2, 1, 3), (40.0, 60.5, 30.0, 80.0)))) >= 3


@pytest.mark.parametrize('act', ['f16', 'bf16'])
def test_groupnorm_raw_pooling_in_conv_epilogue_is_bit_identical(gpu, monkeypatch, act):
    """GroupNorm + ReLU + MaxPool behind a convolution: the conv's epilogue pools its RAW output -- per channel the window's
    maximum, or its minimum where gamma < 0 (relu(a x + b) is monotone in x) -- and the next level applies scale / shift on
    load, so the apply + pool pass over the tensor disappears.  Half of all GroupNorm weights negative here: logits,
    probabilities and labels must equal the plan with separate apply passes (SD_NO_GN_DEFER) and the plan that only lacks
    this fusion (SD_NO_GN_POOL_RAW) bit for bit; and stay within the fp32 oracle's tolerance."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    sd = random_state_dict('mivcsj', seed=9, final_scale=5.0)
    g = torch.Generator().manual_seed(1)
    for k in sd:
        if '.norm' in k and k.endswith('.weight'):
            sd[k] = sd[k] * (torch.randint(0, 2, sd[k].shape, generator=g) * 2 - 1).to(sd[k].dtype)
    monkeypatch.setenv('SD_NO_GN_DEFER', '1')
    plain = DenseModel(sd, act_dtype=act, device=gpu)
    monkeypatch.delenv('SD_NO_GN_DEFER')
    monkeypatch.setenv('SD_NO_GN_POOL_RAW', '1')
    nofuse = DenseModel(sd, act_dtype=act, device=gpu)
    monkeypatch.delenv('SD_NO_GN_POOL_RAW')
    fused = DenseModel(sd, act_dtype=act, device=gpu)
    for shape in ((2, 16, 32, 48), (3, 13, 37, 43), (1, 32, 112, 144)):
        x = _input(shape, 5).to(gpu)
        for kind in (L.SD_OUT_LOGITS_F32, L.SD_OUT_PROBS_U8):
            a = plain.forward_batch(x, kind).clone()
            n0 = nofuse.forward_batch(x, kind, slot=1).clone()
            b = fused.forward_batch(x, kind, slot=1)
            assert torch.equal(a, b) and torch.equal(n0, b), (act, shape, kind)
    # ... and the fp32 oracle agrees within the usual tolerance (negative gammas go through the minimum path)
    model = build_unet('mivcsj', seed=9)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if '.norm' in k and k.endswith('.weight'):
                v.mul_((torch.randint(0, 2, v.shape, generator=g) * 2 - 1).to(v.dtype))
    _run_case(gpu, model, (12, 35, 41), act)


def test_op_kernels_report_what_ran(gpu, monkeypatch):
    """sd_debug_op_kernel: every plan op names the launch that computed it and that launch's kernel symbol -- the fused first
    convolution (uint8 input: MODE 5, float32 input: MODE 1), poolings in conv epilogues, the level-0 decoder's members."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    dm = DenseModel(random_state_dict('semseg_spine', seed=0, final_scale=8.0), 'bf16', gpu)
    x = torch.randint(0, 256, (4, 64, 128, 128), dtype=torch.uint8, device=gpu)
    ids = list(range(1, dm.out_channels))
    dm.forward_labels_batch(x, ids, [127.5] * len(ids))
    ks = dm.op_kernels()
    kinds = dm.op_kinds
    assert len(ks) == dm.n_ops and all(e >= 0 and name for e, name in ks)
    assert ks[0][0] == 1 and 'MODE=5' in ks[1][1] and ks[1][0] == 1                      # first conv inside the second
    for i, k in enumerate(kinds):
        if k == L.SD_OP_POOL:
            assert ks[i][0] == i - 1 and ks[i][1].startswith('k_conv_mfma')             # pooled in the producing conv's epilogue
    up = max(i for i, k in enumerate(kinds) if k == L.SD_OP_UPCONV)
    assert all(ks[i] == (up, 'k_dec0<labels>') for i in range(up, dm.n_ops))           # level-0 decoder: one streaming launch
    dm.forward_batch(x.float() / 255., L.SD_OUT_PROBS_U8)
    ks = dm.op_kernels()
    assert 'MODE=1' in ks[1][1] and ks[-1] == (up, 'k_dec0<probs u8>')
    monkeypatch.setenv('SD_NO_FUSE', '1')
    plain = DenseModel(random_state_dict('semseg_spine', seed=0, final_scale=8.0), 'bf16', gpu)
    plain.forward_batch(x, L.SD_OUT_PROBS_U8)
    kp = plain.op_kernels()
    assert all(e == i for i, (e, _) in enumerate(kp)) and kp[0][1].startswith('k_conv_first<uint8 input, bf16 MFMA')
