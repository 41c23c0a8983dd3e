"""GPU parity of the drop-in API (Predictor / dense_predicton_helper / dense_predictor / predict_dense_to_kd and the
device-side tiling + label kernels) against the oracle and the committed golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.predictor_ref import PredictorRef, dense_predicton_helper_ref, label_rule_ref
from oracle.unet_ref import build_cnn3, build_unet

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')

# stated tolerance on softmax probabilities for bf16 storage (fp32 accumulate): absolute, on p in [0,1]
TOL_P = 2.5e-2
# => on uint8 probabilities floor(255 p): 255*TOL_P + 1 levels
TOL_U8 = int(255 * TOL_P) + 1


def test_tile_gather_scatter_identity(gpu):
    """K1/K12: zero-padded tile extraction + crop-only stitching reproduce the volume (pins the index math)."""
    from syconn_amd.engine import tile_gather, tile_scatter
    rng = np.random.default_rng(0)
    vol = torch.from_numpy(rng.random((24, 40, 36)).astype(np.float32)).to(gpu)
    tile, ol = np.array([12, 20, 18]), np.array([4, 6, 5])
    out = torch.zeros((1, 24, 40, 36), device=gpu)
    for pos in np.ndindex(2, 2, 2):
        lo = tile * np.array(pos)
        t = tile_gather(vol, lo - ol, tile + 2 * ol)
        ref = torch.zeros(tuple(tile + 2 * ol))
        padded = torch.zeros((24 + 8, 40 + 12, 36 + 10))
        padded[4:-4, 6:-6, 5:-5] = vol.cpu()
        ref = padded[lo[0]:lo[0] + 20, lo[1]:lo[1] + 32, lo[2]:lo[2] + 28]
        assert torch.equal(t.cpu(), ref)
        tile_scatter(t[None].contiguous(), ol, tile, out, lo)
    assert torch.equal(out[0], vol)
    v8 = torch.from_numpy(rng.integers(0, 256, (9, 11, 13), dtype=np.uint8)).to(gpu)
    t8 = tile_gather(v8, (-2, 3, 5), (8, 8, 8)).cpu()
    assert t8[:2].sum() == 0 and torch.equal(t8[2:, :, :], v8.cpu()[:6, 3:11, 5:13])


def test_label_rule_on_device_matches_reference_goldens(gpu):
    """K11 against vectors produced by the reference's own label code (prediction.py:813-833)."""
    from syconn_amd.engine import postproc_labels
    from syconn_amd.handler.prediction import _resolve_threshold
    g = np.load(f'{G}/g3_label_rule.npz')
    cases = json.load(open(f'{G}/g3_label_rule_cases.json'))
    pred = torch.from_numpy(g['pred']).to(gpu)
    n = 0
    for name, c in cases.items():
        for j, ids in enumerate(c['target_channels']):
            if len(ids) < 2:
                continue
            thr = [_resolve_threshold(c['thresholds'][i]) for i in ids]
            for dt in (torch.uint8, torch.int64):
                out = postproc_labels(pred, ids, thr, out_dtype=dt).cpu().numpy().astype(np.uint64)
                assert np.array_equal(out, g[f'{name}_{j}_data']), (name, j)
            n += 1
    assert n >= 5
    # exhaustive: every uint8 value against awkward thresholds, vs numpy
    vals = torch.arange(256, dtype=torch.uint8).repeat(2, 1).reshape(2, 1, 16, 16).contiguous().to(gpu)
    for t in [0.0, 0.2, 0.3, 50.99999999999999, 51.00000000000001, 127.5, 254.999, 255.0, 300.0]:
        tt = _resolve_threshold(t)
        ref, _ = label_rule_ref(vals.cpu().numpy(), (1, 1), [None, t])
        out = postproc_labels(vals, (1, 1), [tt, tt]).cpu().numpy()
        assert np.array_equal(out, ref.astype(np.uint8)), t


def _em_like(shape, seed):
    """structured synthetic EM: smoothed noise rescaled to 0..255."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    v = gaussian_filter(rng.random(shape), 2.0)
    v = (v - v.min()) / (v.max() - v.min())
    return (v * 255).astype(np.uint8)


def test_predictor_tiled_matches_oracle(gpu):
    """Predictor.predict with 2x2x2 tiles + halo vs the oracle's tiled_apply (same tile grid, same zero padding)."""
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('myelin', seed=101, n_blocks=3, start_filts=8)
    g = np.load(f'{G}/g5_tiled_apply.npz')
    vol = g['vol']
    kw = dict(tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6), out_shape=(2, 24, 40, 40), strict_shapes=True)
    p = Predictor(model, apply_softmax=True, **kw)
    out = p.predict(vol)
    assert isinstance(out, torch.Tensor) and out.device.type == 'cpu' and out.dtype == torch.float32
    assert tuple(out.shape) == (1, 2, 24, 40, 40)
    err = np.abs(out.numpy() - g['unet_tiled_probs']).max()
    print('tiled probs max abs err vs oracle golden', err)
    assert err < TOL_P
    # float64 input (the reference's warm-up passes float64 zeros, prediction.py:781) and Tensor input
    out64 = p.predict(vol.astype(np.float64))
    assert torch.equal(out64, out)
    assert torch.equal(p.predict(torch.from_numpy(vol)), out)
    # tiling must change nothing but the border effects: tile-interior == same call through the oracle
    ref = PredictorRef(model, apply_softmax=True, **kw).predict(vol)
    assert np.abs(out.numpy() - ref.numpy()).max() < TOL_P
    # logits + argmax options
    lg = Predictor(model, apply_softmax=False, **kw).predict(vol)
    assert torch.allclose(lg.softmax(1), out, atol=1e-5)
    am = Predictor(model, apply_softmax=True, apply_argmax=True, **kw).predict(vol)
    assert torch.equal(am, out.argmax(1))
    # error conventions
    with pytest.raises(ValueError):
        Predictor(model, tile_shape=(11, 20, 20), overlap_shape=(0, 0, 0), out_shape=(2, 24, 40, 40),
                  strict_shapes=True).predict(vol)
    with pytest.raises(ValueError):
        Predictor(model, out_shape=(3, 24, 40, 40)).predict(vol)
    with pytest.raises(ValueError):
        Predictor('/nonexistent/model.pts')
    # non-strict shapes: trailing partial tiles are zero-padded and cropped
    pn = Predictor(model, tile_shape=(16, 32, 32), overlap_shape=(4, 6, 6), strict_shapes=False)
    on = pn.predict(vol)
    assert tuple(on.shape) == (1, 2, 24, 40, 40) and bool(torch.isfinite(on).all())


def test_predictor_loads_torchscript_like_syconn(gpu, tmp_path):
    """SyConn ships TorchScript traces (cnn_myelin.py:107 -> model.pts, loaded by path at prediction.py:777)."""
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('myelin', seed=7, n_blocks=3, start_filts=16)
    ts = torch.jit.trace(model, torch.randn(1, 1, 8, 16, 16))
    path = str(tmp_path / 'model.pts')
    ts.save(path)
    vol = (_em_like((16, 32, 32), 3).astype(np.float32) / 255.)[None, None]
    a = Predictor(path).predict(vol)
    b = Predictor(model).predict(vol)
    assert torch.equal(a, b)
    with torch.no_grad():
        ref = model(torch.from_numpy(vol)).softmax(1)
    assert (a - ref).abs().max() < TOL_P
    torch.save(model, str(tmp_path / 'model.pt'))
    assert torch.equal(Predictor(str(tmp_path / 'model.pt')).predict(vol), b)


def test_dense_predicton_helper_fast_path_is_bit_identical(gpu):
    """uint8 on the device (LUT normalisation, fused floor(255 p)) == reference-order host math on the same
    device probabilities (raw.astype(f32)/255 -> predict -> *255 -> astype(uint8))."""
    from syconn_amd.handler.prediction import Predictor, dense_predicton_helper
    model = build_unet('semseg_spine', seed=8, n_blocks=3, start_filts=16, final_scale=4.0)
    raw = _em_like((24, 40, 40), 4)
    p = Predictor(model, tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6), out_shape=(5, 24, 40, 40),
                  strict_shapes=True, apply_softmax=True)
    slow = dense_predicton_helper(raw.astype(np.float32) / 255., p, is_zyx=True, return_zyx=True)
    fast = p.predict_proba_u8_device(torch.from_numpy(raw).to(gpu)).cpu().numpy()
    assert slow.dtype == np.uint8 and np.array_equal(slow, fast)
    # and against the oracle within the stated tolerance
    ref = dense_predicton_helper_ref(raw.astype(np.float32) / 255.,
                                     PredictorRef(model, tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6),
                                                  out_shape=(5, 24, 40, 40), strict_shapes=True), True, True)
    d = np.abs(slow.astype(np.int16) - ref.astype(np.int16))
    print('uint8 prob diff vs oracle: max', d.max(), 'mean', d.mean())
    assert d.max() <= TOL_U8
    # xyz in / xyz out
    xyz = dense_predicton_helper(np.ascontiguousarray((raw.astype(np.float32) / 255.).swapaxes(0, 2)), p)
    assert np.array_equal(xyz, slow.swapaxes(-1, -3))


def test_config1_cnn3_end_to_end(gpu):
    """BASELINE.json config 1 through the HIP path vs the committed oracle output."""
    from syconn_amd.handler.prediction import Predictor, dense_predicton_helper
    g = np.load(f'{G}/g6_config1.npz')
    vol = np.random.default_rng(0).integers(0, 256, (64, 64, 64), dtype=np.uint8)
    p = Predictor(build_cnn3(0), tile_shape=(32, 32, 32), overlap_shape=(8, 8, 8), out_shape=(2, 64, 64, 64),
                  strict_shapes=True, apply_softmax=True)
    out = dense_predicton_helper(vol.astype(np.float32) / 255., p, is_zyx=True, return_zyx=True)
    d = np.abs(out.astype(np.int16) - g['out_u8'].astype(np.int16))
    print('config 1 uint8 diff: max', d.max(), 'frac != 0:', (d > 0).mean())
    assert out.shape == (2, 64, 64, 64) and d.max() <= 3
    assert np.array_equal(out.argmax(0)[d.max(0) == 0], g['out_u8'].argmax(0)[d.max(0) == 0])


def test_config1_cnn3_reference_precision_is_bit_exact(gpu):
    """BASELINE.json config 1 ("plumbing, bit for bit") in the reference-precision mode (act_dtype='f32'): the uint8 output of
    the whole drop-in chain -- Predictor tiling with overlap, fp32 network, softmax, *255 truncation, axis handling -- equals the
    committed oracle output (torch CPU, generated in the build container) and the argmax labels are identical.  (Against
    torch-CPU run on THIS box the fp32 logits differ by <= 2e-7: its oneDNN kernel sums in another order)"""
    from syconn_amd.handler.prediction import Predictor, dense_predicton_helper
    g = np.load(f'{G}/g6_config1.npz')
    vol = np.random.default_rng(0).integers(0, 256, (64, 64, 64), dtype=np.uint8)
    p = Predictor(build_cnn3(0), tile_shape=(32, 32, 32), overlap_shape=(8, 8, 8), out_shape=(2, 64, 64, 64),
                  strict_shapes=True, apply_softmax=True, act_dtype='f32')
    out = dense_predicton_helper(vol.astype(np.float32) / 255., p, is_zyx=True, return_zyx=True)
    d = np.abs(out.astype(np.int16) - g['out_u8'].astype(np.int16))
    print('config 1 (f32 mode) uint8 diff: max', d.max(), 'count != 0:', int((d > 0).sum()), 'of', d.size)
    assert out.shape == (2, 64, 64, 64) and out.dtype == np.uint8
    assert int((d > 0).sum()) <= 2 and d.max() <= 1          # (a product within one float32 ulp of an integer may truncate either way)
    assert np.array_equal(out.argmax(0), g['out_u8'].argmax(0))


def _make_wd(tmp_path, model, arch_name, shape_xyz, seed, geo, ngpus=1):
    from syconn_amd import global_params
    from syconn_amd.handler.config import generate_default_conf
    from syconn_amd.knossos import KnossosDataset
    wd = str(tmp_path / 'wd')
    kd_path = str(tmp_path / 'kd_raw')
    generate_default_conf(wd, scaling=(10, 10, 25), kd_seg=kd_path,
                          key_value_pairs=[('ngpus_per_node', ngpus), ('nnodes_total', 1), ('dense_prediction', geo)])
    os.makedirs(f'{wd}/models/{arch_name}', exist_ok=True)
    ts = torch.jit.trace(model, torch.randn(1, 1, 8, 16, 16))
    ext = 'pt' if arch_name == 'mivcsj' else 'pts'
    if ext == 'pts':
        ts.save(f'{wd}/models/{arch_name}/model.pts')
    else:
        torch.save(model, f'{wd}/models/{arch_name}/model.pt')
    kd = KnossosDataset()
    kd.initialize_without_conf(kd_path, boundary=shape_xyz, scale=(10, 10, 25), experiment_name='synth',
                               mags=[1, 2, 4])
    vol = _em_like(tuple(shape_xyz[::-1]), seed)
    kd.save_raw(offset=(0, 0, 0), mags=[1, 2, 4], data=vol, data_mag=1, fast_resampling=True, upsample=False)
    os.environ.pop('syconn_wd', None)
    global_params.wd = wd
    return wd, kd_path, vol


def _oracle_volume(model, vol, geo, box_xyz, n_channel):
    """Reference-order CPU computation of the whole volume: chunks + halo, tiles + halo, uint8, crop."""
    cs, ol, ts = (np.array(geo[k]) for k in ('chunk_size', 'overlap_shape_tiles', 'tile_shape'))
    nz, ny, nx = [int(np.ceil(box_xyz[i] / cs[i])) for i in (2, 1, 0)]
    out = np.zeros((n_channel, nz * cs[2], ny * cs[1], nx * cs[0]), np.uint8)
    pad = np.zeros(tuple(np.array(out.shape[1:]) + 2 * ol[::-1]), np.uint8)
    D, H, W = vol.shape
    pad[ol[2]:ol[2] + D, ol[1]:ol[1] + H, ol[0]:ol[0] + W] = vol
    pr = PredictorRef(model, tile_shape=ts[::-1], overlap_shape=ol[::-1],
                      out_shape=(n_channel, *(cs + 2 * ol)[::-1]), strict_shapes=True, apply_softmax=True)
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                z0, y0, x0 = iz * cs[2], iy * cs[1], ix * cs[0]
                raw = pad[z0:z0 + cs[2] + 2 * ol[2], y0:y0 + cs[1] + 2 * ol[1], x0:x0 + cs[0] + 2 * ol[0]]
                pred = dense_predicton_helper_ref(raw.astype(np.float32) / 255., pr, True, True)
                pred = pred[..., ol[2]:-ol[2], ol[1]:-ol[1], ol[0]:-ol[0]]
                out[:, z0:z0 + cs[2], y0:y0 + cs[1], x0:x0 + cs[0]] = pred
    return out[:, :D, :H, :W]


def test_predict_myelin_end_to_end(gpu, tmp_path):
    """exec_dense_prediction.predict_myelin -> predict_dense_to_kd -> worker process -> dense_predictor on a small
    synthetic KnossosDataset (mag 4 like the reference), compared with the oracle run in reference order."""
    from syconn_amd import global_params
    from syconn_amd.exec.exec_dense_prediction import predict_myelin
    from syconn_amd.handler.basics import kd_factory
    model = build_unet('myelin', seed=11, n_blocks=3, start_filts=8, final_scale=4.0)
    geo = {'overlap_shape_tiles': [6, 6, 4], 'chunk_size': [40, 40, 24], 'tile_shape': [26, 26, 16],
           'act_dtype': 'bf16'}
    shape_xyz = (280, 240, 160)                          # at mag 4: 70 x 60 x 40 -> 2x2x2 chunks, 8 tiles each
    wd, kd_path, vol = _make_wd(tmp_path, model, 'myelin', shape_xyz, 21, geo)
    predict_myelin()
    kd_out = kd_factory(f'{wd}/knossosdatasets/myelin/')
    got = kd_out.load_raw(size=shape_xyz, offset=(0, 0, 0), mag=4)                  # z,y,x at mag 4
    vol4 = vol[::4, ::4, ::4]
    ref = _oracle_volume(model, vol4, geo, (70, 60, 40), 2)[1]
    assert got.shape == ref.shape
    assert int(ref.max()) - int(ref.min()) >= 16         # (a probability map with a spread, not a constant)
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    print('myelin prob map vs oracle: max diff', d.max(), 'mean', d.mean())
    assert d.max() <= TOL_U8
    # pyramid written with order-0 resampling (mags [4, 8, 16])
    got8 = kd_out.load_raw(size=shape_xyz, offset=(0, 0, 0), mag=8)
    assert np.array_equal(got8, got[::2, ::2, ::2][:got8.shape[0], :got8.shape[1], :got8.shape[2]])
    # existing target without overwrite -> ValueError (prediction.py:686-691)
    with pytest.raises(ValueError):
        predict_myelin()
    global_params.wd = None


def test_predict_cellorganelles_labels_end_to_end(gpu, tmp_path):
    """3-head mivcsj target: GroupNorm U-Net, label rule with default thresholds, uint64 overlay (config 5 path)."""
    from syconn_amd import global_params
    from syconn_amd.exec.exec_dense_prediction import predict_cellorganelles
    from syconn_amd.handler.basics import kd_factory
    model = build_unet('mivcsj', seed=12, n_blocks=3, start_filts=8, final_scale=6.0)
    geo = {'overlap_shape_tiles': [6, 6, 4], 'chunk_size': [40, 40, 24], 'tile_shape': [26, 26, 16],
           'act_dtype': 'f16'}
    shape_xyz = (70, 60, 40)
    wd, kd_path, vol = _make_wd(tmp_path, model, 'mivcsj', shape_xyz, 22, geo)
    predict_cellorganelles()
    kd_out = kd_factory(f'{wd}/knossosdatasets/mivcsj/')
    got = kd_out.load_seg(size=shape_xyz, offset=(0, 0, 0), mag=1)
    assert got.dtype == np.uint64
    probs = _oracle_volume(model, vol, geo, shape_xyz, 4)
    ref, raw_flag = label_rule_ref(probs, (1, 2, 3), [None] * 4)
    assert not raw_flag
    # label must match wherever every involved uint8 probability is further than the tolerance from its threshold
    tol = 12                                                    # fp16 storage + per-channel GroupNorm of a tiny net
    safe = np.all(np.abs(probs[1:].astype(np.int16) - 127.5) > tol, axis=0)
    mism = got != ref
    print(f'labels: {safe.mean():.3f} of voxels margin-safe; mismatches safe {int((mism & safe).sum())}, '
          f'unsafe {int((mism & ~safe).sum())}; label hist {np.bincount(ref.ravel().astype(np.int64), minlength=4)}')
    assert not (mism & safe).any() and safe.mean() > 0.5
    assert len(np.unique(ref)) >= 2
    # on-disk overlay format: snappy-in-zip cubes, pyramid [1, 2, 4] by order-0 resampling
    import glob
    assert glob.glob(f'{wd}/knossosdatasets/mivcsj/mag1/x0000/y0000/z0000/*_mag1_x0000_y0000_z0000.seg.sz.zip')
    got2 = kd_out.load_seg(size=shape_xyz, offset=(0, 0, 0), mag=2)
    assert np.array_equal(got2, got[::2, ::2, ::2][:got2.shape[0], :got2.shape[1], :got2.shape[2]])
    global_params.wd = None


def test_downsample2_is_strided_pick(gpu):
    """sd_downsample2 == data[::2, ::2, ::2] (the order-0 mag pyramid), uint8 and 8-byte labels, odd extents."""
    from syconn_amd.engine import downsample2, mag_pyramid
    g = torch.Generator().manual_seed(3)
    for shape in ((7, 9, 11), (64, 48, 130), (1, 1, 1), (33, 2, 65)):
        a = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        assert torch.equal(downsample2(a.to(gpu)).cpu(), a[::2, ::2, ::2])
        b = torch.randint(0, 2 ** 62, shape, dtype=torch.int64, generator=g)
        assert torch.equal(downsample2(b.to(gpu)).cpu(), b[::2, ::2, ::2])
    a = torch.randint(0, 256, (41, 63, 65), dtype=torch.uint8, generator=g)
    lv = mag_pyramid(a.to(gpu), 3)
    assert torch.equal(lv[1].cpu(), a[::2, ::2, ::2]) and torch.equal(lv[2].cpu(), a[::4, ::4, ::4])


def test_config3_volume_in_overlapping_chunks(gpu):
    """BASELINE config 3 at reduced scale: a volume cut into overlapping chunks (useful region + halo = one model
    input tile, zeros outside the volume), predicted chunk by chunk through parallel.predict_volume_distributed
    (world size 1 here; the 2-rank scatter/gather logic is covered by tests/test_distributed_cpu.py) and compared
    with the oracle run in the same geometry."""
    from syconn_amd import parallel as par
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('semseg_axon', seed=31, start_filts=16, final_scale=4.0)
    vol_shape, chunk, halo = (60, 100, 90), (28, 48, 48), (4, 8, 8)
    vol = _em_like(vol_shape, 9)
    tile = tuple(c + 2 * h for c, h in zip(chunk, halo))
    p = Predictor(model, apply_softmax=True, device=gpu)

    def predict_fn(ch):
        pr = p.predict_proba_u8_device(ch)
        return pr[:, halo[0]:-halo[0], halo[1]:-halo[1], halo[2]:-halo[2]].contiguous()
    out = par.predict_volume_distributed(torch.from_numpy(vol), vol_shape, chunk, halo, predict_fn, n_out=6,
                                         device=gpu).cpu().numpy()
    assert out.shape == (6, *vol_shape)
    # oracle, same chunk grid / zero padding
    grid = [-(-vol_shape[i] // chunk[i]) for i in range(3)]
    pad = np.zeros([g * c + 2 * h for g, c, h in zip(grid, chunk, halo)], np.uint8)
    pad[halo[0]:halo[0] + vol_shape[0], halo[1]:halo[1] + vol_shape[1], halo[2]:halo[2] + vol_shape[2]] = vol
    ref = np.zeros((6, *[g * c for g, c in zip(grid, chunk)]), np.uint8)
    pr = PredictorRef(model, tile_shape=tile, overlap_shape=(0, 0, 0), out_shape=(6, *tile), strict_shapes=True)
    for iz in range(grid[0]):
        for iy in range(grid[1]):
            for ix in range(grid[2]):
                z, y, x = iz * chunk[0], iy * chunk[1], ix * chunk[2]
                raw = pad[z:z + tile[0], y:y + tile[1], x:x + tile[2]]
                u8 = dense_predicton_helper_ref(raw.astype(np.float32) / 255., pr, True, True)
                ref[:, z:z + chunk[0], y:y + chunk[1], x:x + chunk[2]] = \
                    u8[:, halo[0]:-halo[0], halo[1]:-halo[1], halo[2]:-halo[2]]
    ref = ref[:, :vol_shape[0], :vol_shape[1], :vol_shape[2]]
    d = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    print('config-3 style volume: uint8 prob diff max', d.max(), 'mean', d.mean())
    assert d.max() <= TOL_U8
    lab, _ = label_rule_ref(out, (1, 2, 3, 4, 5), [None] * 6)
    lab_ref, _ = label_rule_ref(ref, (1, 2, 3, 4, 5), [None] * 6)
    safe = np.all(np.abs(ref[1:].astype(np.int16) - 127.5) > TOL_U8, axis=0)
    assert not ((lab != lab_ref) & safe).any()


def test_config3_geometry_batching_and_streams_do_not_change_results(gpu):
    """BASELINE config 3 geometry (model-input tile 128^3 = useful (112,96,96) + halo (8,16,16) per side, semseg_axon)
    on a 336x288x288 volume = 27 tiles: tiles per launch set and HIP streams are scheduling choices only -- the uint8
    probabilities must be bit-identical for batch 1 / automatic (8) / 5 (ragged last set) and 1 / 2 streams."""
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('semseg_axon', seed=4, final_scale=6.0)
    g = torch.Generator().manual_seed(9)
    vol = torch.randint(0, 256, (336, 288, 288), dtype=torch.uint8, generator=g).to(gpu)
    outs = []
    for bs, ns in ((1, 1), (None, 1), (5, 2)):
        p = Predictor(model, tile_shape=(112, 96, 96), overlap_shape=(8, 16, 16), out_shape=(6, 336, 288, 288),
                      strict_shapes=True, apply_softmax=True, batch_size=bs, n_streams=ns)
        outs.append(p.predict_proba_u8_device(vol).clone())
        del p
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    s = outs[0].to(torch.int32).sum(0)
    assert int(s.max()) <= 255 and int(s.min()) > 255 - 6


def test_config3_full_volume_equals_per_tile_path(gpu):
    """BASELINE configs[2] at FULL size: semseg_axon, a 512^3 uint8 volume in overlapping 128^3 model tiles (useful
    (112,96,96) + halo (8,16,16), grid 5x6x6 = 180 tiles, the last tile of every axis ragged, zeros outside the volume).
    The stitched uint8 probabilities of the batched Predictor run must equal, bit for bit, what a SINGLE forward of each
    sampled tile (gathered with zero padding, cropped by the halo) gives: corners, faces, interior and the ragged ends."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel, tile_gather
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('semseg_axon', seed=4, final_scale=6.0)
    g = torch.Generator().manual_seed(2)
    vol = torch.randint(0, 256, (512, 512, 512), dtype=torch.uint8, generator=g).to(gpu)
    tile, ol = np.array((112, 96, 96)), np.array((8, 16, 16))
    p = Predictor(model, tile_shape=tuple(tile), overlap_shape=tuple(ol), out_shape=(6, 512, 512, 512), apply_softmax=True,
                  act_dtype='bf16')
    full = p.predict_proba_u8_device(vol)
    assert tuple(full.shape) == (6, 512, 512, 512)
    s = full[:, ::7, ::5, ::3].to(torch.int32).sum(0)
    assert int(s.max()) <= 255 and int(s.min()) > 255 - 6
    dm = DenseModel(model, act_dtype='bf16', device=gpu)
    for pos in ((0, 0, 0), (4, 5, 5), (2, 3, 1), (4, 0, 2), (0, 5, 0), (1, 2, 5), (3, 4, 4)):
        lo = tile * np.array(pos)
        keep = np.minimum(tile, 512 - lo)
        t = tile_gather(vol, lo - ol, tile + 2 * ol)
        one = dm.forward(t, L.SD_OUT_PROBS_U8)
        want = one[:, ol[0]:ol[0] + keep[0], ol[1]:ol[1] + keep[1], ol[2]:ol[2] + keep[2]]
        got = full[:, lo[0]:lo[0] + keep[0], lo[1]:lo[1] + keep[1], lo[2]:lo[2] + keep[2]]
        assert torch.equal(got, want), pos


def test_predict_labels_device_equals_probs_then_rule(gpu):
    """Predictor.predict_labels_u8_device (label rule in the final epilogue, tiled) == label rule applied to
    predict_proba_u8_device, incl. a ragged last launch set and a single-tile volume."""
    from syconn_amd.engine import postproc_labels
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('mivcsj', seed=8, n_blocks=3, start_filts=8, final_scale=6.0)
    g = torch.Generator().manual_seed(2)
    ids, thr = (1, 2, 3), (100.0, 90.5, 80.0)
    for shape, tile, ol, bs in (((48, 60, 72), (16, 20, 24), (4, 6, 6), 5), ((16, 24, 24), None, None, None)):
        vol = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).to(gpu)
        p = Predictor(model, tile_shape=tile, overlap_shape=ol, apply_softmax=True, batch_size=bs)
        want = postproc_labels(p.predict_proba_u8_device(vol), list(ids), list(thr))
        got = p.predict_labels_u8_device(vol, ids, thr)
        assert got.shape == want.shape and torch.equal(got, want)


def test_reference_geometry_chunk_properties(gpu):
    """One chunk in the reference's hard-coded geometry (prediction.py:672-677): chunk 482x481x236 + halo (30,31,20) =
    (276,543,542) zyx, 12 tiles of 138x181x271 + overlap (20,31,30) = model input 178x243x331 (odd extents at every
    level, 7.4 GiB of activations per tile), myelin U-Net.  Full size -> checked through properties: the result does
    not depend on the launch-set size / stream count, probabilities of a voxel sum to 255 up to truncation, and an
    interior tile equals the same region predicted on its own (tile + overlap read straight from the chunk)."""
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('myelin', seed=3, final_scale=6.0)
    g = torch.Generator().manual_seed(17)
    raw = torch.randint(0, 256, (276, 543, 542), dtype=torch.uint8, generator=g).to(gpu)
    kw = dict(strict_shapes=True, tile_shape=(138, 181, 271), out_shape=(2, 276, 543, 542), overlap_shape=(20, 31, 30),
              apply_softmax=True)
    a = Predictor(model, **kw).predict_proba_u8_device(raw)
    b = Predictor(model, n_streams=2, batch_size=2, **kw).predict_proba_u8_device(raw)
    assert torch.equal(a, b)
    s = a.to(torch.int32).sum(0)
    assert int(s.max()) <= 255 and int(s.min()) >= 253          # each of the 2 classes loses < 1, + float rounding
    # tile (z,y,x) index (0,1,0): rows 181..362; its model input = the chunk region incl. overlap, zero-padded outside
    sub = torch.zeros((178, 243, 331), dtype=torch.uint8, device=gpu)
    sub[20:, :, 30:] = raw[0:158, 150:393, 0:301]
    one = Predictor(model, apply_softmax=True).predict_proba_u8_device(sub)
    assert torch.equal(one[:, 20:158, 31:212, 30:301], a[:, 0:138, 181:362, 0:271])


def test_chunked_volume_prediction_equals_whole_volume_tiling(gpu):
    """BASELINE configs[2]-[4] geometry: chunks that carry a halo of real neighbouring data and continue the tile grid
    (`halo_included`) must reproduce the labels of the same tile grid run over the whole volume in one Predictor call --
    through parallel.predict_volume_distributed (world size 1: same code path as the multi-GPU run, no collectives)."""
    from syconn_amd import parallel as par
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.handler.prediction import Predictor
    sd = random_state_dict('semseg_spine', seed=3, final_scale=8.0, n_blocks=3, start_filts=16)
    tile, halo, chunk = (16, 24, 24), (4, 8, 8), (32, 48, 24)
    vol_shape = (70, 100, 50)                     # not a multiple of the chunk nor of the tile
    vol = torch.from_numpy(np.random.default_rng(4).integers(0, 256, vol_shape, dtype=np.uint8))
    pred = Predictor(sd, device=gpu, tile_shape=tile, overlap_shape=halo, apply_softmax=True)
    ids, thr = [1, 2, 3, 4], [127.5] * 4
    whole = pred.predict_labels_u8_device(vol.to(gpu), ids, thr).cpu()

    def predict_fn(ch):
        return pred.predict_labels_u8_device(ch, ids, thr, halo_included=True)[None]
    copies0 = par.HOST_BOX_COPIES
    for pipelined in (True, False):
        got = par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=1, device=gpu,
                                             pipelined=pipelined)
        assert got.shape == (1, *vol_shape) and torch.equal(got[0], whole)
    assert len(torch.unique(whole)) >= 2
    # on the device rank 0's CPU cuts and stitches nothing: the volume goes up in contiguous z-slabs, chunk + halo boxes are cut by
    # sd_tile_gather, results placed by sd_tile_scatter in a device-resident volume and downloaded as slabs (no host box copy per round)
    assert par.HOST_BOX_COPIES == copies0
    # a caller-owned (page-locked) result tensor is filled in place; a volume that is already on the device needs no upload
    mine = torch.empty((1, *vol_shape), dtype=torch.uint8).pin_memory()
    got = par.predict_volume_distributed(vol.to(gpu), vol_shape, chunk, halo, predict_fn, n_out=1, device=gpu, out=mine)
    assert got is mine and torch.equal(mine[0], whole) and par.HOST_BOX_COPIES == copies0
    with pytest.raises(ValueError):
        par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=1, device=gpu, out=mine[:, :-1])
    # rounds dealt from a cost-sorted chunk list (the Predictor's own cost model): the same volume
    cm = pred.chunk_cost_model(halo, True)
    got = par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=1, device=gpu, chunk_cost=lambda vb: cm.chunk_cost(chunk, vb))
    assert torch.equal(got[0], whole)
    # a volume whose device copy + result cannot fit rank 0's HBM is refused before anything is allocated or sent (RuntimeError = the
    # reference's out-of-memory convention)
    with pytest.raises(RuntimeError, match='do not fit'):
        par.predict_volume_distributed(vol, (200000, 200000, 64), chunk, halo, predict_fn, n_out=1, device=gpu)


def test_two_concurrent_workers_write_the_same_dataset_as_one(gpu, tmp_path, monkeypatch):
    """predict_dense_to_kd with chunk ids dealt to TWO worker processes that run at the same time (on this box: both on
    the one GPU) and whose chunks (40x40x24) share the 64^3 target cubes: the written KnossosDataset -- probability map and
    its pyramid -- must be byte-identical to the one a single worker writes.  Exercises chunkify -> pickle-stream jobs ->
    concurrent dense_predictor processes -> per-cube locked read-modify-write."""
    from syconn_amd import global_params
    from syconn_amd.handler import prediction as P
    from syconn_amd.handler.basics import kd_factory
    from syconn_amd.handler.config import generate_default_conf
    model = build_unet('myelin', seed=14, n_blocks=3, start_filts=8, final_scale=4.0)
    geo = {'overlap_shape_tiles': [6, 6, 4], 'chunk_size': [40, 40, 24], 'tile_shape': [26, 26, 16], 'act_dtype': 'f16'}
    shape_xyz = (150, 130, 70)                               # 4 x 4 x 3 = 48 chunks over 3 x 3 x 2 cubes of 64^3
    wd, kd_path, vol = _make_wd(tmp_path, model, 'myelin', shape_xyz, 23, geo)
    outs = {}
    for nworkers in (1, 2):
        # ngpu_total = nnodes_total * ngpus_per_node decides the number of jobs (prediction.py:708-709)
        generate_default_conf(wd, scaling=(10, 10, 20), kd_seg=kd_path,
                              key_value_pairs=[('ngpus_per_node', 1), ('nnodes_total', nworkers), ('dense_prediction', geo)])
        global_params.wd = wd
        global_params.config._load(wd)                          # same working dir, rewritten config.yml
        assert global_params.config.ngpu_total == nworkers
        monkeypatch.setenv('SYCONN_AMD_WORKERS_PER_GPU', str(nworkers))
        P.predict_dense_to_kd(kd_path, f'{wd}/knossosdatasets/', f'{wd}/models/myelin/model.pts', n_channel=2,
                              target_names=[f'myelin_w{nworkers}'], target_channels=[(1,)], mag=1, overwrite=True,
                              cube_shape_kd=(64, 64, 64))
        kd_out = kd_factory(f'{wd}/knossosdatasets/myelin_w{nworkers}/')
        outs[nworkers] = [kd_out.load_raw(size=shape_xyz, offset=(0, 0, 0), mag=m) for m in (1, 2, 4)]
    for a, b in zip(outs[1], outs[2]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert outs[1][0].std() > 1.0
    global_params.wd = None


def test_tiles_beyond_the_dataset_are_skipped_without_changing_the_inside(gpu):
    """`valid_box` (prediction.py:679-683: the fit_box_size chunk grid overhangs the dataset, the reference predicts all of it):
    tiles whose cropped result lies entirely outside the box are not predicted -- fewer launches, zeros there -- and every voxel
    inside the box is bit-identical to the full prediction; a box that touches every tile changes nothing."""
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('myelin', seed=2, n_blocks=3, start_filts=16, final_scale=4.0)
    g = torch.Generator().manual_seed(11)
    raw = torch.randint(0, 256, (24, 60, 64), dtype=torch.uint8, generator=g)
    raw[:, 40:, :] = 0
    raw[:, :, 45:] = 0                                   # (what kd.load_raw returns beyond the boundary)
    kw = dict(strict_shapes=True, tile_shape=(12, 20, 16), out_shape=(2, 24, 60, 64), overlap_shape=(4, 6, 6), apply_softmax=True,
              act_dtype='bf16', batch_size=5)
    p = Predictor(model, **kw)
    full = p.predict_proba_u8_device(raw.to(gpu)).cpu()
    box = ((4, 6, 6), (24, 40, 45))                      # chunk proper inside the dataset, chunk + halo coordinates
    part = p.predict_proba_u8_device(raw.to(gpu), valid_box=box).cpu()
    (z0, y0, x0), (z1, y1, x1) = box
    assert torch.equal(part[:, z0:z1, y0:y1, x0:x1], full[:, z0:z1, y0:y1, x0:x1])
    assert int(part[:, :, 40:, :].max()) == 0 and int(part[:, :, :, 48:].max()) == 0      # tile rows y >= 40, columns x >= 48: skipped
    assert int(full[:, :, 40:, :].max()) > 0
    lab_full = p.predict_labels_u8_device(raw.to(gpu), [1], [100.0]).cpu()
    lab_part = p.predict_labels_u8_device(raw.to(gpu), [1], [100.0], valid_box=box).cpu()
    assert torch.equal(lab_part[z0:z1, y0:y1, x0:x1], lab_full[z0:z1, y0:y1, x0:x1])
    same = p.predict_proba_u8_device(raw.to(gpu), valid_box=((0, 0, 0), (24, 60, 64))).cpu()
    assert torch.equal(same, full)


@pytest.mark.parametrize('arch,act', [('myelin', 'bf16'), ('myelin', 'f16x2'), ('semseg_axon', 'f16'), ('syntype', 'bf16'),
                                      ('mivcsj', 'f16')])
def test_boundary_tiles_on_clipped_windows_are_bit_identical(gpu, arch, act):
    """`clip_tiles` (Predictor._tiled / sd_plan_clip_window): tiles that reach beyond the `valid_box` run on the part of their
    window the voxels inside the box depend on.  Full-width networks (the fused level-0 forms, the streaming decoder, the
    split plan), 2 x 2 x 2 tiles, the box ends inside the second tile of every axis: probabilities and labels inside the box are
    bit-identical to full windows, the clipped windows are really smaller, and a GroupNorm network keeps full windows."""
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.handler.prediction import Predictor
    sd = random_state_dict(arch, seed=4, final_scale=4.0)
    tile, ol, shape = (24, 96, 128), (4, 8, 8), (48, 192, 256)
    box = ((12, 60, 60), (30, 110, 150))          # (near side: as if a wide halo ring were cropped afterwards)
    g = torch.Generator().manual_seed(12)
    raw = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
    raw[30:], raw[:, 110:], raw[:, :, 150:] = 0, 0, 0          # (what kd.load_raw returns beyond the boundary)
    nc = random_state_dict(arch, seed=4)['conv_final.bias'].numel()
    kw = dict(strict_shapes=True, tile_shape=tile, out_shape=(nc, *shape), overlap_shape=ol, apply_softmax=True, act_dtype=act)
    clip, full = Predictor(sd, **kw), Predictor(sd, clip_tiles=False, **kw)
    wins = [clip._dm.clipped_window(o, o + w, t + 2 * o, a)[1] for a, (o, w, t) in enumerate(zip(ol, (6, 14, 22), tile))]
    starts = [clip._dm.clipped_window(o + w, t + o, t + 2 * o, a)[0] for a, (o, w, t) in enumerate(zip(ol, box[0], tile))]
    if arch == 'mivcsj':
        assert wins == [t + 2 * o for t, o in zip(tile, ol)] and starts == [0, 0, 0]
    else:
        assert all(w <= t + 2 * o for w, t, o in zip(wins, tile, ol)) and wins[1] < tile[1] + 2 * ol[1] \
            and wins[2] < tile[2] + 2 * ol[2], wins           # (z: syntype's reach exceeds the 32-plane window)
        assert starts[1] >= 8 and starts[2] >= 8, starts
    x = raw.to(gpu)
    (z0, y0, x0), (z1, y1, x1) = box
    a = clip.predict_proba_u8_device(x, valid_box=box).cpu()
    b = full.predict_proba_u8_device(x, valid_box=box).cpu()
    assert torch.equal(a[:, z0:z1, y0:y1, x0:x1], b[:, z0:z1, y0:y1, x0:x1]) and len(torch.unique(b[:, z0:z1, y0:y1, x0:x1])) >= 16
    assert int(a[:, z1:].max()) == 0 and int(a[:, :, y1:].max()) == 0 and int(a[:, :, :, x1:].max()) == 0   # beyond the box: zeros
    assert int(a[:, :z0].max()) == 0 and int(a[:, :, :y0].max()) == 0 and int(a[:, :, :, :x0].max()) == 0
    ids, thr = list(range(1, nc)), [110.0] * (nc - 1)
    la = clip.predict_labels_u8_device(x, ids, thr, valid_box=box).cpu()
    lb = full.predict_labels_u8_device(x, ids, thr, valid_box=box).cpu()
    assert torch.equal(la[z0:z1, y0:y1, x0:x1], lb[z0:z1, y0:y1, x0:x1]) and len(torch.unique(lb)) >= 2
    # without a box only the volume's own end clips, and there is none here: identical everywhere
    assert torch.equal(clip.predict_proba_u8_device(x).cpu(), full.predict_proba_u8_device(x).cpu())
    assert not clip.overflowed() and not full.overflowed()


def test_config3_full_width_model_two_by_two_by_two_tiles_vs_oracle(gpu):
    """BASELINE configs[2] with the FULL-WIDTH semseg_axon model (48 filters, 9.1 M parameters) on a 224 x 192 x 192 volume =
    2 x 2 x 2 model tiles of 128^3 in the config's geometry (useful (112,96,96) + halo (8,16,16), zeros outside the volume):
    the Predictor's tiled, stitched uint8 probabilities against the torch-CPU oracle run through the same tiled_apply
    restatement -- in the reference-precision plan (what `float16=False` selects) within one uint8 level on a vanishing fraction
    of the voxels and with identical argmax, in bf16 within the stated storage tolerance."""
    from syconn_amd.handler.prediction import Predictor
    model = build_unet('semseg_axon', seed=12, final_scale=6.0)
    vol_shape, tile, halo = (224, 192, 192), (112, 96, 96), (8, 16, 16)
    vol = _em_like(vol_shape, 21)
    kw = dict(tile_shape=tile, overlap_shape=halo, out_shape=(6, *vol_shape), strict_shapes=True, apply_softmax=True)
    ref = dense_predicton_helper_ref(vol.astype(np.float32) / 255., PredictorRef(model, **kw), True, True)
    for act, max_lvl, frac, agree in (('f16x2', 1, 2e-3, 0.99999), ('bf16', 12, 1.0, 0.998)):
        got = Predictor(model, device=gpu, act_dtype=act, **kw).predict_proba_u8_device(torch.from_numpy(vol).to(gpu)).cpu().numpy()
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
        same = float((got.argmax(0) == ref.argmax(0)).mean())
        print(f'config-3 geometry, full-width semseg_axon, {act}: max level diff {d.max()}, frac != 0 {float((d > 0).mean()):.2e}, '
              f'argmax agreement {same:.6f}')
        assert got.shape == ref.shape == (6, *vol_shape)
        assert d.max() <= max_lvl and float((d > 0).mean()) <= frac and same >= agree, (act, int(d.max()), same)


def test_default_configuration_end_to_end_is_reference_precision(gpu, tmp_path):
    """The drop-in with NOTHING configured (no `act_dtype`, tile skipping at its default): predict_myelin and
    predict_cellorganelles on a volume whose chunk grid overhangs the dataset run in the reference-precision plan 'f16x2' --
    the myelin probability map within one uint8 level of the oracle on a vanishing fraction of the voxels, the mivcsj labels
    equal to the oracle's wherever every probability is at least two levels from its threshold (and on > 99.9 % overall)."""
    from syconn_amd import global_params
    from syconn_amd.exec.exec_dense_prediction import predict_cellorganelles, predict_myelin
    from syconn_amd.handler.basics import kd_factory
    geo = {'overlap_shape_tiles': [6, 6, 4], 'chunk_size': [40, 40, 24], 'tile_shape': [26, 26, 16]}      # geometry only
    model = build_unet('myelin', seed=11, n_blocks=3, start_filts=8, final_scale=4.0)
    shape_xyz = (260, 200, 112)                          # at mag 4: 65 x 50 x 28 -> 2 x 2 x 2 chunks, the last ones mostly outside
    wd, kd_path, vol = _make_wd(tmp_path, model, 'myelin', shape_xyz, 23, geo)
    assert global_params.config['dense_prediction']['act_dtype'] == 'f16x2'
    predict_myelin()
    got = kd_factory(f'{wd}/knossosdatasets/myelin/').load_raw(size=shape_xyz, offset=(0, 0, 0), mag=4)
    ref = _oracle_volume(model, vol[::4, ::4, ::4], geo, (65, 50, 28), 2)[1]
    assert int(ref.max()) - int(ref.min()) >= 16
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    print('default configuration, myelin vs oracle: max diff', d.max(), 'frac != 0', float((d > 0).mean()))
    assert got.shape == ref.shape and d.max() <= 1 and float((d > 0).mean()) < 2e-3
    global_params.wd = None
    model2 = build_unet('mivcsj', seed=12, n_blocks=3, start_filts=8, final_scale=6.0)
    shape2 = (70, 60, 40)
    wd2, _, vol2 = _make_wd(tmp_path / 'b', model2, 'mivcsj', shape2, 22, geo)
    predict_cellorganelles()
    lab = kd_factory(f'{wd2}/knossosdatasets/mivcsj/').load_seg(size=shape2, offset=(0, 0, 0), mag=1)
    probs = _oracle_volume(model2, vol2, geo, shape2, 4)
    want, _ = label_rule_ref(probs, (1, 2, 3), [None] * 4)
    safe = np.all(np.abs(probs[1:].astype(np.int16) - 127.5) > 2, axis=0)
    mism = lab != want
    print(f'default configuration, mivcsj labels: agreement {1 - mism.mean():.6f}, margin-safe mismatches {int((mism & safe).sum())}')
    assert not (mism & safe).any() and mism.mean() < 1e-3
    global_params.wd = None


@pytest.mark.parametrize('arch,act', [('myelin', 'bf16'), ('myelin', 'f16'), ('myelin', 'f16x2'), ('semseg_axon', 'bf16'),
                                      ('semseg_axon', 'f16x2'), ('syntype', 'f16'), ('mivcsj', 'f16')])
def test_output_box_of_interest_leaves_the_kept_values_unchanged(gpu, arch, act):
    """`sd_model_set_roi` (DenseModel.forward_batch(roi=...)): with an output box set, the decoder computes sub-boxes only (the
    fused level-0 decoder is replaced by its layers) -- probabilities, logits and labels INSIDE the box are bit-identical to a
    whole-tile pass for boxes in the middle, at a corner, one voxel thick and odd; batches of tiles; a GroupNorm network ignores
    the box; fewer launches are not required, identical values are."""
    from syconn_amd import _lib as L
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    sd = random_state_dict(arch, seed=6, final_scale=4.0)
    dm = DenseModel(sd, act_dtype=act, device=gpu)
    g = torch.Generator().manual_seed(3)
    shape = (40, 112, 144)
    x = torch.randint(0, 256, (2, *shape), dtype=torch.uint8, generator=g).to(gpu)
    full_p = dm.forward_batch(x, L.SD_OUT_PROBS_U8).cpu()
    full_l = dm.forward_batch(x, L.SD_OUT_LOGITS_F32).cpu()
    ids, thr = list(range(1, dm.out_channels)), [120.0] * (dm.out_channels - 1)
    full_lab = dm.forward_labels_batch(x, ids, thr).cpu()
    assert len(torch.unique(full_p)) >= 16 and len(torch.unique(full_lab)) >= 2      # (not a constant answer)
    for roi in (((8, 16, 16), (32, 96, 128)), ((0, 0, 0), (17, 33, 65)), ((39, 50, 3), (40, 51, 144)), ((5, 7, 9), (36, 101, 139))):
        (z0, y0, x0), (z1, y1, x1) = roi
        p = dm.forward_batch(x, L.SD_OUT_PROBS_U8, roi=roi).cpu()
        assert torch.equal(p[:, :, z0:z1, y0:y1, x0:x1], full_p[:, :, z0:z1, y0:y1, x0:x1]), roi
        lg = dm.forward_batch(x, L.SD_OUT_LOGITS_F32, roi=roi).cpu()
        assert torch.equal(lg[:, :, z0:z1, y0:y1, x0:x1], full_l[:, :, z0:z1, y0:y1, x0:x1]), roi
        lab = dm.forward_labels_batch(x, ids, thr, roi=roi).cpu()
        assert torch.equal(lab[:, z0:z1, y0:y1, x0:x1], full_lab[:, z0:z1, y0:y1, x0:x1]), roi
    assert torch.equal(dm.forward_batch(x, L.SD_OUT_PROBS_U8).cpu(), full_p)          # (the box is gone again)
    assert torch.equal(dm.forward(x[0], L.SD_OUT_PROBS_U8).cpu(), full_p[0])
    with pytest.raises(ValueError):
        dm.forward_batch(x, L.SD_OUT_PROBS_U8, roi=((0, 0, 0), (41, 10, 10)))
    assert not dm.overflowed()


def test_output_box_of_interest_on_the_sequential_config1_net(gpu):
    """BASELINE configs[0]'s 3-layer CNN (plan_from_sequential: first conv, conv, final) with an output box: values inside the box
    are those of the whole-tile pass in every storage type (the second convolution runs on a sub-box, the first one in full)."""
    from syconn_amd import _lib as L
    from syconn_amd.engine import DenseModel
    net = build_cnn3(0)
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, 256, (3, 24, 40, 56), dtype=torch.uint8, generator=g).to(gpu)
    for act in ('bf16', 'f16x2', 'f32'):
        dm = DenseModel(net, act_dtype=act, device=gpu)
        full = dm.forward_batch(x, L.SD_OUT_PROBS_U8).cpu()
        for roi in (((4, 8, 8), (20, 32, 48)), ((0, 0, 0), (3, 40, 5))):
            (z0, y0, x0), (z1, y1, x1) = roi
            got = dm.forward_batch(x, L.SD_OUT_PROBS_U8, roi=roi).cpu()
            assert torch.equal(got[:, :, z0:z1, y0:y1, x0:x1], full[:, :, z0:z1, y0:y1, x0:x1]), (act, roi)


def test_file_system_mode_datasets_are_byte_identical_with_and_without_clipping(gpu, tmp_path):
    """predict_dense_to_kd on one synthetic KnossosDataset with `clip_boundary_tiles` on (default: clipped windows, output
    boxes, sub-box decoder) and off (whole windows, as the reference computes): every cube file of the target dataset -- all
    three mags -- has the same bytes, and the probability map has a spread.  Full-width myelin net, chunks that overhang
    the dataset in x and y, the reference-precision plan."""
    import hashlib
    from syconn_amd import global_params
    from syconn_amd.exec.exec_dense_prediction import predict_myelin
    from syconn_amd.handler.basics import kd_factory
    model = build_unet('myelin', seed=3, final_scale=8.0)
    shape_xyz = (400, 360, 96)                                     # at mag 4: 100 x 90 x 24
    digests = {}
    for clip in (True, False):
        sub = tmp_path / f'clip{int(clip)}'
        sub.mkdir()
        geo = {'overlap_shape_tiles': [12, 12, 4], 'chunk_size': [64, 56, 24], 'tile_shape': [44, 40, 16], 'act_dtype': 'f16x2',
               'clip_boundary_tiles': clip}
        wd, kd_path, vol = _make_wd(sub, model, 'myelin', shape_xyz, 33, geo)
        predict_myelin()
        files = sorted(p for p in (sub / 'wd' / 'knossosdatasets' / 'myelin').rglob('*') if p.is_file() and p.suffix in ('.raw', '.zip', '.sz'))
        assert len(files) >= 3
        digests[clip] = {str(p.relative_to(sub)): hashlib.sha256(p.read_bytes()).hexdigest() for p in files}
        if clip:
            got = kd_factory(f'{wd}/knossosdatasets/myelin/').load_raw(size=shape_xyz, offset=(0, 0, 0), mag=4)
            assert int(got.max()) - int(got.min()) >= 16
        global_params.wd = None
    assert digests[True] == digests[False]
