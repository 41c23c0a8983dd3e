"""CPU: the product's host-side logic (no compute calls) against the reference-pinned golden vectors."""
import json
import os
import pickle
import re

import numpy as np
import pytest
import torch

from oracle.unet_ref import ARCHS, build_cnn3, build_unet

G = os.path.join(os.path.dirname(__file__), 'golden')


class StubPredictor:
    def predict(self, inp):
        x = torch.as_tensor(np.asarray(inp), dtype=torch.float32)
        return (torch.cat([x, 1 - x, torch.full_like(x, 0.5)], dim=1) * 3).softmax(1)


def test_wrapper_math_matches_reference_goldens():
    from syconn_amd.handler.prediction import dense_predicton_helper, xyz2zyx, zyx2xyz
    g = np.load(f'{G}/g1_wrapper.npz')
    raw = g['raw_xyz']
    for a in (0, 1):
        for b in (0, 1):
            out = dense_predicton_helper(raw, StubPredictor(), is_zyx=bool(a), return_zyx=bool(b))
            assert out.dtype == np.uint8 and np.array_equal(out, g[f'out_{a}{b}'])
    assert np.array_equal(xyz2zyx(raw), g['xyz2zyx']) and np.array_equal(zyx2xyz(raw), g['zyx2xyz'])


def test_chunkify_matches_reference_goldens():
    from syconn_amd.handler.basics import chunkify
    g = np.load(f'{G}/g2_chunkify.npz')
    for n_items, n in [(75, 8), (3, 8), (8, 8), (10, 3), (0, 4), (1, 1)]:
        parts = chunkify(list(range(n_items)), n)
        assert [len(p) for p in parts] == g[f'n{n_items}_k{n}_len'].tolist()
        assert [v for p in parts for v in p] == g[f'n{n_items}_k{n}_flat'].tolist()


def test_threshold_resolution_and_integer_cut():
    """The device label kernel compares uint8 >= cut with cut = floor(t)+1; check against numpy's uint8 > float."""
    from syconn_amd.handler.prediction import _resolve_threshold
    assert _resolve_threshold(None) == 127.5 and _resolve_threshold(0.3) == 255 * 0.3 and _resolve_threshold(100) == 100
    vals = np.arange(256, dtype=np.uint8)
    for t in [None, 0.0, 0.2, 0.3, 0.5, 0.999, 1.0, 50.99999999999999, 51.00000000000001, 127.5, 254.999, 255, 300]:
        tt = _resolve_threshold(t)
        cut = 0 if tt < 0 else (256 if tt >= 255 else int(np.floor(tt)) + 1)
        assert np.array_equal(vals > tt, vals.astype(np.int64) >= cut), t


@pytest.mark.parametrize('arch', sorted(ARCHS))
def test_plan_from_unet(arch):
    from syconn_amd import _lib as L
    from syconn_amd.plan import plan_from_model
    model = build_unet(arch, seed=0)
    ops, blob, info = plan_from_model(model)
    kw = ARCHS[arch]
    kinds = [o.kind for o in ops]
    nb = kw['n_blocks']
    assert kinds.count(L.SD_OP_POOL) == nb - 1 and kinds.count(L.SD_OP_UPCONV) == nb - 1
    assert kinds.count(L.SD_OP_CONV) == 2 * nb + 2 * (nb - 1) and kinds[-1] == L.SD_OP_FINAL
    assert ops[-1].cout == kw['out_channels'] and info['out_channels'] == kw['out_channels']
    n_gn = kinds.count(L.SD_OP_GROUPNORM)
    assert n_gn == (2 * nb + 3 * (nb - 1) if kw['normalization'].startswith('group') else 0)
    # planar blocks -> kz == 1 for convs and pools of that level
    convs = [o for o in ops if o.kind == L.SD_OP_CONV]
    for i in range(nb):
        assert convs[2 * i].kz == (1 if i in kw['planar_blocks'] else 3)
    # legacy batch_norm=True layout: no norm after the first conv of a block
    if not kw.get('full_norm', True):
        assert convs[0].norm == 0 and convs[1].norm == 1
    # the same plan comes out of a TorchScript trace (how SyConn ships models: cnn_myelin.py:107)
    if arch == 'myelin':
        ts = torch.jit.trace(model, torch.randn(1, 1, 8, 16, 16))
        ops2, blob2, _ = plan_from_model(ts)
        assert len(ops2) == len(ops) and np.array_equal(blob, blob2)
        for a, b in zip(ops, ops2):
            assert bytes(a) == bytes(b)


def test_plan_from_sequential_and_rejections():
    from syconn_amd import _lib as L
    from syconn_amd.plan import plan_from_model
    ops, blob, info = plan_from_model(build_cnn3(0))
    assert [o.kind for o in ops] == [L.SD_OP_CONV, L.SD_OP_CONV, L.SD_OP_FINAL] and ops[0].relu == 1
    with pytest.raises(ValueError):
        plan_from_model(torch.nn.Sequential(torch.nn.Conv3d(1, 4, 5, padding=2), torch.nn.Conv3d(4, 2, 1)))
    with pytest.raises(ValueError):
        plan_from_model({'foo.weight': torch.zeros(1)})


def test_knossos_roundtrip_and_chunk_grid(tmp_path):
    from syconn_amd.handler.basics import kd_factory
    from syconn_amd.knossos import ChunkDataset, KnossosDataset
    rng = np.random.default_rng(5)
    kd = KnossosDataset()
    kd.initialize_without_conf(str(tmp_path / 'raw'), boundary=(150, 140, 70), scale=(10, 10, 25),
                               experiment_name='synth', mags=[1, 2, 4])
    vol = rng.integers(0, 256, (70, 140, 150), dtype=np.uint8)   # z,y,x
    kd.save_raw(offset=(0, 0, 0), mags=[1, 2, 4], data=vol, data_mag=1, fast_resampling=True, upsample=False)
    kd2 = kd_factory(str(tmp_path / 'raw'))
    assert kd2.experiment_name == 'synth' and kd2.boundary.tolist() == [150, 140, 70]
    sub = kd2.load_raw(size=(40, 30, 20), offset=(100, 120, 60), mag=1)
    assert sub.shape == (20, 30, 40)
    ref = np.zeros((20, 30, 40), np.uint8)
    ref[:10, :20, :] = vol[60:70, 120:140, 100:140]
    assert np.array_equal(sub, ref)                                # zeros outside the boundary
    neg = kd2.load_raw(size=(20, 20, 20), offset=(-10, -10, -10), mag=1)
    assert np.array_equal(neg[10:, 10:, 10:], vol[:10, :10, :10]) and not neg[:10].any()
    m2 = kd2.load_raw(size=(150, 140, 70), offset=(0, 0, 0), mag=2)
    assert np.array_equal(m2, vol[::2, ::2, ::2][:35, :70, :75])
    lab = rng.integers(0, 2 ** 40, (10, 12, 14), dtype=np.uint64)
    kd2.save_seg(offset=(3, 4, 5), mags=[1], data=lab, data_mag=1)
    assert np.array_equal(kd2.load_seg(size=(14, 12, 10), offset=(3, 4, 5), mag=1), lab)
    # chunk grid of BASELINE config 4 in reference geometry: 2048x2048x512 -> 5x5x3 = 75 chunks
    class _KD:  # noqa
        boundary = np.array([2048, 2048, 512])
    cd = ChunkDataset()
    cd.initialize(_KD(), np.array([2048, 2048, 512]), np.array([482, 481, 236]), str(tmp_path / 'cd'),
                  box_coords=np.zeros(3), fit_box_size=True, overlap=np.array([30, 31, 20]))
    assert len(cd.chunk_dict) == 75
    assert cd.chunk_dict[1].coordinates.tolist() == [0, 0, 236] and cd.chunk_dict[3].coordinates.tolist() == [0, 481, 0]


def test_config_and_wrappers_bind_reference_parameters(tmp_path, monkeypatch):
    """exec_dense_prediction wrappers bind exactly the reference's parameters (exec_dense_prediction.py:52-150)."""
    from syconn_amd import global_params
    from syconn_amd.exec import exec_dense_prediction as E
    from syconn_amd.handler.config import generate_default_conf
    wd = str(tmp_path / 'wd')
    generate_default_conf(wd, scaling=(10, 10, 25), kd_seg='/data/kd_seg/',
                          key_value_pairs=[('ngpus_per_node', 4), ('nnodes_total', 2)])
    monkeypatch.delenv('syconn_wd', raising=False)
    global_params.wd = wd
    cfg = global_params.config
    assert cfg.working_dir == os.path.abspath(wd) and cfg.ngpu_total == 8 and cfg['scaling'] == [10, 10, 25]
    assert cfg.mpath_myelin.endswith('/models//myelin/model.pts') and cfg.mpath_mivcsj.endswith('/mivcsj/model.pt')
    assert cfg['dense_prediction']['chunk_size'] == [482, 481, 236]
    calls = []
    monkeypatch.setattr(E, 'predict_dense_to_kd', lambda *a, **k: calls.append((a, k)))
    E.predict_myelin(); E.predict_synapsetype(); E.predict_cellorganelles(); E.predict_er(); E.predict_golgi()
    got = [(k['n_channel'], k['mag'], list(map(tuple, k['target_channels'])), k['target_names']) for _, k in calls]
    assert got == [(2, 4, [(1,)], ['myelin']), (4, 1, [(1, 2)], ['syntype_v2']), (4, 1, [(1, 2, 3)], ['mivcsj']),
                   (2, 1, [(1,)], ['er']), (2, 1, [(1,)], ['golgi'])]
    assert calls[0][0][0] == '/data/kd_seg/' and calls[0][0][2] == cfg.mpath_myelin
    # env syconn_wd (set for worker processes) wins over global_params.wd
    wd2 = str(tmp_path / 'wd2')
    generate_default_conf(wd2)
    monkeypatch.setenv('syconn_wd', wd2)
    assert global_params.config.working_dir == os.path.abspath(wd2)
    monkeypatch.delenv('syconn_wd')
    global_params.wd = None


def test_predict_dense_to_kd_argument_errors(tmp_path, monkeypatch):
    """ValueError conventions of prediction.py:659-662 and :686-691 (no GPU needed: they fire before dispatch)."""
    from syconn_amd import global_params
    from syconn_amd.handler.config import generate_default_conf
    from syconn_amd.handler.prediction import predict_dense_to_kd
    from syconn_amd.knossos import KnossosDataset
    wd = str(tmp_path / 'wd')
    generate_default_conf(wd)
    monkeypatch.delenv('syconn_wd', raising=False)
    global_params.wd = wd
    kd = KnossosDataset()
    kd.initialize_without_conf(str(tmp_path / 'raw'), (64, 64, 64), (1, 1, 1), 'synth', mags=[1])
    with pytest.raises(ValueError):
        predict_dense_to_kd(str(tmp_path / 'raw'), str(tmp_path / 'out'), 'nomodel.pts', 2,
                            target_names=['a', 'b'], target_channels=[(1,)])
    os.makedirs(tmp_path / 'out' / 'pred')
    with pytest.raises(ValueError):
        predict_dense_to_kd(str(tmp_path / 'raw'), str(tmp_path / 'out'), 'nomodel.pts', 2, overwrite=False)
    global_params.wd = None


def test_job_pickle_stream_contract(tmp_path):
    """in.pkl = one pickle per tuple element (batchjob_utils.py:478-480), read back by the worker loop
    (batchjob_predict_dense.py:9-15)."""
    params = ([1, 2, 3], 'kd', 'target', 'model', np.array([30, 31, 20]), np.array([30, 31, 20]), [271, 181, 138],
              np.array([482, 481, 236]), 2, [(1,)], ['p/'], [None, None], 4, (np.zeros(3), np.ones(3)))
    fn = tmp_path / 'job_0.pkl'
    with open(fn, 'wb') as f:
        for p in params:
            pickle.dump(p, f)
    args = []
    with open(fn, 'rb') as f:
        while True:
            try:
                args.append(pickle.load(f))
            except EOFError:
                break
    assert len(args) == 14 and args[0] == [1, 2, 3] and args[12] == 4


def test_host_box_copy_and_zero_match_numpy():
    """sd_host_box_copy / sd_host_zero (threaded strided box copies of the chunk pipeline) against numpy slicing, incl.
    views into larger volumes on both sides, 1-voxel boxes and a box large enough to be split over threads."""
    import torch
    from syconn_amd._lib import host_box_copy, host_zero
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 255, (70, 90, 130), dtype=np.uint8)
    for (z, y, x), (dz, dy, dx) in (((0, 0, 0), (70, 90, 130)), ((3, 5, 7), (40, 33, 120)), ((69, 89, 129), (1, 1, 1)),
                                    ((10, 0, 2), (1, 90, 1))):
        dst = np.full((80, 100, 140), 7, np.uint8)
        want = dst.copy()
        want[2:2 + dz, 4:4 + dy, 6:6 + dx] = vol[z:z + dz, y:y + dy, x:x + dx]
        host_box_copy(dst[2:2 + dz, 4:4 + dy, 6:6 + dx], vol[z:z + dz, y:y + dy, x:x + dx], n_threads=5)
        assert np.array_equal(dst, want)
    t = torch.from_numpy(rng.integers(1, 255, (3_000_001,), dtype=np.uint8))
    host_zero(t, n_threads=7)
    assert int(t.sum()) == 0


def test_random_state_dict_loads_into_oracle_and_plans_identically():
    """syconn_amd.cnn (product-side architecture tables + seeded weights) names every parameter like elektronn3 does: the
    state_dict loads into the oracle UNet unchanged and gives the same plan as the nn.Module path."""
    import torch
    from oracle.unet_ref import ARCHS as ORACLE_ARCHS, UNet
    from syconn_amd.cnn import ARCHS as SPEC_ARCHS, random_state_dict
    from syconn_amd.plan import plan_from_model
    assert SPEC_ARCHS == ORACLE_ARCHS
    for arch in ('myelin', 'syntype', 'mivcsj'):
        sd = random_state_dict(arch, seed=5, final_scale=3.0)
        m = UNet(in_channels=1, **ORACLE_ARCHS[arch]).eval()
        assert not any(m.load_state_dict(sd))
        ops_a, blob_a, info_a = plan_from_model(sd)
        ops_b, blob_b, info_b = plan_from_model(m)
        assert len(ops_a) == len(ops_b) and info_a == info_b and np.array_equal(blob_a, blob_b)
        assert all(bytes(a) == bytes(b) for a, b in zip(ops_a, ops_b))


# ---- round 4 -----------------------------------------------------------------------------------------------------------
def test_config_defaults_select_reference_precision_and_tile_skipping():
    """`dense_prediction.act_dtype` defaults to the reference-precision plan ('f16x2' = what the reference's float16=False
    means, prediction.py:777-779); the fast plans are an explicit choice; tiles beyond the dataset are skipped by default."""
    from syconn_amd.handler.config import DEFAULTS
    from syconn_amd.handler.prediction import _ACT_NAMES, _FALLBACK
    assert DEFAULTS['dense_prediction']['act_dtype'] == 'f16x2'
    assert DEFAULTS['dense_prediction']['skip_tiles_outside_dataset'] is True
    assert _FALLBACK == {'f16': 'bf16', 'f16x2': 'f32'} and _ACT_NAMES['split'] == 'f16x2'


def test_worker_script_reads_a_stream_of_pickles(tmp_path):
    """batchjob_predict_dense.py's input contract (batchjob_utils.py:227-232): back-to-back pickles, one per tuple element."""
    import pickle
    from syconn_amd.batchjob_scripts.batchjob_predict_dense import read_pickle_stream
    items = [[1, 2, 3], 'kd_path', None, np.arange(4), (1.5, 'x')]
    f = tmp_path / 'job.pkl'
    with open(f, 'wb') as fh:
        for it in items:
            pickle.dump(it, fh)
    got = read_pickle_stream(str(f))
    assert len(got) == len(items) and got[0] == [1, 2, 3] and got[2] is None and np.array_equal(got[3], np.arange(4))
    (tmp_path / 'empty.pkl').write_bytes(b'')
    assert read_pickle_stream(str(tmp_path / 'empty.pkl')) == []


def test_object_segmentation_checks_the_sigma_count_and_the_gaussian_restatement_equals_scipy():
    """`sigmas` (object_extraction_steps.py:77-81, 296-298): a wrong count raises like the reference (:137-139); the oracle's
    restatement of vigra's filter equals scipy's ``gaussian_filter(mode='mirror', truncate=3)`` wherever both use the same
    window (sigma >= 1/6: scipy shrinks the window to one tap below that, vigra keeps radius 1)."""
    from scipy.ndimage import gaussian_filter
    from oracle.objseg_ref import gaussian_kernel_ref, gaussian_smoothing_ref
    from syconn_amd.extraction.object_extraction_steps import object_segmentation

    class _CS:
        chunk_dict = {}
    kw = dict(morph_ops={}, min_seed_vx={}, scaling=(10, 10, 20), overlap=(1, 1, 1))
    with pytest.raises(Exception, match='does not match'):
        object_segmentation(_CS(), ['mi', 'vc'], {}, [0.5, 0.5], sigmas=[[0, 0, 0]], **kw)
    rows, _, props = object_segmentation(_CS(), [], {}, [], sigmas=[], **kw)          # nothing to do: no chunk, no dataset
    assert rows == [] and props == {}
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 256, (9, 14, 23), dtype=np.uint8)
    for sigma in (1.0, (1.5, 0.7, 2.2), (0.0, 1.0, 0.0)):
        a = gaussian_smoothing_ref(vol, sigma)
        b = gaussian_filter(vol.astype(np.float32), sigma, mode='mirror', truncate=3.0)
        assert a.dtype == np.float32 and np.allclose(a, b, rtol=0, atol=2e-4), float(np.abs(a - b).max())
    k = gaussian_kernel_ref(0.1)
    assert len(k) == 3 and abs(k.sum() - 1) < 1e-15 and k[1] > 0.999999
    assert len(gaussian_kernel_ref(2.0)) == 13


def test_load_raw_into_a_caller_buffer_and_recycled_write_combining(tmp_path):
    """`load_raw(out=)` fills a caller-owned (page-locked in dense_predictor) buffer: zeros beyond the boundary and in missing
    cubes even when the buffer held data; the write-combining cache recycles its cube buffers without leaking voxels of one cube
    into the next."""
    from syconn_amd.knossos import KnossosDataset
    kd = KnossosDataset()
    kd._cube_shape = (32, 32, 32)
    kd.initialize_without_conf(str(tmp_path / 'kd'), boundary=(70, 50, 40), scale=(1, 1, 1), experiment_name='t', mags=[1])
    rng = np.random.default_rng(3)
    vol = rng.integers(1, 256, (40, 50, 70), dtype=np.uint8)                      # z, y, x; no zeros inside
    kd.enable_write_combining(max_cubes=4)
    for z0 in (0, 20):                                                             # two z-slabs: cubes complete only at flush
        for x0 in (0, 35):
            kd.save_raw(offset=(x0, 0, z0), mags=[1], data=vol[z0:z0 + 20, :, x0:x0 + 35], data_mag=1)
    kd.flush()
    want = kd.load_raw(size=(90, 60, 48), offset=(-10, -5, -4), mag=1)
    buf = np.full((48, 60, 90), 7, dtype=np.uint8)
    got = kd.load_raw(size=(90, 60, 48), offset=(-10, -5, -4), mag=1, out=buf)
    assert got is buf and np.array_equal(buf, want)
    assert np.array_equal(buf[4:44, 5:55, 10:80], vol) and buf[:4].max() == 0 and buf[:, :, 80:].max() == 0
    with pytest.raises(ValueError):
        kd.load_raw(size=(90, 60, 48), offset=(0, 0, 0), mag=1, out=np.zeros((48, 60, 91), np.uint8))


def test_predict_volume_distributed_hands_the_valid_box_to_predict_fn():
    """A `predict_fn` with a `valid_box` keyword receives, per chunk, the part of the chunk proper that lies inside the volume in
    chunk + halo coordinates (what Predictor uses to skip tiles beyond the dataset); one without the keyword is called as before."""
    from syconn_amd import parallel as par
    vol_shape, chunk, halo = (10, 20, 30), (8, 16, 16), (1, 2, 3)
    vol = torch.arange(int(np.prod(vol_shape)), dtype=torch.int64).remainder(251).to(torch.uint8).reshape(vol_shape)
    seen = []

    def fn(ch, valid_box=None):
        seen.append(valid_box)
        return ch[None, halo[0]:-halo[0], halo[1]:-halo[1], halo[2]:-halo[2]].contiguous()

    out = par.predict_volume_distributed(vol, vol_shape, chunk, halo, fn, n_out=1)
    assert torch.equal(out[0], vol)
    assert seen == [((1, 2, 3), (9, 18, 19)), ((1, 2, 3), (9, 18, 17)), ((1, 2, 3), (9, 6, 19)), ((1, 2, 3), (9, 6, 17)),
                    ((1, 2, 3), (3, 18, 19)), ((1, 2, 3), (3, 18, 17)), ((1, 2, 3), (3, 6, 19)), ((1, 2, 3), (3, 6, 17))]
    out2 = par.predict_volume_distributed(vol, vol_shape, chunk, halo, lambda ch: fn(ch), n_out=1)
    assert torch.equal(out2[0], vol)


def test_label_split_reports_a_posteriori_margins():
    """oracle/label_margin.py: one tolerance per storage type (no per-architecture exception) and the unsafe fractions at twice
    the MEASURED error next to the a-priori ones."""
    from oracle.label_margin import TOL_LOGIT_REL, label_split, merge_splits, stated_tolerance
    # a-priori rule: constants of the 17-rounding BatchNorm nets, scaled by sqrt(stored roundings / 17) -- mivcsj: 5 levels, GroupNorm
    from oracle.label_margin import stored_roundings
    assert stored_roundings('semseg_spine') == stored_roundings('semseg_axon') == stored_roundings('syntype') == 17
    assert stored_roundings('mivcsj') == 44
    assert stated_tolerance('semseg_spine', 'f16') == TOL_LOGIT_REL['f16'] and stated_tolerance('semseg_axon', 'bf16') == TOL_LOGIT_REL['bf16']
    assert abs(stated_tolerance('mivcsj', 'f16') - 1.3e-3 * (44 / 17) ** 0.5) < 1e-12 and 2.0e-3 < stated_tolerance('mivcsj', 'f16') < 2.2e-3
    assert stated_tolerance('mivcsj', 'f16x2') == TOL_LOGIT_REL['f16x2'] == 1e-5
    g = torch.Generator().manual_seed(0)
    ref = torch.randn((3, 8, 9, 10), generator=g) * 3
    got = ref + torch.randn(ref.shape, generator=g) * 1e-3
    from oracle.predictor_ref import label_rule_ref
    lab = torch.from_numpy(label_rule_ref((got.softmax(0).numpy() * 255).astype(np.uint8), [1, 2], [None] * 3)[0].astype(np.uint8))
    r = label_split(ref, got, got.softmax(0), lab, [1, 2], [None] * 3, 1e-2)
    assert 0.0 <= r['label_unsafe_frac_2x_measured_err'] <= r['label_unsafe_frac'] <= 1.0
    assert r['argmax_mismatch_safe'] == 0 and r['label_mismatch_safe'] == 0
    m = merge_splits([r, r])
    assert m['voxels'] == 2 * r['voxels'] and abs(m['label_unsafe_frac_2x_measured_err'] - r['label_unsafe_frac_2x_measured_err']) < 1e-12


def _c_clip_window(ops, lo, hi, full, axis, multiple):
    import ctypes as C
    from syconn_amd import _lib as L
    arr = (L.OpDesc * len(ops))(*ops)
    start, extent = C.c_int32(), C.c_int32()
    L.check(L.load().sd_plan_clip_window(arr, len(ops), axis, lo, hi, full, multiple, C.byref(start), C.byref(extent)))
    return int(start.value), int(extent.value)


@pytest.mark.parametrize('arch,axis', [('myelin', 0), ('myelin', 2), ('syntype', 0), ('syntype', 1), ('mivcsj', 2)])
def test_clipped_windows_keep_the_wanted_outputs_of_the_oracle_unet(arch, axis):
    """`sd_plan_clip_window` (model tiles of which only a part is wanted run on a clipped window): the oracle U-Net's outputs
    lo <= index < hi are the same on the clipped and the full window -- odd and even extents, also when the input BEYOND the
    window is not zero (the cone argument does not use that) -- and a window two voxels past the wanted ones is not enough
    somewhere; GroupNorm nets are never clipped."""
    from syconn_amd.plan import plan_from_model
    net = build_unet(arch, seed=3, start_filts=8 if arch == 'mivcsj' else 4)
    ops, _, _ = plan_from_model(net)
    full_shape = [21, 45, 45]
    full_shape[axis] = 91
    gen = torch.Generator().manual_seed(7)
    x = torch.rand((1, 1, *full_shape), generator=gen)
    with torch.no_grad():
        ref = net(x)
    tol = 2e-6 * float(ref.abs().max())
    too_small_differs = False
    for need in (9, 20, 33):
        s0, e = _c_clip_window(ops, 0, need, 91, axis, 1)
        assert s0 == 0
        if arch == 'mivcsj':
            assert e == 91
            continue
        assert need < e <= need + 64
        for ext in (e, e + 1, _c_clip_window(ops, 0, need, 91, axis, 8)[1]):
            with torch.no_grad():
                got = net(x.narrow(2 + axis, 0, ext))
            a, b = ref.narrow(2 + axis, 0, need), got.narrow(2 + axis, 0, need)
            assert torch.allclose(a, b, rtol=0, atol=tol), (need, ext, float((a - b).abs().max()))
        with torch.no_grad():
            got = net(x.narrow(2 + axis, 0, need + 2))
        too_small_differs |= not torch.allclose(ref.narrow(2 + axis, 0, need), got.narrow(2 + axis, 0, need), atol=1e-4)
    assert too_small_differs or arch == 'mivcsj'
    # near side: outputs lo <= index < hi on the window [start, start + extent)
    for lo, hi in ((60, 80), (48, 91), (70, 75), (3, 30)):
        start, ext = _c_clip_window(ops, lo, hi, 91, axis, 8)
        if arch == 'mivcsj':
            assert (start, ext) == (0, 91)
            continue
        assert start + ext <= 91 and start <= lo and start % 8 == 0 and (start > 0 or lo < 60)
        with torch.no_grad():
            got = net(x.narrow(2 + axis, start, ext))
        a, b = ref.narrow(2 + axis, lo, hi - lo), got.narrow(2 + axis, lo - start, hi - lo)
        assert torch.allclose(a, b, rtol=0, atol=tol), (lo, hi, start, ext, float((a - b).abs().max()))


def test_clip_window_in_c_equals_its_python_restatement_on_every_architecture():
    """`sd_plan_clip_window` against oracle/clip_ref.py: all 8 architectures + the sequential config-1 net, every axis, random
    wanted ranges, window sizes and multiples; argument errors are SD_ERR_INVALID -> ValueError."""
    from oracle.clip_ref import clipped_window
    from syconn_amd.plan import plan_from_model
    rng = np.random.default_rng(5)
    plans = [plan_from_model(build_unet(a, seed=1, start_filts=8))[0] for a in ARCHS] + [plan_from_model(build_cnn3(0))[0]]
    smaller = 0
    for ops in plans:
        for axis in range(3):
            for _ in range(40):
                full = int(rng.integers(8, 400))
                lo = int(rng.integers(0, full))
                hi = int(rng.integers(lo + 1, full + 1))
                mult = int(rng.choice([1, 2, 8, 16, 32]))
                got = _c_clip_window(ops, lo, hi, full, axis, mult)
                assert got == tuple(int(v) for v in clipped_window(ops, lo, hi, full, axis, mult)), (axis, lo, hi, full, mult)
                assert 0 <= got[0] <= lo and got[0] + got[1] <= full and hi <= got[0] + got[1]
                smaller += got[1] < full
    assert smaller > 300
    with pytest.raises(ValueError):
        _c_clip_window(plans[0], 5, 3, 10, 0, 8)
    with pytest.raises(ValueError):
        _c_clip_window(plans[0], 0, 3, 10, 3, 8)


def test_models_with_other_activations_are_refused():
    """A state_dict does not show the activation function: a traced or eager model that is not ReLU / transposed-conv (everything
    SyConn's cnn_*.py build is) raises ValueError instead of being predicted as if it were."""
    import torch.nn as nn
    from syconn_amd.plan import plan_from_model
    net = build_unet('myelin', seed=0, start_filts=4)
    plan_from_model(net)
    traced = torch.jit.trace(net, torch.randn(1, 1, 8, 16, 16))
    plan_from_model(traced)

    def swap(m):
        for name, ch in m.named_children():
            if isinstance(ch, nn.ReLU):
                setattr(m, name, nn.LeakyReLU(0.1))
            else:
                swap(ch)
    leaky = build_unet('myelin', seed=0, start_filts=4)
    swap(leaky)
    if any(isinstance(m, nn.LeakyReLU) for m in leaky.modules()):
        with pytest.raises(ValueError, match='LeakyReLU'):
            plan_from_model(leaky)
        with pytest.raises(ValueError, match='leaky_relu'):
            plan_from_model(torch.jit.trace(leaky, torch.randn(1, 1, 8, 16, 16)))
    seq = nn.Sequential(nn.Conv3d(1, 4, 3, padding=1), nn.Tanh(), nn.Conv3d(4, 2, 1))
    with pytest.raises(ValueError, match='Tanh'):
        plan_from_model(seq)
