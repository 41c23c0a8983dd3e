"""GPU: map_myelin2coords through the C ABI (sd_box_majority) against the golden outputs
of the reference's own function (tests/golden/g7_myelin2coords.npz) and the numpy oracle -- integer work, bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

from oracle.myelin_ref import box_majority_ref  # noqa: E402

pytestmark = pytest.mark.gpu


def _make_wd(tmp_path, vol4, mag):
    from syconn_amd import global_params
    from syconn_amd.handler.config import generate_default_conf
    from syconn_amd.knossos import KnossosDataset
    wd = str(tmp_path / 'wd')
    os.makedirs(wd, exist_ok=True)
    generate_default_conf(wd, scaling=(10., 10., 25.))
    kd = KnossosDataset()
    bnd = np.array(vol4.shape[::-1]) * mag
    kd.initialize_without_conf(wd + '/knossosdatasets/myelin/', bnd, (10., 10., 25.), 'myelin', mags=[1, mag])
    kd.save_raw(offset=(0, 0, 0), mags=[mag], data=vol4, data_mag=mag, fast_resampling=True, upsample=False)
    global_params.wd = wd
    return wd


def test_map_myelin2coords_matches_reference_golden(gpu, tmp_path):
    from syconn_amd import global_params
    from syconn_amd.reps import super_segmentation_helper as ssh
    g = np.load(os.path.join(GOLDEN, 'g7_myelin2coords.npz'))
    vol4, coords, mag = g['vol4'], g['coords'], int(g['mag'])
    _make_wd(tmp_path, vol4, mag)
    try:
        assert np.array_equal(ssh.map_myelin2coords(coords, mag=mag), g['default'])
        assert np.array_equal(ssh.map_myelin2coords(coords, cube_edge_avg=np.array([5, 7, 3]), mag=mag), g['edge_5_7_3'])
        assert np.array_equal(ssh.map_myelin2coords(coords, thresh_proba=100, thresh_majority=0.3, mag=mag),
                              g['thresh_100_maj_0p3'])
        assert np.array_equal(ssh.map_myelin2coords(coords, thresh_proba=140.5, thresh_majority=0.1, mag=mag),
                              g['thresh_frac_maj_0p1'])
        # spatial bucketing must not matter
        old = ssh._REGION_VOX
        ssh._REGION_VOX = 16
        assert np.array_equal(ssh.map_myelin2coords(coords, mag=mag), g['default'])
        ssh._REGION_VOX = old
        out = ssh.map_myelin2coords(np.zeros((0, 3), np.int64), mag=mag)
        assert out.shape == (0,) and out.dtype == np.uint8
        global_params.wd = str(tmp_path / 'nowhere')
        os.makedirs(global_params.wd, exist_ok=True)
        from syconn_amd.handler.config import generate_default_conf
        generate_default_conf(global_params.wd, scaling=(10., 10., 25.))
        with pytest.raises(ValueError):                    # no myelin KD (super_segmentation_helper.py:602-603)
            ssh.map_myelin2coords(coords, mag=mag)
    finally:
        global_params.wd = None


def test_box_majority_kernel_large_random(gpu):
    """The kernel alone at a larger size against the numpy oracle: 20k boxes incl. ones that leave the volume."""
    from syconn_amd.reps.super_segmentation_helper import box_majority_device
    rng = np.random.default_rng(5)
    vol = rng.integers(0, 256, (64, 96, 128), dtype=np.uint8)
    org_zyx = np.stack([rng.integers(-8, 70, 20000), rng.integers(-12, 100, 20000), rng.integers(-12, 130, 20000)], 1)
    for edge_zyx, tp, tm in (((5, 11, 11), 127, 0.5), ((1, 1, 1), 200.5, 0.0), ((7, 3, 9), 0, 0.99), ((4, 4, 4), 255, 0.0)):
        got = box_majority_device(torch.from_numpy(vol).to(gpu), torch.from_numpy(org_zyx.astype(np.int32)).to(gpu),
                                  edge_zyx, tp, tm).cpu().numpy()
        e = np.asarray(edge_zyx)
        # oracle works on coordinates: centre c with offset = c - edge//2 at mag 1 -> pass c = origin + edge//2
        ref = box_majority_ref(vol, (org_zyx + e // 2)[:, ::-1], cube_edge_avg=e[::-1], thresh_proba=tp,
                               thresh_majority=tm, mag=1)
        assert np.array_equal(got, ref), (edge_zyx, tp, tm)
