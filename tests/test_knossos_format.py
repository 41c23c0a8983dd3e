"""CPU: the KNOSSOS on-disk formats of syconn_amd.knossos: raw uint8 cubes,
snappy-in-zip overlay cubes (``*.seg.sz.zip``) and the order-0 mag pyramid of save_raw / save_seg."""
import os
import sys
import zipfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.snappy_ref import decompress_ref  # noqa: E402
from syconn_amd.knossos import KnossosDataset  # noqa: E402


def _kd(tmp_path, boundary=(300, 200, 150), mags=(1, 2, 4)):
    kd = KnossosDataset()
    kd.initialize_without_conf(str(tmp_path / 'kd'), boundary, (10., 10., 25.), 'exp', mags=list(mags))
    return kd


def test_seg_cubes_are_snappy_in_zip(tmp_path):
    kd = _kd(tmp_path)
    rng = np.random.default_rng(0)
    seg = np.zeros((150, 200, 300), np.uint64)                      # z, y, x
    seg[10:140, 20:180, 30:280] = 2 ** 40 + 7
    seg[rng.integers(0, 150, 200), rng.integers(0, 200, 200), rng.integers(0, 300, 200)] = rng.integers(1, 2 ** 62, 200)
    kd.save_seg(offset=(0, 0, 0), mags=[1], data=seg, data_mag=1)
    fn = tmp_path / 'kd' / 'mag1' / 'x0001' / 'y0001' / 'z0000' / 'exp_mag1_x0001_y0001_z0000.seg.sz.zip'
    assert fn.is_file()
    with zipfile.ZipFile(fn) as zf:
        assert zf.namelist() == ['exp_mag1_x0001_y0001_z0000.seg.sz']
        blob = zf.read('exp_mag1_x0001_y0001_z0000.seg.sz')
    cube = np.frombuffer(decompress_ref(blob), '<u8').reshape(128, 128, 128)     # independent decoder
    want = np.zeros((128, 128, 128), np.uint64)
    want[:, :72, :128] = seg[0:128, 128:200, 128:256]
    assert np.array_equal(cube, want)
    # reading back through the API, incl. a box that straddles cubes and leaves the dataset
    assert np.array_equal(kd.load_seg(size=(300, 200, 150), offset=(0, 0, 0), mag=1), seg)
    box = kd.load_seg(size=(100, 90, 60), offset=(250, 150, 120), mag=1)
    want = np.zeros((60, 90, 100), np.uint64)
    want[:30, :50, :50] = seg[120:150, 150:200, 250:300]
    assert np.array_equal(box, want)


def test_partial_update_of_an_existing_seg_cube(tmp_path):
    kd = _kd(tmp_path)
    a = np.full((20, 30, 40), 5, np.uint64)
    kd.save_seg(offset=(10, 10, 10), mags=[1], data=a, data_mag=1)
    b = np.full((10, 10, 10), 9, np.uint64)
    kd.save_seg(offset=(45, 35, 25), mags=[1], data=b, data_mag=1)          # overlaps the first box at its corner
    got = kd.load_seg(size=(64, 64, 64), offset=(0, 0, 0), mag=1)
    want = np.zeros((64, 64, 64), np.uint64)
    want[10:30, 10:40, 10:50] = 5
    want[25:35, 35:45, 45:55] = 9
    assert np.array_equal(got, want)


def test_mag_pyramid_levels_written_one_by_one_equal_one_call(tmp_path):
    """dense_predictor writes each pyramid level with its own data_mag (levels come from the device kernel); the
    files must be identical to one save_raw(mags=[1,2,4]) call on the mag-1 data."""
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, (90, 130, 170), dtype=np.uint8)
    kd1 = _kd(tmp_path / 'a')
    kd1.save_raw(offset=(128, 0, 0), mags=[1, 2, 4], data=data, data_mag=1, fast_resampling=True, upsample=False)
    kd2 = _kd(tmp_path / 'b')
    for k in range(3):
        r = 2 ** k
        kd2.save_raw(offset=(128, 0, 0), mags=[r], data=np.ascontiguousarray(data[::r, ::r, ::r]), data_mag=r,
                     fast_resampling=True, upsample=False)
    for mag in (1, 2, 4):
        size = (300, 200, 150)
        x = kd1.load_raw(size=size, offset=(0, 0, 0), mag=mag)
        y = kd2.load_raw(size=size, offset=(0, 0, 0), mag=mag)
        assert x.shape == tuple(s // mag for s in size[::-1]) and np.array_equal(x, y)
        assert np.array_equal(x[:-(-90 // mag), :-(-130 // mag), 128 // mag:128 // mag - (-170 // mag)],
                              data[::mag, ::mag, ::mag])


def test_pyknossos_conf_is_parsed_or_refused(tmp_path):
    """kd_factory prefers a *.pyk.conf (basics.py:58-60): it must open the dataset it describes, or raise -- never an empty dataset."""
    from syconn_amd.handler.basics import kd_factory
    from syconn_amd.knossos import KnossosDataset
    root = tmp_path / 'kd'
    kd = KnossosDataset()
    kd.initialize_without_conf(str(root), boundary=(200, 150, 70), scale=(10., 10., 25.), experiment_name='pyk', mags=[1, 2, 4],
                               create_pyk_conf=True, create_knossos_conf=False)
    raw = np.random.default_rng(0).integers(0, 256, (70, 150, 200), dtype=np.uint8)
    kd.save_raw(offset=(0, 0, 0), mags=[1, 2, 4], data=raw, data_mag=1, fast_resampling=True)
    k2 = kd_factory(str(root))
    assert tuple(k2.boundary) == (200, 150, 70) and k2.experiment_name == 'pyk' and list(k2.available_mags) == [1, 2, 4]
    assert np.allclose(k2.scale, (10., 10., 25.))
    assert np.array_equal(k2.load_raw(size=(200, 150, 70), offset=(0, 0, 0), mag=1), raw)
    assert np.array_equal(k2.load_raw(size=(50, 40, 10), offset=(8, 4, 4), mag=2), raw[::2, ::2, ::2][2:7, 2:22, 4:29])      # (size / offset in mag-1 voxels)
    bad = tmp_path / 'bad.pyk.conf'
    bad.write_text('[Dataset]\n_BaseName = x\n')
    with pytest.raises(ValueError):
        KnossosDataset().initialize_from_pyknossos_path(str(bad))
    bad.write_text('experiment name "x";\n')
    with pytest.raises(ValueError):
        KnossosDataset().initialize_from_conf(str(bad))
    with pytest.raises(ValueError):
        k2.load_raw(size=(8, 8, 8), offset=(0, 0, 0), mag=1, out=np.empty((8, 8, 9), np.uint8))
