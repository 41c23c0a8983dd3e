"""Generate tests/golden/g7_myelin2coords.npz (run ONCE in the build container, where /root/reference exists).

G7 pins ``map_myelin2coords`` against the REFERENCE'S OWN function: it is lifted from
/root/reference/syconn/reps/super_segmentation_helper.py:550-615 by AST and executed here; its three external names
are bound to stand-ins: ``global_params.config.working_dir`` -> a temp dir, ``kd_factory`` -> the in-repo
KnossosDataset (knossos_utils is not installed; its ``load_raw(size, offset, mag)`` contract is SURVEY.md row K),
``os`` -> os.  Only inputs and outputs are stored.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import lift  # noqa: E402
from syconn_amd.handler.basics import kd_factory  # noqa: E402
from syconn_amd.knossos import KnossosDataset  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    mag = 4
    shape4 = (40, 70, 90)                                   # z, y, x at mag 4
    from scipy.ndimage import gaussian_filter
    v = gaussian_filter(rng.random(shape4, dtype=np.float32), 3.0)
    vol4 = (255 * (v - v.min()) / (v.max() - v.min())).astype(np.uint8)      # blobs around the 127 threshold
    with tempfile.TemporaryDirectory() as wd:
        kd = KnossosDataset()
        bnd = np.array(shape4[::-1]) * mag
        kd.initialize_without_conf(wd + '/knossosdatasets/myelin/', bnd, (10., 10., 25.), 'myelin', mags=[1, mag])
        kd.save_raw(offset=(0, 0, 0), mags=[mag], data=vol4, data_mag=mag, fast_resampling=True, upsample=False)
        gp = types.SimpleNamespace(config=types.SimpleNamespace(working_dir=wd))
        (fn,) = lift('/root/reference/syconn/reps/super_segmentation_helper.py', ['map_myelin2coords'],
                     {'global_params': gp, 'kd_factory': kd_factory, 'os': os})
        coords = np.concatenate([
            rng.integers(0, bnd, (300, 3)),                                       # anywhere inside
            rng.integers(-30, 30, (40, 3)),                                       # around the origin corner (boxes leave the volume)
            bnd - rng.integers(-30, 30, (40, 3)),                                 # around the far corner
            np.array([[0, 0, 0], bnd - 1, bnd // 2, [-100, -100, -100]]),
        ]).astype(np.int64)
        out = {'vol4': vol4, 'coords': coords, 'mag': np.int64(mag)}
        out['default'] = fn(coords, mag=mag)
        out['edge_5_7_3'] = fn(coords, cube_edge_avg=np.array([5, 7, 3]), mag=mag)
        out['thresh_100_maj_0p3'] = fn(coords, thresh_proba=100, thresh_majority=0.3, mag=mag)
        out['thresh_frac_maj_0p1'] = fn(coords, thresh_proba=140.5, thresh_majority=0.1, mag=mag)
    np.savez_compressed(f'{HERE}/g7_myelin2coords.npz', **out)
    print({k: (v.shape, int(v.sum())) for k, v in out.items() if k not in ('vol4', 'coords', 'mag')})


if __name__ == '__main__':
    main()
