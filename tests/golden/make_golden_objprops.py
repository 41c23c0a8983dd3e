"""Known-answer vectors for the label-volume statistics.

The reference's natives (/root/reference/syconn/extraction/find_object_properties_C.pyx) cannot be built in this image
(Cython 3.2.9 rejects the file), so nothing can be executed.  What the reference DOES hold is a known-answer test:
/root/reference/tests/test_segmentation_analysis.py:19-52 -- a 2x2x2 uint64 sample volume whose expected voxel counts,
bounding boxes and representative-coordinate property are derived from ``np.unique`` / ``np.where``.  This script stores
that sample volume (its literal values are data of that test) together with expectations derived the same way, plus a
second, larger case built by the recipe of /root/reference/tests/test_mapobjects_dense.py:14-27 (random uint64 ids in
[0, 1000) on a 50^3 cube -- that test holds no expected values; here they come from numpy as in the first test).

    python tests/golden/make_golden_objprops.py      ->  tests/golden/g8_objprops.npz
"""
import os

import numpy as np


def expectations(vol):
    ids, counts = np.unique(vol, return_counts=True)
    counts, ids = counts[ids != 0], ids[ids != 0]
    lo = np.zeros((len(ids), 3), np.int64)
    hi = np.zeros((len(ids), 3), np.int64)
    for i, e in enumerate(ids):
        w = np.transpose(np.where(vol == e))
        lo[i] = w.min(axis=0)
        hi[i] = w.max(axis=0) + 1
    return ids, counts, lo, hi


def main():
    sample = np.array([[[0, 1], [1, 1]], [[5, 2], [2, 1]]], np.uint64)      # test_segmentation_analysis.py:20-25
    rng = np.random.default_rng(50)
    toy = rng.integers(0, 1000, 50 ** 3).reshape((50, 50, 50)).astype(np.uint64)   # test_mapobjects_dense.py:16-18
    out = {}
    for name, vol in (('sample', sample), ('toy', toy)):
        ids, counts, lo, hi = expectations(vol)
        out.update({f'{name}_vol': vol, f'{name}_ids': ids, f'{name}_counts': counts, f'{name}_bb_lo': lo, f'{name}_bb_hi': hi})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g8_objprops.npz'), **out)


if __name__ == '__main__':
    main()
