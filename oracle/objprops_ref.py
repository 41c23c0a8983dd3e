"""ORACLE (test infrastructure, NOT product code): CPU restatement of the reference's label-volume statistics, the only
native code of SyConn (Cython -> C++): /root/reference/syconn/extraction/find_object_properties_C.pyx

* ``find_object_properties``      :24-49    one raster scan (x outermost, z innermost) over a uint32/uint64 label volume,
                                            per non-zero id: representative coordinate = FIRST voxel met, bounding box
                                            [min, max + 1), voxel count;
* ``map_subcell_C``               :72-109   per subcellular volume: subcell id -> cell id -> number of voxels where both
                                            are non-zero;
* ``map_subcell_extract_props``   :112-192  both of the above in one scan (cell properties, properties of every
                                            subcellular volume, overlap counts).

Only ``tests/``, ``__graft_entry__.smoke()``, ``tools/`` probes and ``bench.py``'s ``cpu_baseline`` leg may import this.

PINNING: the .pyx cannot be built here (Cython 3.2.9 rejects its unused ``ctypedef vector[n_type[:, :, :]]`` at :16, and
the sources may not be altered), so there is no ``oracle/_ref`` for it.  The restatement is pinned by the reference's
own known-answer test /root/reference/tests/test_segmentation_analysis.py:19-52 (2x2x2 sample volume; expectations
derived from ``np.unique`` / ``np.where`` exactly as that test derives them), reproduced in
``tests/test_objprops.py`` against ``tests/golden/g8_objprops.npz``.

Two forms: ``*_loops`` are literal pure-Python restatements (small inputs only), ``*_np`` are vectorised numpy
equivalents (checked against the loops on random volumes) used for the larger parity cases and the CPU baseline.
Return types mirror what Cython hands back for the C++ maps: plain dicts with list values.
"""
from typing import Dict, List, Tuple

import numpy as np


# -- literal restatements ------------------------------------------------------------------------------------------------
def find_object_properties_loops(chunk: np.ndarray):
    """find_object_properties_C.pyx:24-49."""
    rep_coords: Dict[int, List[int]] = {}
    bounding_box: Dict[int, List[List[int]]] = {}
    sizes: Dict[int, int] = {}
    for x in range(chunk.shape[0]):
        for y in range(chunk.shape[1]):
            for z in range(chunk.shape[2]):
                key = int(chunk[x, y, z])
                if key == 0:
                    continue
                if key in sizes:
                    bb = bounding_box[key]
                    bb[0][0] = min(bb[0][0], x); bb[0][1] = min(bb[0][1], y); bb[0][2] = min(bb[0][2], z)
                    bb[1][0] = max(bb[1][0], x + 1); bb[1][1] = max(bb[1][1], y + 1); bb[1][2] = max(bb[1][2], z + 1)
                    sizes[key] += 1
                else:
                    bounding_box[key] = [[x, y, z], [x + 1, y + 1, z + 1]]
                    sizes[key] = 1
                    rep_coords[key] = [x, y, z]
    return rep_coords, bounding_box, sizes


def map_subcell_extract_props_loops(ch: np.ndarray, subcell_chs: np.ndarray):
    """find_object_properties_C.pyx:112-192.  `subcell_chs` is (n_subcell, X, Y, Z)."""
    n_sub = subcell_chs.shape[0]
    for ii in range(n_sub):
        assert subcell_chs[ii].shape == ch.shape, 'Segmentation of cells and subcellular structures must have same shape.'
    cell = ({}, {}, {})
    sub_rc = [dict() for _ in range(n_sub)]
    sub_bb = [dict() for _ in range(n_sub)]
    sub_sz = [dict() for _ in range(n_sub)]
    mapping = [dict() for _ in range(n_sub)]

    def update(rc, bb, sz, key, x, y, z):
        if key in bb:
            b = bb[key]
            b[0][0] = min(b[0][0], x); b[0][1] = min(b[0][1], y); b[0][2] = min(b[0][2], z)
            b[1][0] = max(b[1][0], x + 1); b[1][1] = max(b[1][1], y + 1); b[1][2] = max(b[1][2], z + 1)
            sz[key] += 1
        else:
            bb[key] = [[x, y, z], [x + 1, y + 1, z + 1]]
            sz[key] = 1
            rc[key] = [x, y, z]

    for x in range(ch.shape[0]):
        for y in range(ch.shape[1]):
            for z in range(ch.shape[2]):
                key = int(ch[x, y, z])
                for ii in range(n_sub):
                    sk = int(subcell_chs[ii, x, y, z])
                    if sk == 0:
                        continue
                    update(sub_rc[ii], sub_bb[ii], sub_sz[ii], sk, x, y, z)
                    if key != 0:
                        d = mapping[ii].setdefault(sk, {})
                        d[key] = d.get(key, 0) + 1
                if key == 0:
                    continue
                update(cell[0], cell[1], cell[2], key, x, y, z)
    return [cell[0], cell[1], cell[2]], [sub_rc, sub_bb, sub_sz], mapping


def map_subcell_loops(ch: np.ndarray, subcell_chs: np.ndarray):
    """find_object_properties_C.pyx:72-109: the overlap counts only."""
    return map_subcell_extract_props_loops(ch, subcell_chs)[2]


# -- vectorised equivalents ------------------------------------------------------------------------------------------------
def _props_np(vol: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(ids ascending, first raster index, count, bbox (n, 2, 3)) of the non-zero labels."""
    flat = np.ascontiguousarray(vol).reshape(-1)
    nz = np.flatnonzero(flat)
    if nz.size == 0:
        return (np.zeros(0, vol.dtype), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 2, 3), np.int64))
    order = np.argsort(flat[nz], kind='stable')
    idx = nz[order]                                   # raster indices grouped by id, ascending inside a group
    keys = flat[idx]
    starts = np.flatnonzero(np.concatenate(([True], keys[1:] != keys[:-1])))
    ids = keys[starts]
    counts = np.diff(np.concatenate((starts, [keys.size])))
    first = idx[starts]
    xyz = np.stack(np.unravel_index(idx, vol.shape), axis=1)
    lo = np.minimum.reduceat(xyz, starts, axis=0)
    hi = np.maximum.reduceat(xyz, starts, axis=0) + 1
    return ids, first, counts, np.stack((lo, hi), axis=1)


def _dicts(vol_shape, ids, first, counts, bb):
    rc = np.stack(np.unravel_index(first, vol_shape), axis=1) if len(ids) else np.zeros((0, 3), np.int64)
    keys = [int(i) for i in ids]
    return (dict(zip(keys, rc.tolist())), dict(zip(keys, bb.tolist())), dict(zip(keys, [int(c) for c in counts])))


def find_object_properties_np(chunk: np.ndarray):
    return _dicts(chunk.shape, *_props_np(chunk))


def map_subcell_extract_props_np(ch: np.ndarray, subcell_chs: np.ndarray):
    n_sub = subcell_chs.shape[0]
    cell = list(_dicts(ch.shape, *_props_np(ch)))
    sub = [[], [], []]
    mapping = []
    cflat = np.ascontiguousarray(ch).reshape(-1)
    for ii in range(n_sub):
        assert subcell_chs[ii].shape == ch.shape, 'Segmentation of cells and subcellular structures must have same shape.'
        rc, bb, sz = _dicts(ch.shape, *_props_np(subcell_chs[ii]))
        sub[0].append(rc); sub[1].append(bb); sub[2].append(sz)
        sflat = np.ascontiguousarray(subcell_chs[ii]).reshape(-1)
        both = np.flatnonzero((sflat != 0) & (cflat != 0))
        d: Dict[int, Dict[int, int]] = {}
        if both.size:
            pairs, cnt = np.unique(np.stack((sflat[both], cflat[both]), axis=1), axis=0, return_counts=True)
            for (s, c), n in zip(pairs.tolist(), cnt.tolist()):
                d.setdefault(int(s), {})[int(c)] = int(n)
        mapping.append(d)
    return cell, sub, mapping
