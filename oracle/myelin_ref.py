"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``map_myelin2coords``
(/root/reference/syconn/reps/super_segmentation_helper.py:550-615), the first consumer of the myelin probability
map the dense path writes.  PINNED: tests/golden/g7_myelin2coords.npz holds the outputs of
the reference's own function (lifted from the read-only source and executed by tests/golden/make_golden.py on the
same synthetic volume); tests/test_oracle_golden.py checks this restatement against them.

Only tests/ may import this module.
"""
import numpy as np


def box_majority_ref(vol_zyx: np.ndarray, coords_xyz: np.ndarray, cube_edge_avg=(11, 11, 5), thresh_proba=255 // 2,
                     thresh_majority=0.5, mag: int = 4) -> np.ndarray:
    """`vol_zyx`: the whole myelin volume at magnification `mag` (uint8; outside = 0 like kd.load_raw)."""
    edge = np.asarray(cube_edge_avg, dtype=np.int64)
    n_cube_vx = np.prod(edge)                                           # :606
    edge_m1 = edge * mag                                                # :608
    out = np.zeros(len(coords_xyz), dtype=np.uint8)
    D, H, W = vol_zyx.shape
    for i, c in enumerate(np.asarray(coords_xyz, dtype=np.int64)):
        offset = c - edge_m1 // 2                                       # :610
        o = np.floor_divide(offset, mag)                                # load_raw: offset // mag, size // mag
        box = np.zeros((edge[2], edge[1], edge[0]), dtype=np.uint8)
        z0, y0, x0 = int(o[2]), int(o[1]), int(o[0])
        zs, ys, xs = max(z0, 0), max(y0, 0), max(x0, 0)
        ze, ye, xe = min(z0 + edge[2], D), min(y0 + edge[1], H), min(x0 + edge[0], W)
        if ze > zs and ye > ys and xe > xs:
            box[zs - z0:ze - z0, ys - y0:ye - y0, xs - x0:xe - x0] = vol_zyx[zs:ze, ys:ye, xs:xe]
        ratio = np.sum(box > thresh_proba) / n_cube_vx                  # :612
        out[i] = ratio > thresh_majority                                # :613
    return out
