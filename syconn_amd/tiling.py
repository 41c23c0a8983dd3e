"""Tile geometry of ``tiled_apply`` (elektronn3) as the device path runs it -- which model tiles of a volume are
predicted and on which window -- as pure host arithmetic, shared by ``Predictor._tiled`` (what runs) and by the chunk cost model
of ``parallel.predict_volume_distributed`` (what the rounds are ordered by).  Nothing here touches a GPU:
``sd_plan_clip_window`` is host arithmetic on the plan (include/syconn_dense.h)."""
import ctypes as C
import itertools
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


class PlanClipper:
    """``sd_plan_clip_window`` for one plan (a list of ``OpDesc``), cached: (start, extent) of the part of an input window of `full`
    voxels along `axis` on which the outputs lo <= index < hi are what they are on the whole window; extents in multiples of 8 so
    that the poolings of the clipped window stay whole."""

    def __init__(self, ops: Sequence):
        self.lib = L.load()
        self.n_ops = len(ops)
        self._arr = ops if isinstance(ops, C.Array) else (L.OpDesc * len(ops))(*ops)
        self.has_groupnorm = any(int(o.kind) == L.SD_OP_GROUPNORM for o in self._arr)
        self._cache: Dict[tuple, Tuple[int, int]] = {}

    @classmethod
    def for_model(cls, model, group_norm_groups: Optional[int] = None) -> 'PlanClipper':
        from .plan import plan_from_model
        return cls(plan_from_model(model, group_norm_groups)[0])

    def __call__(self, lo: int, hi: int, full: int, axis: int) -> Tuple[int, int]:
        key = (lo, hi, full, axis)
        w = self._cache.get(key)
        if w is None:
            start, extent = C.c_int32(), C.c_int32()
            L.check(self.lib.sd_plan_clip_window(self._arr, self.n_ops, axis, lo, hi, full, 8, C.byref(start), C.byref(extent)),
                    'sd_plan_clip_window')
            w = self._cache[key] = (int(start.value), int(extent.value))
        return w


def tile_grid(spatial: np.ndarray, tile_shape, overlap_shape, strict_shapes: bool = False):
    """(tile, overlap, tiles per axis) for a volume of `spatial` voxels: the tile shrinks to the volume where it is larger
    (elektronn3 pads nothing in that case), ``strict_shapes`` demands divisibility."""
    tile = spatial.copy() if tile_shape is None else np.asarray(tile_shape, dtype=np.int64)
    ol = np.zeros(3, dtype=np.int64) if overlap_shape is None else np.asarray(overlap_shape, dtype=np.int64)
    if len(tile) != 3 or len(ol) != 3:
        raise ValueError('tile_shape / overlap_shape must have 3 entries (z, y, x)')
    if np.any(spatial % tile != 0):
        if strict_shapes:
            raise ValueError(f'spatial inp shape {tuple(spatial)} has to be divisible by '
                             f'tile_shape {tuple(tile)} (strict_shapes=True)')
        if np.any(tile > spatial):
            tile = np.minimum(tile, spatial)
    return tile, ol, np.ceil(spatial / tile).astype(np.int64)


def plan_tile_windows(spatial: np.ndarray, tile: np.ndarray, ol: np.ndarray, ntiles: np.ndarray, valid_box, clip: bool,
                      clipper: Optional[Callable], whole_tile_statistics: bool):
    """Which tiles are predicted, on which window, keeping which box.

    -> (by_window, zero): ``by_window[(window shape, roi)] = [(lo, start, w0, w1), ...]`` -- tile origin `lo` in the output, window start
    `start` inside the full tile window, wanted part ``[w0, w1)`` of the tile's core; tiles of one key share launch sets.  `zero`:
    part of the output is not written by any tile (skipped tiles, clipped cores) and has to read zero.
    `valid_box` = ((z0, y0, x0), (z1, y1, x1)) in output coordinates: tiles whose core lies entirely outside are skipped.  `clip`:
    windows end where the cones of the wanted voxels end (`clipper`); networks with whole-tile statistics (GroupNorm) keep full
    windows and whole cores' boxes."""
    tin = tile + 2 * ol
    pos_list = list(itertools.product(*[range(int(n)) for n in ntiles]))   # z-major, like upstream
    zero = False
    if valid_box is not None:
        v_lo, v_hi = (np.asarray(v, dtype=np.int64) for v in valid_box)
        inside = [pos for pos in pos_list
                  if np.all(tile * np.asarray(pos) < v_hi) and np.all(np.minimum(tile * (np.asarray(pos) + 1), spatial) > v_lo)]
        zero = len(inside) < len(pos_list)
        pos_list = inside
    by_window: Dict[tuple, List[tuple]] = {}
    for pos in pos_list:
        lo = tile * np.asarray(pos, dtype=np.int64)
        keep = np.minimum(tile, spatial - lo)
        w0 = np.zeros(3, dtype=np.int64) if valid_box is None else np.maximum(v_lo - lo, 0)      # wanted: [w0, w1) from lo
        w1 = keep if valid_box is None else np.minimum(keep, v_hi - lo)
        win = [clipper(int(ol[a] + w0[a]), int(ol[a] + w1[a]), int(tin[a]), a) for a in range(3)] if clip else [(0, int(t)) for t in tin]
        zero = zero or bool(np.any(w1 < keep) or np.any(w0 > 0))
        start = np.asarray([w[0] for w in win], dtype=np.int64)
        # the box of the window that is scattered (tiled_apply keeps the core of a tile): the decoder computes only what it
        # depends on (`sd_model_set_roi`); tiles of one window that keep the same box share a launch set
        roi = (tuple(int(v) for v in ol + w0 - start), tuple(int(v) for v in ol + w1 - start)) if clip else None
        if roi is not None and (whole_tile_statistics or np.prod(np.subtract(roi[1], roi[0])) > 0.8 * np.prod([w[1] for w in win])):
            roi = None                       # (nearly the whole window is kept: whole-tile kernels -- the fused level-0 decoder -- win;
                                             # GroupNorm networks: the library ignores the box, it would only split launch sets)
        by_window.setdefault((tuple(w[1] for w in win), roi), []).append((lo, start, w0, w1))
    return by_window, zero


def convolved_voxels(by_window) -> int:
    """Input voxels the network runs over for a plan of `plan_tile_windows`: the cost proxy of a chunk (every layer's work scales
    with the window volume)."""
    return int(sum(len(tiles) * int(np.prod(win)) for (win, _), tiles in by_window.items()))


class ChunkCostModel:
    """Cost of predicting one chunk of a chunked volume prediction, from geometry alone (before anything runs): the voxels of all
    windows its predicted tiles are run on.  Mirrors how ``bench.py`` / ``dense_predictor`` drive the ``Predictor``:

    * ``halo_included=False`` (reference geometry): the chunk + halo box is the Predictor's volume, zero-padded by the overlap and
      tiled (prediction.py:775-781); the halo ring is cropped afterwards (:812), so the wanted box is the chunk proper inside the
      dataset;
    * ``halo_included=True`` (tile128 geometry): the halo is real neighbouring data, the tile grid covers the chunk proper."""

    def __init__(self, clipper: Optional[PlanClipper], tile_shape, overlap_shape, halo, halo_included: bool, clip_tiles: bool = True,
                 skip_outside: bool = True):
        self.clipper, self.tile_shape, self.overlap_shape = clipper, tile_shape, overlap_shape
        self.halo = np.asarray(halo, dtype=np.int64)
        self.halo_included, self.skip_outside = bool(halo_included), bool(skip_outside)
        self.stats = bool(clipper is not None and clipper.has_groupnorm)
        self.clip = bool(clip_tiles and clipper is not None and not self.stats)

    def chunk_cost(self, chunk_shape, valid_box) -> int:
        """`valid_box`: the part of the chunk proper inside the dataset, in chunk + halo coordinates (what
        ``predict_volume_distributed`` hands to its ``predict_fn``)."""
        cs = np.asarray(chunk_shape, dtype=np.int64)
        spatial = cs if self.halo_included else cs + 2 * self.halo
        tile, ol, ntiles = tile_grid(spatial, self.tile_shape, self.overlap_shape)
        vb = None
        if valid_box is not None and self.skip_outside:
            vb = valid_box if not self.halo_included else (tuple(np.asarray(valid_box[0]) - self.halo), tuple(np.asarray(valid_box[1]) - self.halo))
        bw, _ = plan_tile_windows(spatial, tile, ol, ntiles, vb, self.clip, self.clipper, self.stats)
        return convolved_voxels(bw)
