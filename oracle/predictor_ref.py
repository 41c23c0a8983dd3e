"""ORACLE (test infrastructure, NOT product code): CPU restatement of the tiling / post-processing
semantics around the U-Net on SyConn's dense prediction path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Two groups of functions:

* ``tiled_apply`` / :class:`PredictorRef` restate ``elektronn3.inference.inference`` (third-party, absent from
  /root/reference; pinned only by SyConn's call sites /root/reference/syconn/handler/prediction.py:775-781, 863,
  1055-1062) -> PARITY UNPINNED for these two.
* ``dense_predicton_helper_ref``, ``xyz2zyx_ref``, ``zyx2xyz_ref``, ``chunkify_ref``, ``label_rule_ref`` restate
  numpy code that IS in the reference tree; they are pinned by golden vectors produced from the reference's own
  functions (AST-lifted by tests/golden/make_golden.py; see tests/test_oracle_golden.py).
"""
import itertools
from typing import List, Optional, Sequence

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# elektronn3.inference.inference.tiled_apply (row P3)
def tiled_apply(func, inp: torch.Tensor, tile_shape, overlap_shape, out_shape) -> torch.Tensor:
    """Zero-pad `inp` (N,C,D,H,W) by `overlap_shape` per side, run `func` on every
    ``tile_shape + 2*overlap_shape`` input tile (z-major tile order), keep the central `tile_shape`
    region of each output tile, no blending.  `out_shape` = (N, C_out, D, H, W)."""
    tile_shape = np.asarray(tile_shape, dtype=np.int64)
    overlap_shape = np.asarray(overlap_shape, dtype=np.int64)
    spatial = np.asarray(inp.shape[2:], dtype=np.int64)
    padded_shape = spatial + 2 * overlap_shape
    inp_padded = torch.zeros((*inp.shape[:2], *padded_shape.tolist()), dtype=inp.dtype)
    inp_padded[(slice(None), slice(None)) + tuple(
        slice(int(l), int(h)) for l, h in zip(overlap_shape, padded_shape - overlap_shape))] = inp
    out = torch.empty(tuple(int(s) for s in out_shape), dtype=torch.float32)
    tiles = np.ceil(np.asarray(out_shape[2:]) / tile_shape).astype(np.int64)
    crop = (slice(None), slice(None)) + tuple(
        slice(int(o), int(o + t)) for o, t in zip(overlap_shape, tile_shape))
    for tile_pos in itertools.product(*[range(int(t)) for t in tiles]):
        tile_pos = np.asarray(tile_pos, dtype=np.int64)
        lo = tile_shape * tile_pos
        hi = tile_shape * (tile_pos + 1)
        inp_slice = (slice(None), slice(None)) + tuple(
            slice(int(l), int(h + 2 * o)) for l, h, o in zip(lo, hi, overlap_shape))
        out_slice = (slice(None), slice(None)) + tuple(slice(int(l), int(h)) for l, h in zip(lo, hi))
        out_tile = func(inp_padded[inp_slice])
        out[out_slice] = out_tile[crop]
    return out


class PredictorRef:
    """Subset of ``elektronn3.inference.Predictor`` that SyConn pins (rows P1, P2, P4):
    positional `model` (nn.Module), `tile_shape`/`overlap_shape` (z,y,x), `out_shape` (C,z,y,x),
    `strict_shapes`, `apply_softmax`; ``predict(inp)`` takes np.ndarray / Tensor (N,1,D,H,W) of any float
    dtype and returns a CPU float32 Tensor (N,C,D,H,W)."""

    def __init__(self, model, tile_shape=None, overlap_shape=None, out_shape=None, strict_shapes=False,
                 apply_softmax=True, forward=None):
        self.model = model
        self.tile_shape = None if tile_shape is None else np.asarray(tile_shape)
        self.overlap_shape = None if overlap_shape is None else np.asarray(overlap_shape)
        self.out_shape = None if out_shape is None else tuple(int(s) for s in out_shape)
        self.strict_shapes = strict_shapes
        self.apply_softmax = apply_softmax
        self._forward = forward if forward is not None else model

    @torch.no_grad()
    def _predict(self, t: torch.Tensor) -> torch.Tensor:
        out = self._forward(t)
        if self.apply_softmax:
            out = out.softmax(1)
        return out

    @torch.no_grad()
    def predict(self, inp) -> torch.Tensor:
        inp = torch.as_tensor(np.asarray(inp) if not isinstance(inp, torch.Tensor) else inp).to(torch.float32)
        spatial = np.asarray(inp.shape[2:])
        tile = spatial if self.tile_shape is None else self.tile_shape
        ol = np.zeros_like(tile) if self.overlap_shape is None else self.overlap_shape
        if self.out_shape is None:
            raise ValueError('out_shape is required')
        if self.strict_shapes:
            if np.any(spatial % tile != 0):
                raise ValueError(f'spatial input shape {spatial} is not divisible by tile_shape {tile}')
        elif np.any(spatial % tile != 0):
            raise NotImplementedError('non-strict shapes are not used by SyConn')
        out_shape = (inp.shape[0], *self.out_shape)
        return tiled_apply(self._predict, inp, tile, ol, out_shape)


# ------------------------------------------------------------------------------------------------
# numpy wrappers that exist in the reference tree
def xyz2zyx_ref(vol: np.ndarray) -> np.ndarray:
    """/root/reference/syconn/handler/prediction.py:279-292: swap the last and third-last axes."""
    return vol.swapaxes(-1, -3)


def zyx2xyz_ref(vol: np.ndarray) -> np.ndarray:
    """/root/reference/syconn/handler/prediction.py:295-307."""
    return vol.swapaxes(-1, -3)


def dense_predicton_helper_ref(raw: np.ndarray, predictor, is_zyx=False, return_zyx=False) -> np.ndarray:
    """/root/reference/syconn/handler/prediction.py:846-868: predict, ``*255``, truncating uint8 cast."""
    if not is_zyx:
        raw = xyz2zyx_ref(raw)
    pred = predictor.predict(raw[None, None]).numpy()
    pred = np.array(pred[0]) * 255
    pred = pred.astype(np.uint8)
    if not return_zyx:
        pred = zyx2xyz_ref(pred)
    return pred


def chunkify_ref(lst, n: int) -> List[list]:
    """/root/reference/syconn/handler/basics.py:545-561: ``[lst[i::n] for i in range(min(n, len(lst)))]``."""
    if len(lst) < n:
        n = len(lst)
    return [lst[i::n] for i in range(n)]


def label_rule_ref(pred: np.ndarray, ids: Sequence[int], channel_thresholds: Sequence[Optional[float]]):
    """/root/reference/syconn/handler/prediction.py:813-833 for ONE target: `pred` uint8 (C, ...).

    Single id -> (that channel's uint8 probability map, True [= save_as_raw]); several ids -> uint64 label volume
    where, in id order, ``data[pred[l] > t] = l`` with t = threshold (None -> 255/2; <1 -> 255*t)."""
    data = np.zeros_like(pred[0]).astype(np.uint64)
    save_as_raw = not (len(ids) > 1)
    for label in ids:
        t = channel_thresholds[label]
        if not save_as_raw:
            if t is None:
                t = 255 / 2
            if t < 1.:
                t = 255 * t
            data[pred[label] > t] = label
        else:
            data = pred[label]
    return data, save_as_raw
