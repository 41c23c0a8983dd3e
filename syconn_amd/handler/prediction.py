"""Drop-in for the dense part of ``syconn.handler.prediction`` (/root/reference/syconn/handler/prediction.py:594-868)
plus the ``elektronn3.inference.Predictor`` subset SyConn uses on this path.

Same names, argument meaning and error behaviour as the reference; the compute underneath is the HIP library
(``include/syconn_dense.h``).  There is no CPU fallback: without the built library / an MI355X every entry point
that needs compute raises.
"""
import itertools
import logging
import os
import shutil
from logging import Logger
from typing import Any, Iterable, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from .. import global_params
from ..knossos import ChunkDataset, KnossosDataset
from . import basics
from .basics import chunkify
from .config import initialize_logging

log_main = logging.getLogger('syconn_amd.handler')
log_reps = log_main


# ------------------------------------------------------------------------------------------------------
# axis helpers (prediction.py:279-307)
def xyz2zyx(vol: np.ndarray) -> np.ndarray:
    """[..., X, Y, Z] -> [..., Z, Y, X] (prediction.py:279-292)."""
    return vol.swapaxes(-1, -3)


def zyx2xyz(vol: np.ndarray) -> np.ndarray:
    """[..., Z, Y, X] -> [..., X, Y, Z] (prediction.py:295-307)."""
    return vol.swapaxes(-1, -3)


# storage types whose range guard can fire -> the plan of the same precision class with fp32's exponent range
_FALLBACK = {'f16': 'bf16', 'f16x2': 'f32'}
_ACT_NAMES = {'fp16': 'f16', 'float16': 'f16', 'bfloat16': 'bf16', 'fp32': 'f32', 'float32': 'f32', 'split': 'f16x2'}


# ------------------------------------------------------------------------------------------------------
class Predictor:
    """MI355X implementation of ``elektronn3.inference.Predictor`` (third-party; constructed at
    prediction.py:777-779 and :1062): tiled, overlap-and-crop inference of a 3D U-Net.

    Supported (= what SyConn's dense path uses): `model` as ``nn.Module`` or path to a TorchScript ``.pts`` /
    pickled ``.pt`` file, `state_dict_src`, `device`, `tile_shape`/`overlap_shape` (z,y,x), `out_shape`
    (C,z,y,x), `strict_shapes`, `apply_softmax`, `apply_argmax`, `float16`, `batch_size`, `verbose`.
    `transform`, `augmentations`, `offset` (valid convolutions) and `argmax_with_threshold` are unused by SyConn
    and rejected.

    Precision.  The reference computes in fp32 (`float16=False`, prediction.py:777-779).  `float16=False` (the default, what
    SyConn passes) therefore selects the REFERENCE-PRECISION plan 'f16x2': every activation and weight is kept as two fp16
    numbers hi + lo and every product is three matrix-core passes with fp32 accumulation -- logits within ~1e-6 of the fp32
    oracle's range (stated tolerance 1e-5), argmax / threshold-rule labels equal except where the oracle itself sits on a
    decision boundary, at ~3.4x the time of 'f16'.  `float16=True` (elektronn3: ``model.half()``) selects 'f16'.  The extra
    keyword `act_dtype` overrides both: 'f16x2', 'f16' (0.06 % of the threshold-rule labels of the 128^3 headline tile differ
    from the fp32 oracle, max logit error 8.7e-4 of the logit range), 'bf16' (0.45 %, 7.4e-3; fp32's exponent range, ~3 %
    faster than 'f16'; tests/test_gpu_labels_headline.py), or 'f32' (fp32 storage and FMA arithmetic
    off the matrix cores, ~30x slower than 'f16').  Accumulation is fp32 in every plan.  The fp16-based plans are
    range-guarded: an overflow (> 65504) is detected on the device; with `overflow_fallback` (default: on unless `act_dtype`
    was given explicitly) the prediction is repeated in the next plan with fp32's exponent range ('f16' -> 'bf16', 'f16x2' ->
    'f32') and the Predictor stays there, otherwise ``ActivationOverflowError`` (a ``RuntimeError``) is raised.
    `batch_size`: tiles per launch set (default: automatic,
    see `_batch_for`).  `n_streams` (default: 1, or 2 for single-tile launch sets; env SYCONN_AMD_STREAMS): batches alternate over that many HIP
    streams, each with its own workspace.

    ``predict(inp)`` takes an ``np.ndarray`` / ``Tensor`` of shape (N,1,D,H,W) and any float dtype and returns a
    float32 CPU tensor (N,C,D,H,W).  Device-memory exhaustion raises ``RuntimeError`` (the reference's
    tile-halving loop relies on that, prediction.py:783-794); `strict_shapes` violations raise ``ValueError``.
    """

    def __init__(self, model, state_dict_src=None, device=None, batch_size=None, tile_shape=None,
                 overlap_shape=None, offset=None, out_shape=None, out_dtype=None, float16=False,
                 apply_softmax=True, transform=None, augmentations=None, strict_shapes=False, apply_argmax=False,
                 argmax_with_threshold=None, verbose=False, report_inf_speed=False, act_dtype=None,
                 group_norm_groups=None, n_streams=None, defer_guard=False, overflow_fallback=None, clip_tiles=True,
                 sticky_fallback=True):
        from ..engine import DenseModel, StreamRing
        if transform is not None or augmentations is not None or argmax_with_threshold is not None:
            raise NotImplementedError('transform / augmentations / argmax_with_threshold are not used by SyConn\'s '
                                      'dense path and not implemented')
        if offset is not None and np.any(np.asarray(offset) != 0):
            raise NotImplementedError('valid-convolution offsets are not used by SyConn\'s dense path')
        if isinstance(model, str):
            model = os.path.expanduser(model)
            if not os.path.isfile(model):
                raise ValueError(f'Model path {model} not found.')
            if model.endswith('.pts'):
                model = torch.jit.load(model, map_location='cpu')
            elif model.endswith('.pt'):
                model = torch.load(model, map_location='cpu', weights_only=False)
            else:
                raise ValueError(f'{model} has an unknown file extension. Supported are: .pt and .pts')
        if isinstance(state_dict_src, str):
            state_dict_src = torch.load(state_dict_src, map_location='cpu', weights_only=False)
            state_dict_src = state_dict_src.get('model_state_dict', state_dict_src)
        if state_dict_src is not None and not isinstance(model, dict):
            model.load_state_dict(state_dict_src)
        if hasattr(model, 'eval'):
            model.eval()
        self.model = model
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None or torch.device(device).type != 'cuda':
            raise RuntimeError('syconn_amd.Predictor needs an MI355X (ROCm device); there is no CPU fallback')
        self.device = torch.device(device)
        self.batch_size = batch_size
        self.tile_shape = None if tile_shape is None else np.asarray(tile_shape, dtype=np.int64)
        self.overlap_shape = None if overlap_shape is None else np.asarray(overlap_shape, dtype=np.int64)
        if self.overlap_shape is not None and np.any(self.overlap_shape < 0):
            raise ValueError('overlap_shape must be non-negative')
        self.out_shape = None if out_shape is None else tuple(int(s) for s in out_shape)
        self.out_dtype = torch.float32 if out_dtype is None else out_dtype
        self.float16 = float16
        self.apply_softmax = apply_softmax
        self.strict_shapes = strict_shapes
        self.apply_argmax = apply_argmax
        self.verbose = verbose
        self.report_inf_speed = report_inf_speed
        # fp16 range guard: fp16 storage overflows above 65504 (the reference computes in fp32 and cannot).  The library flags
        # it (sd_model_overflow).  With `overflow_fallback` the prediction is then repeated in the plan of the same precision
        # class that has fp32's exponent range (_FALLBACK) and the Predictor stays there; without it ActivationOverflowError is
        # raised.  Default: on for the default storage type, off for an explicitly requested one.
        self._fallback = (act_dtype is None) if overflow_fallback is None else bool(overflow_fallback)
        # `sticky_fallback`: after an overflow the Predictor STAYS in the fallback plan (default; 'f16x2' -> 'f32' is ~9x slower).  False:
        # only the prediction that overflowed is repeated there and the next one runs in the configured plan again (what dense_predictor
        # asks for: one hot chunk must not slow the worker's remaining chunks); `n_fallbacks` counts the repeats either way.
        self.sticky_fallback = bool(sticky_fallback)
        self.n_fallbacks = 0
        self._dm_fallback = None
        # `defer_guard`: do not synchronise after every prediction (pipelined callers); the caller asks `overflowed()` once
        # its stream of predictions is done and repeats them in bf16 itself
        self.defer_guard = bool(defer_guard)
        # tiles that overhang the volume are predicted on the part of their window that matters (see `_tiled`); same values
        self.clip_tiles = bool(clip_tiles)
        if act_dtype is None:
            # the reference's precision is what `float16` says: False (SyConn's call, prediction.py:777-779) = fp32 results ->
            # the split-fp16 reference-precision plan; True (elektronn3: model.half()) = fp16 storage
            act_dtype = 'f16' if float16 else 'f16x2'
        act_dtype = _ACT_NAMES.get(act_dtype, act_dtype)
        self.act_dtype = act_dtype
        self._gn_groups = group_norm_groups
        self._dm = DenseModel(model, act_dtype=act_dtype, device=self.device, group_norm_groups=group_norm_groups)
        self.out_channels = self._dm.out_channels
        # HIP streams that tiles alternate over.  Given (argument or SYCONN_AMD_STREAMS): that many, always.  Default: one, and
        # two for tiles that run one per launch set (the reference's 178 x 243 x 331 windows): the small deep layers and the
        # launch tail of one tile then run under the big layers of the next (+2.5 ... 5 % on the reference geometry; batches of
        # 128^3 tiles fill the GPU by themselves: +-0)
        if n_streams is None and 'SYCONN_AMD_STREAMS' in os.environ:
            n_streams = int(os.environ['SYCONN_AMD_STREAMS'])
        self._ring = StreamRing(self.device, 1 if n_streams is None else n_streams)
        self._ring2 = StreamRing(self.device, 2) if n_streams is None else None

    # -- geometry --------------------------------------------------------------------------------------
    def _geometry(self, spatial: np.ndarray):
        tile = spatial.copy() if self.tile_shape is None else self.tile_shape
        ol = np.zeros(3, dtype=np.int64) if self.overlap_shape is None else self.overlap_shape
        if len(tile) != 3 or len(ol) != 3:
            raise ValueError('tile_shape / overlap_shape must have 3 entries (z, y, x)')
        if self.out_shape is not None:
            if self.out_shape[0] != self.out_channels:
                raise ValueError(f'out_shape[0]={self.out_shape[0]} but the model predicts {self.out_channels} '
                                 f'channels')
            if tuple(self.out_shape[1:]) != tuple(int(s) for s in spatial):
                raise ValueError(f'out_shape {self.out_shape} does not match the input shape {tuple(spatial)}')
        if np.any(spatial % tile != 0):
            if self.strict_shapes:
                raise ValueError(f'spatial inp shape {tuple(spatial)} has to be divisible by '
                                 f'tile_shape {tuple(tile)} (strict_shapes=True)')
            if np.any(tile > spatial):
                tile = np.minimum(tile, spatial)
        ntiles = np.ceil(spatial / tile).astype(np.int64)
        return tile, ol, ntiles

    def _tiled(self, vol: torch.Tensor, out: torch.Tensor, out_kind: int, label_args=None, halo_included: bool = False,
               valid_box=None):
        """tiled_apply (elektronn3) on the device: zero-padded tile extraction, forward,
        crop of the overlap, write into `out` (C,D,H,W).  vol: (D,H,W) uint8 / float32 on the device.
        `halo_included`: `vol` already carries `overlap_shape` voxels of REAL neighbouring data per side (zeros where the
        dataset ends) instead of being zero-padded here; `out` then covers the inner region only.  This is how a chunk of a
        larger volume is predicted so that the result equals the same tile grid run over the whole volume.
        `valid_box` = ((z0, y0, x0), (z1, y1, x1)) in `out` coordinates: the part of `out` that lies INSIDE the dataset.  The
        reference's chunk grid (``fit_box_size=True``, prediction.py:679-683) covers up to 1.9x the dataset and it predicts
        every tile of every chunk, also those whose whole (cropped) result lies beyond the dataset boundary -- values of a
        zero input that nothing downstream reads.  With a `valid_box` such tiles are not predicted (their region of `out` is
        zero); every voxel inside the dataset is unchanged, because a tile's result depends on nothing but that tile.
        `self.clip_tiles`: a tile that reaches beyond `vol` or the `valid_box` (the dataset ends inside it; or its outer rim
        is the chunk's halo ring, which dense_predictor crops at prediction.py:812) is predicted on the part of its window the
        wanted voxels can depend on (`sd_plan_clip_window`: the borders of every layer stay outside their cones) -- same
        values, less arithmetic; beyond the box `out` is zero; networks with GroupNorm (whole-tile statistics) keep full windows."""
        from ..engine import tile_gather, tile_scatter
        ol_in = np.zeros(3, dtype=np.int64) if self.overlap_shape is None else self.overlap_shape
        spatial = np.asarray(vol.shape, dtype=np.int64) - (2 * ol_in if halo_included else 0)
        shift = ol_in if halo_included else np.zeros(3, dtype=np.int64)
        tile, ol, ntiles = self._geometry(spatial)
        tin = tile + 2 * ol
        single = bool(np.all(ntiles == 1) and np.all(ol == 0) and np.all(tile == spatial) and not halo_included)
        nch = 1 if label_args is not None else self.out_channels

        def run(inp, outp, slot, roi=None):           # inp (n,D,H,W) -> outp (n,nch,D,H,W); roi: the part of the window that is kept
            if label_args is not None:
                self._dm.forward_labels_batch(inp, label_args[0], label_args[1], out=outp[:, 0], slot=slot, roi=roi)
            else:
                self._dm.forward_batch(inp, out_kind, outp, slot=slot, roi=roi)

        if single:
            run(vol[None], out[None], 0)
            return
        # independent tiles go through the network `nb` at a time (sd_forward_batch: one set of launches, every
        # kernel sees nb times as many blocks); batches alternate over `n_streams` HIP streams.
        # Which tiles run, on which window (full, or clipped at the far side to what the wanted voxels depend on), keeping which box:
        # syconn_amd.tiling.plan_tile_windows -- the same arithmetic parallel.predict_volume_distributed orders its rounds by
        from ..tiling import plan_tile_windows
        by_window, zero = plan_tile_windows(spatial, tile, ol, ntiles, valid_box, self.clip_tiles, self._dm.clipped_window,
                                            self._dm.has_groupnorm)
        if not by_window:
            out.zero_()
            return
        if zero:                                 # beyond the dataset `out` reads zero, whatever was skipped or clipped
            out.zero_()
        nb = self._batch_for(tin, max(len(g) for g in by_window.values()))
        n_tiles = sum(len(g) for g in by_window.values())
        ring = self._ring2 if (self._ring2 is not None and nb == 1 and n_tiles > 1) else self._ring
        tbuf = [torch.empty(nb * int(np.prod(tin)), dtype=vol.dtype, device=self.device) for _ in range(ring.n)]
        obuf = [torch.empty(nb * nch * int(np.prod(tin)), dtype=out.dtype, device=self.device) for _ in range(ring.n)]
        i = 0
        with ring:
            for (win, roi), tiles in by_window.items():
                nvox = int(np.prod(win))
                for b0 in range(0, len(tiles), nb):
                    group = tiles[b0:b0 + nb]
                    n, k = len(group), ring.slot(i)
                    tb = tbuf[k][:n * nvox].view(n, *win)
                    ob = obuf[k][:n * nch * nvox].view(n, nch, *win)
                    with ring.stream(i):
                        for j, (lo, start, _, _) in enumerate(group):
                            tile_gather(vol, lo - ol + shift + start, win, tb[j])
                        run(tb, ob, k, roi)
                        for j, (lo, start, w0, w1) in enumerate(group):
                            tile_scatter(ob[j], ol + w0 - start, w1 - w0, out, lo + w0)
                    i += 1

    def chunk_cost_model(self, halo, halo_included: bool, skip_outside: bool = True):
        """Cost of a chunk of a chunked volume prediction through this Predictor, from geometry alone (``syconn_amd.tiling``)."""
        from ..tiling import ChunkCostModel, PlanClipper
        return ChunkCostModel(PlanClipper(self._dm._ops_arr), self.tile_shape, self.overlap_shape, halo, halo_included,
                              self.clip_tiles, skip_outside)

    def _guarded(self, run):
        """Run one tiled prediction; if the fp16 range guard fired, repeat it in bf16 (default storage type) or raise."""
        from ..engine import DenseModel
        run()
        if self.defer_guard or self.act_dtype not in _FALLBACK or not self._dm.overflowed():
            return
        nxt = _FALLBACK[self.act_dtype]
        if not self._fallback:
            raise L.ActivationOverflowError(
                f"fp16 activation overflow (a stored activation exceeded 65504): results invalid; use act_dtype='{nxt}'")
        self.n_fallbacks += 1
        if self._dm_fallback is None:
            self._dm_fallback = DenseModel(self.model, act_dtype=nxt, device=self.device, group_norm_groups=self._gn_groups)
        if self.sticky_fallback:
            log_main.warning(f'syconn_amd.Predictor: fp16 activation overflow detected -- switching this Predictor from '
                             f'{self.act_dtype} to {nxt} storage and repeating the prediction')
            self.act_dtype = nxt
            self._dm, self._dm_fallback = self._dm_fallback, None
            run()
            return
        log_main.warning(f'syconn_amd.Predictor: fp16 activation overflow detected -- repeating this prediction in {nxt} storage '
                         f'(fallback {self.n_fallbacks}); the next one runs in {self.act_dtype} again')
        main, self._dm = self._dm, self._dm_fallback
        try:
            run()
        finally:
            self._dm = main

    def overflowed(self) -> bool:
        """fp16 range guard of the predictions since the last call (synchronises the current stream); for `defer_guard`."""
        return self._dm.overflowed()

    @torch.no_grad()
    def probe_memory(self, spatial_shape) -> None:
        """What dense_predictor's warm-up prediction of a zero chunk is for (prediction.py:781-794: a CUDA out-of-memory
        ``RuntimeError`` there makes the caller halve the tile shape): reserve everything a prediction of a (D,H,W) volume holds
        at the same time -- result tensor, tile and output buffers and the workspaces of every stream for the largest window --
        and run ONE launch set of zero tiles per stream (first-launch costs of the kernels), instead of predicting all tiles of
        the chunk (12 in the reference's geometry: 0.2 s of a worker's life in the reference-precision plan)."""
        spatial = np.asarray(spatial_shape, dtype=np.int64)
        tile, ol, ntiles = self._geometry(spatial)
        tin = tile + 2 * ol
        n_all = int(np.prod(ntiles))
        nb = self._batch_for(tin, n_all)
        ring = self._ring2 if (self._ring2 is not None and nb == 1 and n_all > 1) else self._ring
        torch.cuda.set_device(self.device)
        shape = [int(t) for t in tin]
        out = torch.empty((self.out_channels, *[int(s) for s in spatial]), dtype=torch.uint8, device=self.device)
        tb = [torch.zeros((nb, *shape), dtype=torch.uint8, device=self.device) for _ in range(ring.n)]
        ob = [torch.empty((nb, self.out_channels, *shape), dtype=torch.uint8, device=self.device) for _ in range(ring.n)]
        roi = (tuple(int(v) for v in ol), tuple(int(v) for v in ol + tile)) if (self.clip_tiles and np.any(ol > 0)) else None
        # (a tile with another box -- boundary tiles keep whole windows, the level-0 decoder is replaced by its layers -- may need a
        # larger workspace than the interior one: reserve for both forms, inside the caller's tile-halving try block)
        with ring:
            for r in ([roi, None] if roi is not None else [None]):
                for k in range(ring.n):
                    with ring.stream(k):
                        self._dm.forward_batch(tb[k], L.SD_OUT_PROBS_U8, ob[k], slot=ring.slot(k), roi=r)
        torch.cuda.current_stream(self.device).synchronize()
        self._dm.overflowed()                    # (clears the range-guard flag; zeros cannot overflow)
        del out, tb, ob

    def _batch_for(self, tin, ntiles: int) -> int:
        """Tiles per launch set: `batch_size` if given (elektronn3's Predictor argument), else as many as keep the
        workspaces of one batch under ~8 GiB (at most 8): 128^3 tiles run 8 at a time, the reference's
        178x243x331 tiles (7.4 GiB of activations each) one at a time."""
        if self.batch_size is not None:
            return max(1, min(int(self.batch_size), ntiles))
        per_tile = self._dm.workspace_bytes(tuple(int(t) for t in tin))
        return int(max(1, min(8, ntiles, (8 << 30) // max(per_tile, 1))))

    # -- public API ------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, inp: Union[np.ndarray, torch.Tensor]) -> torch.Tensor:
        """(N,1,D,H,W) float -> (N,C,D,H,W) float32 CPU tensor (rows P2/P4; called at prediction.py:781, 863)."""
        if isinstance(inp, np.ndarray):
            inp = torch.from_numpy(np.ascontiguousarray(inp))
        if inp.dim() != 5:
            raise ValueError(f'expected (N, C, D, H, W) input, got shape {tuple(inp.shape)}')
        if inp.shape[1] != 1:
            raise ValueError('the dense path feeds single-channel EM data (C must be 1)')
        n = inp.shape[0]
        spatial = tuple(int(s) for s in inp.shape[2:])
        out = torch.empty((n, self.out_channels, *spatial), dtype=torch.float32)
        kind = L.SD_OUT_PROBS_F32 if self.apply_softmax else L.SD_OUT_LOGITS_F32
        torch.cuda.set_device(self.device)
        out_dev = torch.empty((self.out_channels, *spatial), dtype=torch.float32, device=self.device)
        for b in range(n):
            vol = inp[b, 0].to(torch.float32).contiguous().to(self.device)
            self._guarded(lambda: self._tiled(vol, out_dev, kind))
            out[b] = out_dev.cpu()
        if self.apply_argmax:
            out = out.argmax(1)
        return out

    @torch.no_grad()
    def predict_proba_u8_device(self, raw_u8: torch.Tensor, halo_included: bool = False, valid_box=None) -> torch.Tensor:
        """Fast path of ``dense_predicton_helper(raw.astype(float32)/255., self)``: `raw_u8` is the (D,H,W) uint8
        chunk ON THE DEVICE; returns uint8 ``floor(255*softmax)`` (C,D,H,W) on the device.  Bit-identical to the
        slow path by construction: the kernel normalises with the table float32(v)/255 (prediction.py:808) and
        truncates the float32 product p*255 (prediction.py:864-865)."""
        if not self.apply_softmax:
            raise ValueError('uint8 probabilities need apply_softmax=True')
        assert raw_u8.dtype == torch.uint8 and raw_u8.dim() == 3
        torch.cuda.set_device(self.device)
        raw_u8 = raw_u8.contiguous()
        inner = tuple(int(s) - (2 * int(o) if halo_included else 0) for s, o in
                      zip(raw_u8.shape, (self.overlap_shape if self.overlap_shape is not None else (0, 0, 0))))
        out = torch.empty((self.out_channels, *inner), dtype=torch.uint8, device=self.device)
        self._guarded(lambda: self._tiled(raw_u8, out, L.SD_OUT_PROBS_U8, halo_included=halo_included, valid_box=valid_box))
        return out


    @torch.no_grad()
    def predict_labels_u8_device(self, raw_u8: torch.Tensor, ids, thresholds, halo_included: bool = False,
                                 valid_box=None) -> torch.Tensor:
        """`predict_proba_u8_device` followed by the label rule of dense_predictor (prediction.py:813-833) for ONE
        multi-id target, evaluated in the network's final epilogue: (D,H,W) uint8 labels on the device.  `thresholds`
        are the resolved uint8-scale values, one per id."""
        if not self.apply_softmax:
            raise ValueError('labels need apply_softmax=True')
        assert raw_u8.dtype == torch.uint8 and raw_u8.dim() == 3
        torch.cuda.set_device(self.device)
        raw_u8 = raw_u8.contiguous()
        inner = tuple(int(s) - (2 * int(o) if halo_included else 0) for s, o in
                      zip(raw_u8.shape, (self.overlap_shape if self.overlap_shape is not None else (0, 0, 0))))
        out = torch.empty((1, *inner), dtype=torch.uint8, device=self.device)
        self._guarded(lambda: self._tiled(raw_u8, out, L.SD_OUT_PROBS_U8,
                                          label_args=([int(i) for i in ids], [float(t) for t in thresholds]),
                                          halo_included=halo_included, valid_box=valid_box))
        return out[0]


# ------------------------------------------------------------------------------------------------------
def dense_predicton_helper(raw: np.ndarray, predictor: 'Predictor', is_zyx=False, return_zyx=False) -> np.ndarray:
    """prediction.py:846-868.  `raw`: float array (X,Y,Z) or (Z,Y,X) already scaled to 0..1; returns the
    inference result as uint8 (C, ...) between 0..255 (truncated ``pred*255``)."""
    vol_zyx = raw if is_zyx else xyz2zyx(raw)
    probs = predictor.predict(vol_zyx[None, None]).numpy()[0]           # (C, Z, Y, X) float32, batch axis dropped
    out = (np.array(probs) * 255).astype(np.uint8)                      # float32 product, truncating cast (:864-865)
    return out if return_zyx else zyx2xyz(out)


def _resolve_threshold(t) -> float:
    """prediction.py:824-827: ``None -> 255/2``; ``t < 1 -> 255*t``."""
    if t is None:
        t = 255 / 2
    if t < 1.:
        t = 255 * t
    return float(t)


def dense_predictor(args):
    """Worker of the dense prediction (one process = one GPU), prediction.py:723-843.  `args` is the reference's
    14-tuple ``(chunk_ids, kd_p, target_p, model_p, overlap_shape, overlap_shape_tiles, tile_shape, chunk_size,
    n_channel, target_channels, target_kd_path_list, channel_thresholds, mag, cube_of_interest)``.

    Differences to the reference are confined to WHERE the arithmetic runs: the uint8 chunk goes to the GPU once,
    normalisation, tiling, U-Net, softmax, uint8 cast, halo crop and the label rule run there, and only uint8
    results come back (the reference moves fp32 tiles over PCIe in both directions)."""
    from ..engine import mag_pyramid, postproc_labels, tile_scatter
    chunk_ids, kd_p, target_p, model_p, overlap_shape, overlap_shape_tiles, tile_shape, chunk_size, n_channel, \
        target_channels, target_kd_path_list, channel_thresholds, mag, cube_of_interest = args
    import time as _time
    t_start = _time.perf_counter()

    kd = KnossosDataset()
    kd.initialize_from_knossos_path(kd_p)
    cd = ChunkDataset()
    cd.initialize(kd, cube_of_interest[1], chunk_size, target_p + '/cd_tmp/', box_coords=cube_of_interest[0],
                  list_of_coords=[], fit_box_size=True, overlap=overlap_shape)
    target_kd_dict = {path: basics.kd_factory(path) for path in target_kd_path_list}
    if not os.environ.get('SYCONN_AMD_NO_WRITE_COMBINING'):
        # chunks are not aligned to the target cubes: assemble the cubes this worker touches in memory and write each once
        # (KnossosDataset.enable_write_combining; flushed at the end of this worker)
        for tkd in target_kd_dict.values():
            tkd.enable_write_combining(chunk_shape=chunk_size)

    ix = 0
    tile_shape = np.array(tile_shape)
    overlap_shape = np.asarray(overlap_shape)
    overlap_shape_tiles = np.asarray(overlap_shape_tiles)
    chunk_size = np.asarray(chunk_size)
    # storage type of the activations: config['dense_prediction']['act_dtype']; default 'f16x2' = the reference's precision
    # (the fast plans 'f16' / 'bf16' are an explicit configuration choice).  A range-guard overflow on any chunk falls back
    # to the plan with fp32's exponent range instead of killing the worker.
    act_dtype = global_params.config['dense_prediction']['act_dtype'] if _wd_set() else 'f16x2'
    skip_outside = bool(global_params.config['dense_prediction'].get('skip_tiles_outside_dataset', True)) if _wd_set() else True
    clip_tiles = bool(global_params.config['dense_prediction'].get('clip_boundary_tiles', True)) if _wd_set() else True
    sticky = bool(global_params.config['dense_prediction'].get('sticky_overflow_fallback', False)) if _wd_set() else False
    log_main.info(f'dense_predictor: activation storage type {act_dtype} '
                  f'({"reference precision" if act_dtype in ("f16x2", "f32") else "reduced precision, fast plan"})')
    while True:
        try:
            out_shape = (chunk_size + 2 * np.array(overlap_shape)).astype(np.int32)[::-1]  # ZYX
            out_shape = np.insert(out_shape, 0, n_channel)  # output must equal chunk size
            predictor = Predictor(model_p, strict_shapes=True, tile_shape=tile_shape[::-1], out_shape=out_shape,
                                  overlap_shape=overlap_shape_tiles[::-1], apply_softmax=True, act_dtype=act_dtype,
                                  overflow_fallback=True, clip_tiles=clip_tiles, sticky_fallback=sticky)
            try:
                predictor.model.ae = False
            except Exception:  # ScriptModules refuse new attributes; elektronn3's flag has no meaning here
                pass
            # warm-up / memory probe for a full-size chunk (prediction.py:781 predicts float64 zeros of that size)
            if os.environ.get('SYCONN_AMD_FULL_WARMUP'):      # (A/B switch: the reference's full-chunk warm-up)
                _ = predictor.predict_proba_u8_device(
                    torch.zeros(tuple(int(s) for s in out_shape[1:]), dtype=torch.uint8, device=predictor.device))
            else:
                predictor.probe_memory(tuple(int(s) for s in out_shape[1:]))
            break
        except L.ActivationOverflowError:      # (a RuntimeError, but not a memory problem: halving tiles would not help)
            raise
        except RuntimeError:  # device MemoryError
            if np.all(tile_shape % 2):
                raise ValueError('Cannot reduce tile shape anymore. Please adapt the tile/overlap/chunk shape in '
                                 'the function that is calling `dense_predictor`.')
            while tile_shape[ix] % 2:
                ix += 1
            tile_sh_orig = np.array(tile_shape)
            tile_shape[ix] = tile_shape[ix] // 2
            log_main.warning(f'Changed tile shape from {tile_sh_orig} to {tile_shape} to reduce memory requirements.')
            ix = (ix + 1) % 3

    # Host I/O is pipelined around the GPU: one thread reads chunk k+1 from the KnossosDataset while chunk k is
    # predicted, one thread writes the results of chunk k-1 (cube files of neighbouring chunks overlap, so writes stay
    # sequential and in chunk order).  The reference does read -> predict -> write strictly in sequence.
    from concurrent.futures import ThreadPoolExecutor
    dev = predictor.device
    torch.cuda.set_device(dev)
    s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    spent = {'read': 0.0, 'gpu': 0.0, 'write': 0.0}       # seconds per pipeline stage (SYCONN_AMD_IO_TIMING=1 logs them)
    t_setup = _time.perf_counter() - t_start              # datasets opened, model planned, workspace reserved

    read_bufs = {}

    def read_chunk(ch_id):
        t0 = _time.perf_counter()
        ch = cd.chunk_dict[ch_id]
        ol = ch.overlap
        size = np.array(np.array(ch.size) + 2 * np.array(ol), dtype=np.int32)
        coords = np.array(np.array(ch.coordinates) - np.array(ol), dtype=np.int32)
        # page-locked read buffers, reused round robin (3: one being filled, one uploading, one spare): a fresh 80 MB array per
        # chunk costs its page faults, and a pageable upload runs at a fraction of the link rate
        shape_zyx = tuple(int(s) for s in size[::-1])
        if read_bufs.get('shape') != shape_zyx:
            read_bufs.update(shape=shape_zyx, bufs=[torch.empty(shape_zyx, dtype=torch.uint8).pin_memory() for _ in range(3)], evs=[None] * 3, k=0)
        k = read_bufs['k'] % 3
        read_bufs['k'] += 1
        if read_bufs['evs'][k] is not None:
            read_bufs['evs'][k].synchronize()            # its previous upload has left the buffer
        hbuf = read_bufs['bufs'][k]
        if isinstance(kd, KnossosDataset):
            kd.load_raw(size=size * mag, offset=coords * mag, mag=mag, out=hbuf.numpy())          # uint8, ZYX
        else:                                            # (a foreign dataset object: its own allocation, then one copy)
            hbuf.numpy()[...] = kd.load_raw(size=size * mag, offset=coords * mag, mag=mag)
        with torch.cuda.stream(s_in):
            out_dev = hbuf.to(dev, non_blocking=True)
            ev = s_in.record_event()
        read_bufs['evs'][k] = ev
        spent['read'] += _time.perf_counter() - t0
        return out_dev, ev

    pin_pool = {}                        # page-locked download buffers by shape (a pageable D2H copy runs at a fraction of the link rate)

    def write_chunk(jobs, ev):
        t0 = _time.perf_counter()
        # download on the copy-out stream, from the writer thread: only waits for THIS chunk's kernels (event), not for the
        # next chunk's work that the main thread has queued on the compute stream meanwhile
        host = []
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev)
            for save, kwargs, lvl in jobs:
                free = pin_pool.setdefault(tuple(lvl.shape), [])
                h = free.pop() if free else torch.empty(tuple(lvl.shape), dtype=lvl.dtype).pin_memory()
                h.copy_(lvl, non_blocking=True)
                host.append((save, kwargs, h))
            s_out.synchronize()
        for save, kwargs, h in host:
            save(data=h.numpy(), **kwargs)       # (uint8 labels of an overlay dataset are widened cube by cube inside save_seg)
            pin_pool[tuple(h.shape)].append(h)
        spent['write'] += _time.perf_counter() - t0

    class _Inline:                       # SYCONN_AMD_SEQ_IO=1: no overlap, everything in the caller's thread (debugging / A-B)
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def submit(self, fn, *a):
            from concurrent.futures import Future
            f = Future()
            f.set_result(fn(*a))
            return f

    seq_io = bool(os.environ.get('SYCONN_AMD_SEQ_IO'))
    chunk_ids = list(chunk_ids)
    with (_Inline() if seq_io else ThreadPoolExecutor(max_workers=1)) as reader, \
            (_Inline() if seq_io else ThreadPoolExecutor(max_workers=1)) as writer:
        nxt = reader.submit(read_chunk, chunk_ids[0]) if chunk_ids else None
        writes = []
        for n, ch_id in enumerate(chunk_ids):
            ch = cd.chunk_dict[ch_id]
            ol = ch.overlap
            raw_dev, ev_in = nxt.result()
            nxt = reader.submit(read_chunk, chunk_ids[n + 1]) if n + 1 < len(chunk_ids) else None
            t_gpu = _time.perf_counter()
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev_in)
            raw_dev.record_stream(cur)
            # one multi-id target (mivcsj, syntype): only its label volume is needed -> label rule in the final epilogue
            only_labels = len(target_channels) == 1 and len(target_channels[0]) > 1
            # the chunk grid (fit_box_size=True) overhangs the dataset: model tiles whose whole cropped result lies beyond the
            # boundary (or in the halo ring that is cropped below) are not predicted -- zeros instead of the network's answer to
            # a zero input, in a region nothing reads (config['dense_prediction']['skip_tiles_outside_dataset'])
            vbox = None
            if skip_outside:
                c0 = np.asarray(ch.coordinates) - np.asarray(ol)                      # origin of chunk + halo, xyz
                v_lo = np.maximum(np.asarray(ol), -c0)
                v_hi = np.minimum(np.asarray(ol) + np.asarray(ch.size), np.asarray(kd.boundary) // mag - c0)
                vbox = (tuple(int(v) for v in v_lo[::-1]), tuple(int(v) for v in v_hi[::-1]))
            if only_labels:
                ids0 = target_channels[0]
                thr0 = [_resolve_threshold(channel_thresholds[label]) for label in ids0]
                pred_dev = predictor.predict_labels_u8_device(raw_dev, ids0, thr0, valid_box=vbox)[None]   # (1, Z, Y, X) uint8 labels
            else:
                pred_dev = predictor.predict_proba_u8_device(raw_dev, valid_box=vbox)   # (C, Z, Y, X) uint8
            # slice out the original input volume along ZYX (prediction.py:812)
            zyx = tuple(int(s) for s in np.asarray(ch.size)[::-1])
            crop = torch.empty((pred_dev.shape[0], *zyx), dtype=torch.uint8, device=dev)
            tile_scatter(pred_dev, (int(ol[2]), int(ol[1]), int(ol[0])), zyx, crop, (0, 0, 0))
            jobs = []
            for j in range(len(target_channels)):
                ids = target_channels[j]
                path = target_kd_path_list[j]
                save_as_raw = not (len(ids) > 1)
                # the mag pyramid [mag, 2*mag, 4*mag] (order-0, fast_resampling=True) is built on the device; each level
                # is written with its own data_mag, which is what one save_*(mags=[mag, 2*mag, 4*mag]) call produces
                if save_as_raw:
                    # no thresholding and only one label in the target KnossosDataset -> store probability map
                    for k, lvl in enumerate(mag_pyramid(crop[ids[-1]], 3)):
                        lvl.record_stream(s_out)
                        jobs.append((target_kd_dict[path].save_raw,
                                     dict(offset=ch.coordinates * mag, data_mag=mag * 2 ** k,
                                          mags=[mag * 2 ** k], fast_resampling=True, upsample=False), lvl))
                else:
                    thr = [_resolve_threshold(channel_thresholds[label]) for label in ids]
                    lab = crop[0] if only_labels else postproc_labels(crop, ids, thr, out_dtype=torch.uint8)
                    for k, lvl in enumerate(mag_pyramid(lab, 3)):  # uint8 on the device and over PCIe, widened by the writer
                        lvl.record_stream(s_out)
                        jobs.append((target_kd_dict[path].save_seg,
                                     dict(offset=ch.coordinates * mag, data_mag=mag * 2 ** k, mags=[mag * 2 ** k],
                                          fast_resampling=True, upsample=False), lvl))
            ev_done = cur.record_event()
            spent['gpu'] += _time.perf_counter() - t_gpu
            writes.append(writer.submit(write_chunk, jobs, ev_done))
            while len(writes) > 2:                   # bound the host memory held by queued results
                writes.pop(0).result()
        for w in writes:
            w.result()
    t0 = _time.perf_counter()
    for tkd in target_kd_dict.values():
        tkd.flush()
    t_flush = _time.perf_counter() - t0
    spent['write'] += t_flush
    if predictor.n_fallbacks:
        log_main.warning(f'dense_predictor: {predictor.n_fallbacks} of {len(chunk_ids)} chunk(s) overflowed fp16 storage and were predicted '
                         f'in the fallback plan ({_FALLBACK.get(act_dtype, "?")})')
    if os.environ.get('SYCONN_AMD_IO_TIMING'):
        log_main.warning('dense_predictor stages over %d chunk(s): read + H2D %.2f s (reader thread), launch %.2f s (main thread), '
                         'D2H + write %.2f s (writer thread; includes waiting for the kernels and %.2f s of final flush); setup %.2f s, total %.2f s',
                         len(chunk_ids), spent['read'], spent['gpu'], spent['write'], t_flush, t_setup, _time.perf_counter() - t_start)


def _wd_set() -> bool:
    try:
        _ = global_params.config.working_dir
        return True
    except ValueError:
        return False


def _dense_targets(n_channel, target_names, target_channels, channel_thresholds):
    """Defaults and validation of the per-target arguments (prediction.py:655-664): one target 'pred' holding every
    channel, no thresholds; names and channel tuples must pair up."""
    names = ['pred'] if target_names is None else list(target_names)
    channels = [list(range(n_channel))] if target_channels is None else [list(c) for c in target_channels]
    if len(names) != len(channels):
        msg = 'For every target name the target channels have to be specified.'
        log_reps.error(msg)
        raise ValueError(msg)
    thresholds = [None] * n_channel if channel_thresholds is None else list(channel_thresholds)
    return names, channels, thresholds


def _dense_geometry():
    """(overlap, tile overlap, chunk size, tile shape) in x,y,z.  The reference hard-codes them and marks them as
    future config parameters (prediction.py:671-677); here they are ``config['dense_prediction']`` with those values as
    defaults.  Chunk halo == tile halo, as in the reference."""
    geo = global_params.config['dense_prediction']
    halo = np.array(geo['overlap_shape_tiles'])
    return halo, halo, np.array(geo['chunk_size']), list(geo['tile_shape'])


def _create_target_kds(paths, kd, cube_shape_kd, overwrite, log):
    """One empty KnossosDataset per target with a 6-level mag list and a knossos.conf (prediction.py:685-706).
    All existence checks run before anything is deleted or created, so a refused call leaves every target untouched."""
    existing = [p for p in paths if os.path.isdir(p)]
    if existing and not overwrite:
        msg = f'Found existing KD at "{existing[0]}" but overwrite is set to False.'
        log.error(msg)
        raise ValueError(msg)
    for p in existing:
        log.debug('Found existing KD at {}. Removing it now.'.format(p))
        shutil.rmtree(p)
    scale = np.array(global_params.config['scaling'])
    mags = [1 << k for k in range(6)]
    for p in paths:
        tkd = KnossosDataset()
        tkd._cube_shape = cube_shape_kd
        tkd.scales = [scale, ]
        tkd.initialize_without_conf(p, kd.boundary, kd.scale, kd.experiment_name, mags, create_pyk_conf=False,
                                    create_knossos_conf=True)
        try:
            basics.kd_factory(p)
        except ValueError as e:
            log.error(f'Could not initialize KnossosDataset at "{p}". {e}')


def predict_dense_to_kd(kd_path: str, target_path: str, model_path: str, n_channel: int,
                        target_names: Optional[Iterable[str]] = None,
                        target_channels: Optional[Iterable[Iterable[int]]] = None,
                        channel_thresholds: Optional[Iterable[Union[float, Any]]] = None,
                        log: Optional[Logger] = None, mag: int = 1,
                        overlap_shape_tiles: Tuple[int, int, int] = (40, 40, 20),
                        cube_of_interest: Optional[Tuple[np.ndarray]] = None, overwrite: bool = False,
                        cube_shape_kd: Optional[Tuple[int]] = None):
    """Dense prediction of a whole KnossosDataset into target KnossosDataset(s), prediction.py:594-720: builds the
    chunk grid, creates the target datasets, partitions the chunk ids round-robin over ``config.ngpu_total`` workers
    (``chunkify``) and dispatches one worker per GPU.

    As in the reference the keyword `overlap_shape_tiles` is overridden by the configured geometry
    (prediction.py:671-677 hard-codes it; here it is ``config['dense_prediction']``, same defaults)."""
    from ..mp import batchjob_utils as qu
    conf = global_params.config
    if log is None:
        log = initialize_logging('dense_predictions', conf.working_dir + '/logs/', overwrite=False)
    names, channels, thresholds = _dense_targets(n_channel, target_names, target_channels, channel_thresholds)
    overlap_shape, overlap_shape_tiles, chunk_size, tile_shape = _dense_geometry()

    kd = basics.kd_factory(kd_path)
    coi = (np.zeros(3, ), kd.boundary // mag) if cube_of_interest is None else cube_of_interest
    grid = ChunkDataset()
    grid.initialize(kd, coi[1], chunk_size, target_path + '/cd_tmp/', box_coords=coi[0], list_of_coords=[],
                    fit_box_size=True, overlap=overlap_shape)
    chunk_ids = list(grid.chunk_dict.keys())

    target_kds = [f'{target_path}/{name}/' for name in names]
    _create_target_kds(target_kds, kd, (256, 256, 256) if cube_shape_kd is None else cube_shape_kd, overwrite, log)

    # one 14-tuple per worker: its share of the chunk ids (static round-robin) + everything needed to rebuild the state
    shared = (kd_path, target_path, model_path, overlap_shape, overlap_shape_tiles, tile_shape, chunk_size, n_channel,
              channels, target_kds, thresholds, mag, coi)
    jobs = [(share, *shared) for share in chunkify(chunk_ids, conf.ngpu_total)]
    log.info('Started dense prediction of {} in {:d} chunk(s), activation storage type {}.'.format(
        ", ".join(names), len(chunk_ids), conf['dense_prediction']['act_dtype']))
    cores = conf['ncores_per_node']
    if qu.batchjob_enabled():
        cores //= conf['ngpus_per_node']
    qu.batchjob_script(jobs, "predict_dense", n_cores=cores, suffix='_' + '_'.join(names), remove_jobfolder=True,
                       log=log, additional_flags="--gres=gpu:1")
    log.info('Finished dense prediction of {}'.format(", ".join(names)))


def get_myelin_cnn():
    """prediction.py:1047-1063: ``Predictor(torch.jit.load(mpath_myelin))`` with the default tiling of that loader."""
    return Predictor(global_params.config.mpath_myelin, strict_shapes=False, apply_softmax=True)
