"""Thin Python handle on an ``sd_model`` of the HIP library plus the device-side tiling helpers.

PyTorch is used for device memory (caching allocator), streams and H2D/D2H only; every computation is a
hand-written gfx950 kernel behind the C ABI (``include/syconn_dense.h``).
"""
import ctypes as C
from typing import Optional, Sequence

import numpy as np
import threading

import torch

from . import _lib as L
from .plan import plan_from_model

# 'f16x2': the reference-precision plan on the matrix cores (csrc/sd_split.hip: every value kept as fp16 hi + lo, three MFMA
# passes per product, fp32-level logits at ~3x the cost of 'f16'); 'f32': fp32 storage and FMA arithmetic (csrc/sd_f32.hip, slow)
_ACT = {'bf16': L.SD_BF16, 'bfloat16': L.SD_BF16, 'f16': L.SD_F16, 'fp16': L.SD_F16, 'float16': L.SD_F16,
        'f16x2': L.SD_F16X2, 'split': L.SD_F16X2,
        'f32': L.SD_F32, 'fp32': L.SD_F32, 'float32': L.SD_F32}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_ready = threading.local()      # per thread (sd_init also selects the HIP device of the CALLING thread): the device sd_init last
                                # succeeded for -- the check costs 0.14 ms and the tile helpers call this per tile


def require_gpu(device_index: int = 0):
    """Fail loudly when the HIP path cannot run (no CPU fallback exists)."""
    lib = L.load()
    # (the cache is only valid while the thread's current device still is the one sd_init selected: a torch.cuda.device(...)
    # block or a set_device elsewhere changes it behind our back)
    if getattr(_ready, 'device', None) == int(device_index) and torch.cuda.current_device() == int(device_index):
        return lib
    if not torch.cuda.is_available():
        raise RuntimeError('syconn_amd: no MI355X visible to PyTorch-ROCm; this package has no CPU fallback')
    L.check(lib.sd_init(int(device_index)), 'sd_init')
    _ready.device = int(device_index)
    return lib


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.uint8:
        return L.SD_U8
    if t.dtype == torch.float32:
        return L.SD_F32
    raise ValueError(f'unsupported buffer dtype {t.dtype}')


class DenseModel:
    """One network (plan + packed weights) resident on one GPU."""

    def __init__(self, model, act_dtype: str = 'bf16', device: Optional[torch.device] = None,
                 group_norm_groups: Optional[int] = None):
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('syconn_amd.DenseModel needs a ROCm device (no CPU fallback)')
        self.lib = require_gpu(self.device.index or 0)
        torch.cuda.set_device(self.device)
        ops, blob, info = plan_from_model(model, group_norm_groups)
        self.info = info
        self.n_ops = len(ops)
        self.op_kinds = [int(o.kind) for o in ops]
        self.has_groupnorm = L.SD_OP_GROUPNORM in self.op_kinds      # (statistics over whole tiles: no output box, no clipped windows)
        self.ops = ops
        self.out_channels = info['out_channels']
        self.act_dtype = act_dtype
        arr = self._ops_arr = (L.OpDesc * len(ops))(*ops)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        handle = C.c_void_p()
        L.check(self.lib.sd_model_create(arr, len(ops), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size,
                                         _ACT[act_dtype], C.byref(handle)), 'sd_model_create')
        self._h = handle
        self._ws_slots = {}            # workspace per slot: forwards on different HIP streams use different slots
        self._profile = False
        self._clip_cache = {}
        self._roi = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self.lib.sd_model_destroy(h)
            self._h = None

    def clipped_window(self, lo: int, hi: int, full: int, axis: int):
        """(start, extent) of the part of an input window of `full` voxels along `axis` on which the outputs lo <= index < hi
        are what they are on the whole window (`sd_plan_clip_window`); extents in multiples of 8 so that the poolings of the
        clipped window stay whole."""
        key = (lo, hi, full, axis)
        w = self._clip_cache.get(key)
        if w is None:
            start, extent = C.c_int32(), C.c_int32()
            L.check(self.lib.sd_plan_clip_window(self._ops_arr, self.n_ops, axis, lo, hi, full, 8, C.byref(start),
                                                 C.byref(extent)), 'sd_plan_clip_window')
            w = self._clip_cache[key] = (int(start.value), int(extent.value))
        return w

    def _set_roi(self, roi=None):
        """Output box of interest ((z0, y0, x0), (z1, y1, x1)) in tile coordinates, or None for whole tiles (`sd_model_set_roi`): decoder
        layers then compute only what the box depends on; values inside the box are unchanged, the rest of the output is unspecified.
        The box is state of the sd_model handle; the forward methods own it: every call states its box through its `roi` argument
        (default: whole tiles) and switches the handle when that differs from the last call's.  A DenseModel -- like the handle,
        include/syconn_dense.h -- serves ONE host thread at a time (dims, buffer offsets and the box are recomputed per forward)."""
        if roi is None:
            L.check(self.lib.sd_model_set_roi(self._h, None, None), 'sd_model_set_roi')
        else:
            lo = (C.c_int32 * 3)(*[int(v) for v in roi[0]])
            hi = (C.c_int32 * 3)(*[int(v) for v in roi[1]])
            L.check(self.lib.sd_model_set_roi(self._h, lo, hi), 'sd_model_set_roi')
        self._roi = roi

    # -- workspace ------------------------------------------------------------------------------------
    def workspace_bytes(self, shape: Sequence[int]) -> int:
        n = self.lib.sd_workspace_bytes(self._h, int(shape[0]), int(shape[1]), int(shape[2]))
        if n == 0:
            raise ValueError('sd_workspace_bytes: ' + self.lib.sd_last_error().decode())
        return int(n)

    @property
    def _ws(self) -> Optional[torch.Tensor]:
        return self._ws_slots.get(0)

    def _workspace(self, shape, slot: int = 0, batch: int = 1) -> torch.Tensor:
        need = self.workspace_bytes(shape) * batch
        ws = self._ws_slots.get(slot)
        if ws is None or ws.numel() < need:
            self._ws_slots.pop(slot, None)
            ws = None
            # torch.cuda.OutOfMemoryError is a RuntimeError -> the reference's tile-halving loop keeps working
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_slots[slot] = ws
        return ws

    # -- forward --------------------------------------------------------------------------------------
    def forward(self, inp: torch.Tensor, out_kind: int = L.SD_OUT_PROBS_F32, out: Optional[torch.Tensor] = None,
                slot: int = 0):
        """inp: (D,H,W) uint8 (normalised in-kernel as float32(v)/255) or float32, on this device.
        Returns (C,D,H,W) float32 (logits / probabilities) or uint8 (floor(255*p)).  Launches on the CURRENT torch
        stream; forwards that may overlap (different streams) must use different workspace `slot`s."""
        assert inp.is_cuda and inp.dim() == 3 and inp.is_contiguous()
        D, H, W = inp.shape
        if self._roi is not None:
            self._set_roi(None)
        ws = self._workspace((D, H, W), slot)
        odt = torch.uint8 if out_kind == L.SD_OUT_PROBS_U8 else torch.float32
        if out is None:
            out = torch.empty((self.out_channels, D, H, W), dtype=odt, device=self.device)
        assert out.dtype == odt and out.is_contiguous() and out.numel() == self.out_channels * D * H * W
        L.check(self.lib.sd_forward(self._h, inp.data_ptr(), _dtype_code(inp), D, H, W, out.data_ptr(), out_kind,
                                    ws.data_ptr(), ws.numel(), _stream()), 'sd_forward')
        return out

    def forward_batch(self, inp: torch.Tensor, out_kind: int = L.SD_OUT_PROBS_F32, out: Optional[torch.Tensor] = None,
                      slot: int = 0, roi=None):
        """N independent tiles in one set of launches: inp (N,D,H,W) uint8 / float32 -> (N,C,D,H,W).  Identical
        results to N `forward` calls; every kernel sees N times as many blocks (fills the GPU in the small layers)."""
        assert inp.is_cuda and inp.dim() == 4 and inp.is_contiguous()
        N, D, H, W = inp.shape
        if roi != self._roi:
            self._set_roi(roi)
        ws = self._workspace((D, H, W), slot, N)
        odt = torch.uint8 if out_kind == L.SD_OUT_PROBS_U8 else torch.float32
        if out is None:
            out = torch.empty((N, self.out_channels, D, H, W), dtype=odt, device=self.device)
        assert out.dtype == odt and out.is_contiguous() and out.numel() == N * self.out_channels * D * H * W
        L.check(self.lib.sd_forward_batch(self._h, inp.data_ptr(), _dtype_code(inp), N, D, H, W, out.data_ptr(),
                                          out_kind, ws.data_ptr(), ws.numel(), _stream()), 'sd_forward_batch')
        return out

    def forward_labels_batch(self, inp: torch.Tensor, ids: Sequence[int], thresholds: Sequence[float],
                             out: Optional[torch.Tensor] = None, slot: int = 0, roi=None) -> torch.Tensor:
        """N tiles -> (N,D,H,W) uint8 labels: the label rule of dense_predictor (prediction.py:813-833) evaluated in
        the final layer's epilogue on floor(255*softmax); identical to ``postproc_labels(forward_batch(PROBS_U8))``."""
        assert inp.is_cuda and inp.dim() == 4 and inp.is_contiguous()
        N, D, H, W = inp.shape
        if roi != self._roi:
            self._set_roi(roi)
        ws = self._workspace((D, H, W), slot, N)
        if out is None:
            out = torch.empty((N, D, H, W), dtype=torch.uint8, device=self.device)
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == N * D * H * W
        n = len(ids)
        ids_a = (C.c_int32 * n)(*[int(i) for i in ids])
        thr_a = (C.c_double * n)(*[float(t) for t in thresholds])
        L.check(self.lib.sd_forward_labels_batch(self._h, inp.data_ptr(), _dtype_code(inp), N, D, H, W, ids_a, thr_a, n,
                                                 out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                'sd_forward_labels_batch')
        return out

    def overflowed(self) -> bool:
        """fp16 range guard (include/syconn_dense.h: sd_model_overflow): True if a forward pass since the last call stored
        an activation beyond fp16's range (results invalid; 'f16' and 'f16x2').  Synchronises the current stream; always False for
        bf16 / f32."""
        flag = C.c_int(0)
        L.check(self.lib.sd_model_overflow(self._h, _stream(), C.byref(flag)), 'sd_model_overflow')
        return bool(flag.value)

    def check_overflow(self):
        if self.overflowed():
            raise L.ActivationOverflowError(
                'fp16 activation overflow (a stored activation exceeded 65504): the results of this forward pass are invalid; '
                "use act_dtype='bf16' (fp32 exponent range) or 'f32'")

    def read_buffer(self, buf: int) -> torch.Tensor:
        """Activation buffer `buf` of the last forward as float32 (C,d,h,w) -- test support."""
        dims = (C.c_int32 * 4)()
        L.check(self.lib.sd_debug_read_buffer(self._h, buf, self._ws.data_ptr(), None, dims, _stream()))
        out = torch.empty(tuple(int(d) for d in dims), dtype=torch.float32, device=self.device)
        L.check(self.lib.sd_debug_read_buffer(self._h, buf, self._ws.data_ptr(), out.data_ptr(), dims, _stream()))
        return out

    def last_launch_count(self) -> int:
        """Plan ops the last forward executed as launches of their own (fused ops run inside another op's launch)."""
        return int(self.lib.sd_debug_last_launch_count(self._h))

    def op_kernels(self):
        """[(index of the op whose launch computed op i, kernel symbol of that launch)] for the last forward (sd_debug_op_kernel)."""
        out = []
        buf = C.create_string_buffer(160)
        for i in range(self.n_ops):
            e = int(self.lib.sd_debug_op_kernel(self._h, i, buf, 160))
            out.append((e, buf.value.decode()))
        return out

    def profile(self, n_slots: int = 1):
        """Bracket every layer launch with HIP events; forward k records into slot k % n_slots (0 = off)."""
        L.check(self.lib.sd_profile_enable(self._h, int(n_slots)), 'sd_profile_enable')
        self._profile = n_slots

    def profile_read(self, slot: int = 0) -> np.ndarray:
        """Per-layer milliseconds of the forward recorded in `slot` (synchronises the device first)."""
        torch.cuda.synchronize(self.device)
        ms = (C.c_float * self.n_ops)()
        L.check(self.lib.sd_profile_read(self._h, int(slot), ms, self.n_ops), 'sd_profile_read')
        return np.asarray(list(ms), dtype=np.float64)


    def profile_read_clocks(self, slot: int = 0) -> np.ndarray:
        """(n_ops, 4) uint64 stamps of the convolution launches recorded in `slot`: shader cycles / 100 MHz ticks at entry and exit of
        the launch's first workgroup (zeros: the op had no convolution launch of its own)."""
        torch.cuda.synchronize(self.device)
        st = (C.c_uint64 * (4 * self.n_ops))()
        L.check(self.lib.sd_profile_read_clocks(self._h, int(slot), st, self.n_ops), 'sd_profile_read_clocks')
        return np.asarray(list(st), dtype=np.uint64).reshape(self.n_ops, 4)


def probe_mfma_rate(device, n_workgroups: int = 256, waves: int = 8, iters: int = 20000, min_seconds: float = 0.25,
                    random_operands: bool = True):
    """(sustained dense bf16 MFMA TFLOP/s, shader clock in GHz) of this box under a chip-wide pure MFMA loop (`sd_probe_mfma_rate`);
    `random_operands`: pseudo-random fragments that take turns (a convolution's operand toggling) instead of constants."""
    device = torch.device(device)
    lib = require_gpu(device.index or 0)
    tf, ghz = C.c_double(), C.c_double()
    with torch.cuda.device(device):
        L.check(lib.sd_probe_mfma_rate(int(n_workgroups), int(waves), int(iters), float(min_seconds), 1 if random_operands else 0, C.byref(tf), C.byref(ghz),
                                       torch.cuda.current_stream(device).cuda_stream), 'sd_probe_mfma_rate')
    return float(tf.value), float(ghz.value)


class StreamRing:
    """`n` side streams for independent tiles: tile i runs on stream i % n with workspace slot i % n, so the small
    deep-level layers and the launch tails of one tile overlap with the big layers of the next (one tile alone
    cannot fill 256 CUs in every layer).  ``with ring: ... for i: with ring.stream(i): ...`` orders the side
    streams after the caller's stream on entry and the caller's stream after them on exit."""

    def __init__(self, device: torch.device, n: int = 2):
        self.device = device
        self.n = max(1, int(n))
        self._streams = [torch.cuda.Stream(device=device) for _ in range(self.n)] if self.n > 1 else []

    def slot(self, i: int) -> int:
        return i % self.n

    def stream(self, i: int):
        if self.n == 1:
            return torch.cuda.stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self._streams[i % self.n])

    def __enter__(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams:
            s.wait_stream(cur)
        return self

    def __exit__(self, *exc):
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams:
            cur.wait_stream(s)
        return False


# -- tiled_apply helpers on the device (SURVEY.md row P3 / kernels K1, K12, K11) ----------------------------
def tile_gather(vol: torch.Tensor, origin, tile_shape, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    VD, VH, VW = vol.shape
    TD, TH, TW = (int(t) for t in tile_shape)
    if out is None:
        out = torch.empty((TD, TH, TW), dtype=vol.dtype, device=vol.device)
    L.check(lib.sd_tile_gather(vol.data_ptr(), _dtype_code(vol), VD, VH, VW, int(origin[0]), int(origin[1]),
                               int(origin[2]), out.data_ptr(), TD, TH, TW, _stream()), 'sd_tile_gather')
    return out


def tile_scatter(tile: torch.Tensor, crop_lo, keep, vol: torch.Tensor, origin):
    lib = L.load()
    Cn, TD, TH, TW = tile.shape
    _, VD, VH, VW = vol.shape
    assert vol.shape[0] == Cn and vol.dtype == tile.dtype
    L.check(lib.sd_tile_scatter(tile.data_ptr(), _dtype_code(tile), Cn, TD, TH, TW,
                                int(crop_lo[0]), int(crop_lo[1]), int(crop_lo[2]),
                                int(keep[0]), int(keep[1]), int(keep[2]), vol.data_ptr(), VD, VH, VW,
                                int(origin[0]), int(origin[1]), int(origin[2]), _stream()), 'sd_tile_scatter')


def postproc_labels(probs_u8: torch.Tensor, ids: Sequence[int], thresholds: Sequence[float],
                    out_dtype=torch.uint8, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Label rule of /root/reference/syconn/handler/prediction.py:813-833 on the device.
    `thresholds[i]` is the already resolved float threshold of ids[i]."""
    lib = L.load()
    assert probs_u8.dtype == torch.uint8 and probs_u8.is_contiguous()
    Cn = probs_u8.shape[0]
    nvox = probs_u8[0].numel()
    n = len(ids)
    ids_a = (C.c_int32 * n)(*[int(i) for i in ids])
    thr_a = (C.c_double * n)(*[float(t) for t in thresholds])
    if out is not None:
        out_dtype = out.dtype
        assert out.is_contiguous() and out.numel() == nvox and out_dtype in (torch.uint8, torch.int64)
    if out_dtype == torch.uint8:
        if out is None:
            out = torch.empty(probs_u8.shape[1:], dtype=torch.uint8, device=probs_u8.device)
        code = L.SD_U8
    else:
        if out is None:
            out = torch.empty(probs_u8.shape[1:], dtype=torch.int64, device=probs_u8.device)  # uint64 bit pattern
        code = L.SD_U64
    L.check(lib.sd_postproc_labels(probs_u8.data_ptr(), Cn, nvox, ids_a, thr_a, n, out.data_ptr(), code, _stream()),
            'sd_postproc_labels')
    return out


def downsample2(t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One mag-pyramid level on the device: ``t[::2, ::2, ::2]`` of a contiguous (D,H,W) uint8 (or 8-byte integer)
    tensor -- the order-0 resampling of ``save_raw/save_seg(..., fast_resampling=True)`` (prediction.py:834-843)."""
    lib = L.load()
    assert t.is_cuda and t.dim() == 3 and t.is_contiguous()
    if t.dtype == torch.uint8:
        code = L.SD_U8
    elif t.element_size() == 8 and not t.dtype.is_floating_point:
        code = L.SD_U64
    else:
        raise ValueError(f'downsample2: unsupported dtype {t.dtype}')
    D, H, W = (int(v) for v in t.shape)
    if out is None:
        out = torch.empty(((D + 1) // 2, (H + 1) // 2, (W + 1) // 2), dtype=t.dtype, device=t.device)
    L.check(lib.sd_downsample2(t.data_ptr(), code, D, H, W, out.data_ptr(), _stream()), 'sd_downsample2')
    return out


def mag_pyramid(t: torch.Tensor, levels: int = 3):
    """[t, t[::2,::2,::2], t[::4,::4,::4], ...] computed on the device."""
    out = [t.contiguous()]
    for _ in range(levels - 1):
        out.append(downsample2(out[-1]))
    return out
