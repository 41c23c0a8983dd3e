"""Drop-in for the Cython natives re-exported by /root/reference/syconn/extraction/find_object_properties.py:9-11
(implemented in find_object_properties_C.pyx): label-volume statistics of segmentation chunks, computed on the MI355X
by one streaming pass into device hash tables (``include/syconn_dense.h``: ``sd_segstats_*``).

Same call signatures and return structure as the reference (plain dicts with list values, as Cython converts the C++
maps): ids are Python ints, coordinates index the (x, y, z) array that was passed in.  Inputs may be numpy arrays
(uint32 / uint64, copied to the device) or torch tensors already on the device (int32 / int64 bit patterns or
torch.uint32 / torch.uint64).  There is no CPU fallback.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L


def _to_device(vol, device) -> Tuple[torch.Tensor, int]:
    """-> (contiguous device tensor viewed as a signed integer type of the same width, SD_U32 | SD_U64)."""
    if isinstance(vol, np.ndarray):
        if vol.dtype == np.uint64:
            t = torch.from_numpy(np.ascontiguousarray(vol).view(np.int64))
        elif vol.dtype == np.uint32:
            t = torch.from_numpy(np.ascontiguousarray(vol).view(np.int32))
        else:
            raise TypeError(f'label volumes must be uint32 or uint64, got {vol.dtype}')   # the fused type n_type of the .pyx
        t = t.to(device)
    else:
        t = vol
        if t.dtype in (torch.uint64, torch.int64):
            t = t.view(torch.int64)
        elif t.dtype in (torch.uint32, torch.int32):
            t = t.view(torch.int32)
        else:
            raise TypeError(f'label volumes must be 32- or 64-bit integers, got {t.dtype}')
        t = t.to(device).contiguous()
    return t, (L.SD_U64 if t.dtype == torch.int64 else L.SD_U32)


def _pow2_at_least(n: int) -> int:
    return 1 << max(10, int(n - 1).bit_length())


class SegStats:
    """Result of one pass, still as dense arrays (ids ascending): ``cell`` / ``sub[i]`` = (ids, first, size, bbox) with
    bbox (n, 2, 3); ``pairs[i]`` = (subcell ids, cell ids, counts)."""

    def __init__(self):
        self.shape = None
        self.cell = None
        self.sub: List[tuple] = []
        self.pairs: List[tuple] = []


class DeviceScan:
    """Hash tables of ``sd_segstats_scan`` kept on the device and reused from chunk to chunk (the scan initialises them itself).
    After ``scan``: ``tabs[0]`` = cell table (when a cell volume was given), ``tabs[k..]`` = subcell tables, ``ptabs[i]`` = pair table
    of subcell volume i; capacities grow (and stay grown) when a pass reports overflow."""

    def __init__(self, device=None, cap_obj: Optional[int] = None, cap_pair: Optional[int] = None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError('syconn_amd: no MI355X visible to PyTorch-ROCm; this package has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        L.check(self.lib.sd_init(self.device.index or 0), 'sd_init')
        self.cap_obj = _pow2_at_least(cap_obj) if cap_obj else 0
        self.cap_pair = _pow2_at_least(cap_pair) if cap_pair else 0
        self.tabs: List[torch.Tensor] = []
        self.ptabs: List[torch.Tensor] = []
        self.status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.shape = None
        self.has_cell = False
        self.n_sub = 0

    def _room(self, n_tabs: int, n_ptabs: int):
        ob, pb = self.lib.sd_objtable_bytes(self.cap_obj), self.lib.sd_pairtable_bytes(self.cap_pair)
        self.tabs = [t for t in self.tabs if t.numel() == ob][:n_tabs]
        self.ptabs = [t for t in self.ptabs if t.numel() == pb][:n_ptabs]
        while len(self.tabs) < n_tabs:
            self.tabs.append(torch.empty(ob, dtype=torch.uint8, device=self.device))
        while len(self.ptabs) < n_ptabs:
            self.ptabs.append(torch.empty(pb, dtype=torch.uint8, device=self.device))

    def scan(self, cell, subs: Sequence = (), want_props: bool = True, status_out: Optional[torch.Tensor] = None):
        """One streaming pass over `cell` (may be None) and the `subs` volumes, all of one (X, Y, Z) shape and dtype.
        `status_out` (int32[2] on the device): the pass writes its overflow flags there and this call does NOT wait for them -- the
        caller reads them later and repeats the work with larger tables (`cap_obj` / `cap_pair` are then the caller's business);
        without it the flags are awaited here and an overflowed pass is repeated with 4x the capacity."""
        lib, device = self.lib, self.device
        vols = [v for v in ([cell] if cell is not None else []) + list(subs)]
        if not vols:
            raise ValueError('no volume given')
        shape = tuple(int(s) for s in vols[0].shape)
        if len(shape) != 3:
            raise ValueError('label volumes must be 3D (x, y, z)')
        for v in vols:
            assert tuple(v.shape) == shape, 'Segmentation of cells and subcellular structures must have same shape.'
        dev = [_to_device(v, device) for v in vols]
        if len({d for _, d in dev}) != 1:
            raise TypeError('all label volumes of one call must share a dtype')
        dtype = dev[0][1]
        cell_t = dev[0][0] if cell is not None else None
        sub_ts = [t for t, _ in (dev[1:] if cell is not None else dev)]
        n_sub = len(sub_ts)
        nvox = shape[0] * shape[1] * shape[2]
        # capacity guess: label volumes hold far fewer objects than voxels; overflow is detected and the pass repeated
        if not self.cap_obj:
            self.cap_obj = _pow2_at_least(min(2 * nvox, max(1 << 16, nvox // 256)))
        if not self.cap_pair:
            self.cap_pair = self.cap_obj
        stream = torch.cuda.current_stream(device).cuda_stream
        while True:
            self._room(1 + n_sub, n_sub if cell_t is not None else 0)
            tabs, ptabs = self.tabs, self.ptabs
            sub_ptrs = (C.c_void_p * max(n_sub, 1))(*[t.data_ptr() for t in sub_ts])
            sub_tabs = (C.c_void_p * max(n_sub, 1))(*[t.data_ptr() for t in tabs[1:]])
            pair_tabs = (C.c_void_p * max(n_sub, 1))(*[t.data_ptr() for t in ptabs])
            L.check(lib.sd_segstats_scan(cell_t.data_ptr() if cell_t is not None else None, sub_ptrs, n_sub, dtype, *shape,
                                         tabs[0].data_ptr() if cell_t is not None else None, sub_tabs, self.cap_obj, pair_tabs,
                                         self.cap_pair, 1 if want_props else 0,
                                         (self.status if status_out is None else status_out).data_ptr(), stream), 'sd_segstats_scan')
            if status_out is not None:
                break
            st = self.status.cpu().tolist()
            if not any(st):
                break
            if st[0]:
                if self.cap_obj >= 2 * nvox:
                    raise RuntimeError('sd_segstats_scan: object table overflow at maximum capacity')
                self.cap_obj *= 4
            if st[1]:
                self.cap_pair *= 4
        self.shape, self.has_cell, self.n_sub = shape, cell_t is not None, n_sub
        return self

    @property
    def cell_table(self):
        return self.tabs[0] if self.has_cell else None

    @property
    def sub_tables(self):
        return self.tabs[1:1 + self.n_sub]


def segstats(cell, subs: Sequence = (), want_props: bool = True, device=None, cap_obj: Optional[int] = None,
             cap_pair: Optional[int] = None) -> SegStats:
    """One streaming pass over `cell` (may be None) and the `subs` volumes, all of one (X, Y, Z) shape and dtype."""
    sc = DeviceScan(device, cap_obj, cap_pair).scan(cell, subs, want_props)
    lib, device, cap_obj, cap_pair, tabs, ptabs = sc.lib, sc.device, sc.cap_obj, sc.cap_pair, sc.tabs, sc.ptabs
    shape, n_sub = sc.shape, sc.n_sub
    cell_t = True if sc.has_cell else None
    stream = torch.cuda.current_stream(device).cuda_stream

    def objects(tab):
        n_max = cap_obj
        cnt = torch.zeros(1, dtype=torch.int64, device=device)
        ids = torch.empty(n_max, dtype=torch.int64, device=device)
        first = torch.empty(n_max, dtype=torch.int64, device=device)
        size = torch.empty(n_max, dtype=torch.int64, device=device)
        bb = torch.empty((n_max, 6), dtype=torch.int32, device=device)
        L.check(lib.sd_segstats_compact_objects(tab.data_ptr(), cap_obj, ids.data_ptr(), first.data_ptr(), size.data_ptr(),
                                                bb.data_ptr(), n_max, cnt.data_ptr(), stream), 'sd_segstats_compact_objects')
        n = int(cnt.item())
        ids_h = ids[:n].cpu().numpy().view(np.uint64)
        order = np.argsort(ids_h, kind='stable')
        return (ids_h[order], first[:n].cpu().numpy()[order], size[:n].cpu().numpy()[order],
                bb[:n].cpu().numpy().reshape(n, 2, 3)[order])

    def pairs(ptab, stab, ctab):
        n_max = cap_pair
        cnt = torch.zeros(1, dtype=torch.int64, device=device)
        a, b, c = (torch.empty(n_max, dtype=torch.int64, device=device) for _ in range(3))
        L.check(lib.sd_segstats_compact_pairs(ptab.data_ptr(), cap_pair, stab.data_ptr(), ctab.data_ptr(), cap_obj,
                                              a.data_ptr(), b.data_ptr(), c.data_ptr(), n_max, cnt.data_ptr(), stream),
                'sd_segstats_compact_pairs')
        n = int(cnt.item())
        s_h, c_h, n_h = (t[:n].cpu().numpy() for t in (a, b, c))
        s_h, c_h = s_h.view(np.uint64), c_h.view(np.uint64)
        order = np.lexsort((c_h, s_h))
        return s_h[order], c_h[order], n_h[order]

    res = SegStats()
    res.shape = shape
    if want_props:
        if cell_t is not None:
            res.cell = objects(tabs[0])
        res.sub = [objects(t) for t in tabs[1:1 + n_sub]]
    if cell_t is not None:
        res.pairs = [pairs(ptabs[i], tabs[1 + i], tabs[0]) for i in range(n_sub)]
    return res


def _prop_dicts(shape, ids, first, size, bb):
    rc = np.stack(np.unravel_index(first, shape), axis=1) if len(ids) else np.zeros((0, 3), np.int64)
    keys = ids.tolist()
    return dict(zip(keys, rc.tolist())), dict(zip(keys, bb.tolist())), dict(zip(keys, size.tolist()))


def _pair_dict(s, c, n) -> Dict[int, Dict[int, int]]:
    d: Dict[int, Dict[int, int]] = {}
    for sk, ck, cnt in zip(s.tolist(), c.tolist(), n.tolist()):
        d.setdefault(sk, {})[ck] = cnt
    return d


def find_object_properties(chunk):
    """find_object_properties_C.pyx:24-49: ``(rep_coords, bounding_box, sizes)`` of the non-zero ids of `chunk` (x,y,z):
    id -> first voxel in raster order, id -> [[min], [max + 1]], id -> voxel count."""
    r = segstats(chunk)
    return _prop_dicts(r.shape, *r.cell)


def map_subcell_extract_props(ch, subcell_chs):
    """find_object_properties_C.pyx:112-192: ``[rc, bb, size]`` of the cell segmentation, ``[[rc...], [bb...], [size...]]``
    per subcellular volume, and per subcellular volume ``subcell id -> cell id -> overlapping voxels``."""
    subs = [subcell_chs[i] for i in range(len(subcell_chs))]
    r = segstats(ch, subs)
    cell = list(_prop_dicts(r.shape, *r.cell))
    sub = [[], [], []]
    for s in r.sub:
        rc, bb, sz = _prop_dicts(r.shape, *s)
        sub[0].append(rc); sub[1].append(bb); sub[2].append(sz)
    return cell, sub, [_pair_dict(*p) for p in r.pairs]


def map_subcell_C(ch, subcell_chs):
    """find_object_properties_C.pyx:72-109: the overlap counts only."""
    subs = [subcell_chs[i] for i in range(len(subcell_chs))]
    r = segstats(ch, subs, want_props=False)
    return [_pair_dict(*p) for p in r.pairs]
