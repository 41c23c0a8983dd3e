"""Dataset-wide object properties and organelle -> cell overlap counts from KnossosDataset segmentations: step 1 of
``map_subcell_extract_props`` (/root/reference/syconn/proc/sd_proc.py:273-787).

MI355X shape of the work.  The reference turns every chunk into Python dictionaries and folds them into running dictionaries, id
by id, on the host (its worker loop :617-678 with ``merge_prop_dicts`` :1248-1273 and ``merge_map_dicts`` :1300-1322).  Here no
per-object data exists on the host before the dataset is finished:

* per chunk  ``sd_segstats_scan`` (one HBM-bound pass over the label volumes) fills device hash tables; ``sd_chunkprops_append`` /
  ``sd_chunkpairs_append`` turn them into RECORDS appended to device arrays -- applying the "purely inside the chunk and smaller than
  ``min_obj_vx``" filter from the tables' own bounding boxes (an id is on a face of the chunk exactly when its box touches that
  face: no sixfold ``np.unique`` over the faces) and adding the chunk origin;
* per dataset  ``sd_propmerge_objects`` / ``sd_propmerge_pairs``: stable radix sort by id + one segment per id (sizes add up, the
  representative coordinate is the last chunk's, bounding boxes stay one per chunk in chunk order);
* at the API edge the merged tables (``PropTable`` / ``MapTable``, plain numpy) become the dictionaries the reference's workers
  pickle for step 2.  ``convert_nvox2ratio_mapdict`` / ``invert_mdc`` have array forms there (``MapTable.ratios`` /
  ``MapTable.inverted``).

The four dictionary functions of the reference's module are kept by name and contract (in-place merge into the first element)
for callers that hold dictionaries; the chunk driver does not use them.  Not built: the batch-job machinery, the mesh caches and
step 2 (writing SegmentationObject storages) -- SyConn's storage layer, outside the dense-prediction path.

Pinned by tests/golden/g12_propmerge.npz (outputs of the reference's own functions, tests/golden/make_golden_propmerge.py)."""
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from ..handler.basics import kd_factory


# ---------------------------------------------------------------------------------------------------------------------------------
# merged tables (host, numpy) and their dictionary views
class PropTable:
    """Merged properties of one label volume kind: ``ids`` ascending (uint64), ``sizes`` (int64), ``rep_coords`` (n, 3),
    ``boxes`` (m, 2, 3) = every (chunk, id) box in id-major / chunk-minor order, ``box_begin`` (n + 1) offsets into ``boxes``."""

    def __init__(self, ids, sizes, rep_coords, boxes, box_begin):
        self.ids, self.sizes, self.rep_coords, self.boxes, self.box_begin = ids, sizes, rep_coords, boxes, box_begin

    def __len__(self):
        return len(self.ids)

    def as_dicts(self) -> List[dict]:
        """``[rep_coords, bounding_boxes, sizes]`` as the reference's workers return them: id -> [x, y, z], id -> list of
        [[min], [max + 1]] (one entry per chunk holding the id), id -> voxels."""
        keys = self.ids.tolist()
        per_id = np.split(self.boxes, self.box_begin[1:-1]) if len(keys) else []
        return [dict(zip(keys, self.rep_coords.tolist())), dict(zip(keys, (b.tolist() for b in per_id))),
                dict(zip(keys, self.sizes.tolist()))]


class MapTable:
    """Merged overlap counts of one organelle: rows (subcell id, cell id, voxels), ascending by subcell id, then cell id."""

    def __init__(self, sub_ids, cell_ids, counts):
        self.sub_ids, self.cell_ids, self.counts = sub_ids, cell_ids, counts

    def __len__(self):
        return len(self.sub_ids)

    @staticmethod
    def _nested(outer, inner, values) -> Dict[int, Dict[int, object]]:
        out: Dict[int, Dict[int, object]] = {}
        if len(outer) == 0:
            return out
        cut = np.flatnonzero(outer[1:] != outer[:-1]) + 1
        inner_l, values_l = inner.tolist(), values.tolist()
        for a, b in zip(np.concatenate(([0], cut)).tolist(), np.concatenate((cut, [len(outer)])).tolist()):
            out[int(outer[a])] = dict(zip(inner_l[a:b], values_l[a:b]))
        return out

    def as_dict(self) -> Dict[int, Dict[int, int]]:
        """subcell id -> cell id -> overlapping voxels."""
        return self._nested(self.sub_ids, self.cell_ids, self.counts)

    def ratios(self) -> np.ndarray:
        """Array form of ``convert_nvox2ratio_mapdict`` (sd_proc.py:1276-1285): every count divided by the mapped voxels of its
        subcellular object (float64, ``count / sum`` like the reference's division by a numpy integer)."""
        if len(self) == 0:
            return np.zeros(0, np.float64)
        heads = np.flatnonzero(np.concatenate(([True], self.sub_ids[1:] != self.sub_ids[:-1])))
        totals = np.add.reduceat(self.counts.astype(np.int64), heads)
        return self.counts / np.repeat(totals, np.diff(np.concatenate((heads, [len(self)]))))

    def inverted(self, values: Optional[np.ndarray] = None) -> Dict[int, Dict[int, object]]:
        """Array form of ``invert_mdc`` (sd_proc.py:1288-1297): cell id -> subcell id -> value (the counts, or e.g. ``ratios()``)."""
        values = self.counts if values is None else values
        order = np.lexsort((self.sub_ids, self.cell_ids))
        return self._nested(self.cell_ids[order], self.sub_ids[order], values[order])


# ---------------------------------------------------------------------------------------------------------------------------------
# device-side record accumulators
class _TableOverflow(Exception):
    """A hash table of ``sd_segstats_scan`` was too small for some chunk (seen late: the flags are read two chunks behind)."""

    def __init__(self, objects: bool, pairs: bool):
        super().__init__('segstats table overflow')
        self.objects, self.pairs = objects, pairs


class _Records:
    """Growable device arrays appended to at a device-side cursor.  ``fields`` = [(name, dtype, inner width)]; `cursor`: a 1-element
    view into the merger's counter tensor."""

    def __init__(self, device, fields, capacity: int, cursor):
        import torch
        self.torch, self.device, self.fields = torch, device, fields
        self.capacity = int(capacity)
        self.cursor = cursor
        self.arrays = {n: self._new(dt, w, self.capacity) for n, dt, w in fields}

    def _new(self, dtype, width, n):
        return self.torch.empty((n, width) if width > 1 else (n,), dtype=dtype, device=self.device)

    def ptrs(self):
        return [self.arrays[n].data_ptr() for n, _, _ in self.fields]

    def room_for(self, stored: int, n_more: int):
        """`stored` records are known to be in the arrays; make sure `n_more` further ones fit."""
        need = int(stored) + int(n_more)
        if need <= self.capacity:
            return
        cap = max(need, 2 * self.capacity)
        for n, dt, w in self.fields:
            grown = self._new(dt, w, cap)
            keep = min(int(stored) + int(n_more), self.capacity)      # (everything that may have been written so far)
            grown[:keep] = self.arrays[n][:keep]
            self.arrays[n] = grown
        self.capacity = cap


class ChunkMerger:
    """Accumulates the per-chunk tables of one dataset pass on the device and merges them at the end.

    The host never waits for the chunk it has just queued: the append cursors and the overflow flags of the scans are copied to
    page-locked memory on a side stream after every chunk, and chunk k reads the copy posted after chunk k - 2 (complete by then,
    while the GPU still works on chunk k - 1).  Record arrays are kept large enough for what is known to be stored plus the
    pessimistic bound (a full table) of the two chunks in between."""
    LAG = 2

    def __init__(self, names: Sequence[str], min_obj_vx: dict, device, n_chunks: int, capacity: int = 1 << 16):
        import torch
        from .. import _lib as L
        self.L, self.lib, self.torch = L, L.load(), torch
        self.device = torch.device(device)
        self.names = list(names)
        self.min_vx = {k: int(min_obj_vx.get(k, 1)) for k in ['sv'] + self.names}
        obj = [('ids', torch.int64, 1), ('rc', torch.int32, 3), ('bb', torch.int32, 6), ('sizes', torch.int64, 1)]
        pair = [('sub', torch.int64, 1), ('cell', torch.int64, 1), ('cnt', torch.int64, 1)]
        n = len(self.names)
        self.counters = torch.zeros(1 + 2 * n, dtype=torch.int64, device=self.device)          # cursors: cell, subs, pairs
        self.status = torch.zeros((max(int(n_chunks), 1), 2), dtype=torch.int32, device=self.device)      # overflow flags of every chunk's scan
        self.cell = _Records(self.device, obj, capacity, self.counters[0:1])
        self.sub = [_Records(self.device, obj, capacity, self.counters[1 + i:2 + i]) for i in range(n)]
        self.pairs = [_Records(self.device, pair, capacity, self.counters[1 + n + i:2 + n + i]) for i in range(n)]
        self.n_chunks = 0
        self._side = torch.cuda.Stream(device=self.device)
        ring = self.LAG + 1
        self._host_counts = [torch.zeros(1 + 2 * n, dtype=torch.int64).pin_memory() for _ in range(ring)]
        self._host_status = [torch.zeros_like(self.status, device='cpu').pin_memory() for _ in range(ring)]
        self._posted = [torch.cuda.Event() for _ in range(ring)]
        self._known = np.zeros(1 + 2 * n, dtype=np.int64)      # cursor values as of the newest copy that has been read
        self._known_at = -1                                    # ... posted after this chunk

    def status_slot(self):
        """Where the scan of the next chunk writes its overflow flags (``DeviceScan.scan(status_out=...)``)."""
        return self.status[self.n_chunks]

    def _post(self, k: int):
        r = k % (self.LAG + 1)
        ev = self.torch.cuda.current_stream(self.device).record_event()
        with self.torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            self._host_counts[r].copy_(self.counters, non_blocking=True)
            self._host_status[r].copy_(self.status, non_blocking=True)
            self._posted[r].record(self._side)

    def _read(self, k: int):
        """Take in the copy posted after chunk k (waits for it); raises when a scan up to chunk k overflowed its tables."""
        if k <= self._known_at or k < 0:
            return
        r = k % (self.LAG + 1)
        self._posted[r].synchronize()
        self._known = self._host_counts[r].numpy().copy()
        self._known_at = k
        flags = self._host_status[r].numpy()[:k + 1]
        if flags.any():
            raise _TableOverflow(bool(flags[:, 0].any()), bool(flags[:, 1].any()))

    def add_chunk(self, scan, origin):
        """`scan`: a ``DeviceScan`` after ``scan(cell, subs, status_out=self.status_slot())`` over one chunk whose (x, y, z) origin in
        the dataset is `origin`."""
        lib, L = self.lib, self.L
        k = self.n_chunks
        X, Y, Z = scan.shape
        ox, oy, oz = (int(v) for v in origin)
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        cap_o, cap_p = scan.cap_obj, scan.cap_pair
        upper_o, upper_p = min(cap_o, X * Y * Z), min(cap_p, X * Y * Z)
        self._read(k - self.LAG)
        behind = k - 1 - self._known_at            # chunks appended since the counts that are known (at most LAG - 1 ... LAG)
        n = len(self.names)

        def settle(rec, idx, upper):
            rec.room_for(int(self._known[idx]) + behind * upper, upper)
        settle(self.cell, 0, upper_o)
        ids, rc, bb, sz = self.cell.ptrs()
        L.check(lib.sd_chunkprops_append(scan.cell_table.data_ptr(), cap_o, X, Y, Z, ox, oy, oz, self.min_vx['sv'], ids, rc, bb, sz,
                                         self.cell.capacity, self.cell.cursor.data_ptr(), stream), 'sd_chunkprops_append')
        for i, name in enumerate(self.names):
            tab = scan.sub_tables[i]
            settle(self.sub[i], 1 + i, upper_o)
            ids, rc, bb, sz = self.sub[i].ptrs()
            L.check(lib.sd_chunkprops_append(tab.data_ptr(), cap_o, X, Y, Z, ox, oy, oz, self.min_vx[name], ids, rc, bb, sz,
                                             self.sub[i].capacity, self.sub[i].cursor.data_ptr(), stream), 'sd_chunkprops_append')
            settle(self.pairs[i], 1 + n + i, upper_p)
            a, b, c = self.pairs[i].ptrs()
            L.check(lib.sd_chunkpairs_append(scan.ptabs[i].data_ptr(), cap_p, tab.data_ptr(), scan.cell_table.data_ptr(), cap_o, X, Y, Z,
                                             self.min_vx[name], a, b, c, self.pairs[i].capacity, self.pairs[i].cursor.data_ptr(), stream),
                    'sd_chunkpairs_append')
        self._post(k)
        self.n_chunks += 1

    # -- end of the dataset ------------------------------------------------------------------------------------------------------
    def _scratch(self, n: int):
        """One scratch buffer for all merges of `finish` (sized for the largest record count)."""
        need = self.lib.sd_propmerge_temp_bytes(max(int(n), 1))
        if getattr(self, '_tmp', None) is None or self._tmp.numel() < need:
            self._tmp = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
        return self._tmp

    def _merge_objects(self, rec: _Records, n: int) -> PropTable:
        torch, lib = self.torch, self.lib
        assert n <= rec.capacity, 'record arrays overran (internal error: capacity bound)'
        if n == 0:
            return PropTable(np.zeros(0, np.uint64), np.zeros(0, np.int64), np.zeros((0, 3), np.int64), np.zeros((0, 2, 3), np.int64),
                             np.zeros(1, np.int64))
        dev, stream = self.device, torch.cuda.current_stream(self.device).cuda_stream
        uniq = torch.empty(n, dtype=torch.int64, device=dev)
        tot = torch.empty(n, dtype=torch.int64, device=dev)
        rc = torch.empty((n, 3), dtype=torch.int32, device=dev)
        beg = torch.empty(n, dtype=torch.int32, device=dev)
        bbs = torch.empty((n, 6), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        tmp = self._scratch(n)
        a = rec.arrays
        self.L.check(lib.sd_propmerge_objects(a['ids'].data_ptr(), a['sizes'].data_ptr(), a['rc'].data_ptr(), a['bb'].data_ptr(), n,
                                              uniq.data_ptr(), tot.data_ptr(), rc.data_ptr(), beg.data_ptr(), bbs.data_ptr(),
                                              cnt.data_ptr(), tmp.data_ptr(), tmp.numel(), stream), 'sd_propmerge_objects')
        u = int(cnt.item())
        begin = np.concatenate((beg[:u].cpu().numpy().view(np.uint32).astype(np.int64), [n]))
        return PropTable(uniq[:u].cpu().numpy().view(np.uint64), tot[:u].cpu().numpy(), rc[:u].cpu().numpy().astype(np.int64),
                         bbs.cpu().numpy().astype(np.int64).reshape(n, 2, 3), begin)

    def _merge_pairs(self, rec: _Records, n: int) -> MapTable:
        torch, lib = self.torch, self.lib
        assert n <= rec.capacity, 'record arrays overran (internal error: capacity bound)'
        if n == 0:
            return MapTable(np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.int64))
        dev, stream = self.device, torch.cuda.current_stream(self.device).cuda_stream
        o_s, o_c, o_n = (torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3))
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        tmp = self._scratch(n)
        a = rec.arrays
        self.L.check(lib.sd_propmerge_pairs(a['sub'].data_ptr(), a['cell'].data_ptr(), a['cnt'].data_ptr(), n, o_s.data_ptr(),
                                            o_c.data_ptr(), o_n.data_ptr(), cnt.data_ptr(), tmp.data_ptr(), tmp.numel(), stream),
                     'sd_propmerge_pairs')
        u = int(cnt.item())
        return MapTable(o_s[:u].cpu().numpy().view(np.uint64), o_c[:u].cpu().numpy().view(np.uint64), o_n[:u].cpu().numpy())

    def finish(self):
        """-> (cell PropTable, {name: PropTable}, {name: MapTable})"""
        self._read(self.n_chunks - 1)              # every chunk's flags and the final cursors
        counts = [int(v) for v in self._known]
        n = len(self.names)
        self._scratch(max(counts) if counts else 1)
        return (self._merge_objects(self.cell, counts[0]), {nm: self._merge_objects(self.sub[i], counts[1 + i]) for i, nm in enumerate(self.names)},
                {nm: self._merge_pairs(self.pairs[i], counts[1 + n + i]) for i, nm in enumerate(self.names)})


# ---------------------------------------------------------------------------------------------------------------------------------
def map_subcell_extract_props(kd_seg_path: str, kd_organelle_paths: Dict[str, str], n_folders_fs: int = 1000,
                              n_folders_fs_sc: int = 1000, n_chunk_jobs: Optional[int] = None, n_cores: int = 1,
                              cube_of_interest_bb: Optional[Sequence] = None, chunk_size: Optional[Sequence[int]] = None,
                              log=None, overwrite=False, min_obj_vx: Optional[dict] = None, device=None, as_tables: bool = False,
                              chunk_loader: Optional[Callable] = None, table_capacity: Optional[int] = None):
    """Step 1 of the reference's function of this name (sd_proc.py:273-787): over a regular chunk grid (``fit_box_size=True``)
    read the cell segmentation and every organelle segmentation (``load_seg(...).swapaxes(0, 2)``, zeros beyond the dataset),
    gather per-object properties and organelle -> cell overlap counts, drop objects that lie purely inside a chunk and are
    smaller than ``config['cell_objects']['min_obj_vx'][name]``, and merge everything in dataset coordinates.

    Returns ``(cell_props, organelle_props, organelle_maps)``: ``[rc, bb, size]`` of the cell segmentation, ``{name: [rc, bb, size]}``
    and ``{name: {subcell id: {cell id: voxels}}}`` -- the dictionaries the reference's workers pickle for step 2; with
    ``as_tables=True`` the merged ``PropTable`` / ``MapTable`` objects instead (no per-object Python objects are built).
    ``chunk_loader(name, offset_xyz, size_xyz)`` (name ``'sv'`` = cell segmentation) may supply (x, y, z) label volumes, host or
    device, in the place of the KnossosDataset reads."""
    import torch
    from .. import global_params
    from ..extraction.find_object_properties import DeviceScan
    from ..knossos import ChunkDataset
    names = list(kd_organelle_paths.keys())
    kd = kd_factory(kd_seg_path)
    if chunk_loader is None:
        kds = {'sv': kd}
        for name, path in kd_organelle_paths.items():
            kds[name] = kd_factory(path)
            if not np.array_equal(kds[name].boundary, kd.boundary):
                raise ValueError("Data shape of subcellular structures '{}' differs from cell segmentation data. {} vs. {}".format(
                    name, kds[name].boundary, kd.boundary))

        def chunk_loader(name, offset, size):
            return np.ascontiguousarray(kds[name].load_seg(size=size, offset=offset, mag=1).swapaxes(0, 2))
    if min_obj_vx is None:
        min_obj_vx = global_params.config['cell_objects']['min_obj_vx']
    chunk_size = np.asarray([512, 512, 512] if chunk_size is None else chunk_size, dtype=np.int64)
    lo = np.zeros(3, dtype=np.int64) if cube_of_interest_bb is None else np.asarray(cube_of_interest_bb[0], dtype=np.int64)
    hi = np.asarray(kd.boundary, dtype=np.int64) if cube_of_interest_bb is None else np.asarray(cube_of_interest_bb[1], dtype=np.int64)
    cd = ChunkDataset()
    cd.initialize(kd, hi - lo, chunk_size, '', box_coords=lo, fit_box_size=True)
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    chunk_ids = sorted(cd.chunk_dict)
    nvox = int(np.prod(chunk_size))
    # table capacities: label volumes hold far fewer objects than voxels; a chunk that needs more is detected (two chunks late, the
    # host does not wait for the chunk it has just queued) and the pass is repeated with 4x the capacity
    cap_obj = cap_pair = 1 << max(10, int((table_capacity or min(2 * nvox, max(1 << 16, nvox // 256))) - 1).bit_length())
    with torch.cuda.device(device):
        while True:
            scan = DeviceScan(device, cap_obj, cap_pair)
            merger = ChunkMerger(names, min_obj_vx, device, len(chunk_ids))
            try:
                for ch_id in chunk_ids:
                    origin = np.asarray(cd.chunk_dict[ch_id].coordinates, dtype=np.int64)
                    subs = [chunk_loader(n, origin, chunk_size) for n in names]
                    scan.scan(chunk_loader('sv', origin, chunk_size), subs, status_out=merger.status_slot())
                    merger.add_chunk(scan, origin)
                cell_t, sub_t, map_t = merger.finish()
                break
            except _TableOverflow as e:
                if e.objects:
                    if cap_obj >= 2 * nvox:
                        raise RuntimeError('sd_segstats_scan: object table overflow at maximum capacity')
                    cap_obj *= 4
                if e.pairs:
                    cap_pair *= 4
    if as_tables:
        return cell_t, sub_t, map_t
    return cell_t.as_dicts(), {n: t.as_dicts() for n, t in sub_t.items()}, {n: t.as_dict() for n, t in map_t.items()}


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's dictionary helpers, kept by name and contract for callers that hold dictionaries (the driver above works on
# tables).  Each merges IN PLACE into the first element, as sd_proc.py:1248-1322 do.
def merge_prop_dicts(prop_dicts: List[List[dict]], offset: Optional[np.ndarray] = None):
    """``prop_dicts[0]`` = running ``[rep_coords, bounding_boxes, sizes]`` (its second entry maps id -> list of boxes); every further
    triple is one chunk: its coordinates and boxes are shifted by `offset`, its representative coordinates replace earlier ones,
    its boxes are appended, its sizes added.  A triple without objects is skipped."""
    into_rc, into_bb, into_sz = prop_dicts[0]
    shift = np.zeros(3, dtype=np.int64) if offset is None else np.asarray(offset)
    for rc, bb, sz in prop_dicts[1:]:
        if not rc:
            continue
        keys = list(bb)
        boxes = (np.asarray([bb[k] for k in keys]).reshape(len(keys), 2, 3) + shift).tolist()
        rc_keys = list(rc)
        into_rc.update(zip(rc_keys, (np.asarray([rc[k] for k in rc_keys]).reshape(len(rc_keys), 3) + shift).tolist()))
        for k, box in zip(keys, boxes):
            into_bb.setdefault(k, []).append(box)
        for k, n in sz.items():
            into_sz[k] = into_sz.get(k, 0) + n


def merge_map_dicts(map_dicts):
    """``map_dicts[0]`` = running ``subcell id -> cell id -> voxels``; the others are added to it (an id new to the running
    dictionary contributes its inner dictionary itself, not a copy)."""
    into = map_dicts[0]
    for chunk_map in map_dicts[1:]:
        for sub_id, cells in chunk_map.items():
            known = into.get(sub_id)
            if known is None:
                into[sub_id] = cells
                continue
            for cell_id, n in cells.items():
                known[cell_id] = known.get(cell_id, 0) + n


def convert_nvox2ratio_mapdict(map_dc):
    """Overlap voxel counts -> fractions of each subcellular object's mapped voxels, in place (float64 true division)."""
    for cells in map_dc.values():
        counts = np.fromiter(cells.values(), dtype=np.int64, count=len(cells))
        cells.update(zip(list(cells), counts / counts.sum()))


def invert_mdc(mapping_dict):
    """``subcell id -> cell id -> value`` -> ``cell id -> subcell id -> value``."""
    out: Dict[int, dict] = {}
    for sub_id, cells in mapping_dict.items():
        for cell_id, value in cells.items():
            out.setdefault(cell_id, {})[sub_id] = value
    return out
