"""Dataset-wide object properties and organelle -> cell overlap counts from KnossosDataset segmentations: the chunk driver of
``map_subcell_extract_props`` (/root/reference/syconn/proc/sd_proc.py:273-787) up to and including
the dictionaries its first step produces -- per chunk the Cython native ``map_subcell_extract_props``
(find_object_properties_C.pyx:112-192; here the device pass ``sd_segstats_scan``), then ``merge_prop_dicts`` (:1248-1273) /
``merge_map_dicts`` (:1300-1322) into per-dataset dictionaries.  Not built: the worker / batch-job machinery, the mesh caches and
step 2 (writing SegmentationObject storages) -- they belong to SyConn's storage layer, outside the dense-prediction path.

Pinned: ``merge_prop_dicts``, ``merge_map_dicts``, ``convert_nvox2ratio_mapdict``, ``invert_mdc`` by the reference's own functions
(tests/golden/make_golden_propmerge.py, AST-lifted), the per-chunk native by the reference's known-answer test (g8)."""
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

from ..handler.basics import kd_factory


def merge_prop_dicts(prop_dicts: List[List[dict]], offset: Optional[np.ndarray] = None):
    """sd_proc.py:1248-1273: merge ``[rep_coords, bounding_boxes, sizes]`` triples IN PLACE into the first one.  Representative
    coordinates: a later chunk overwrites an earlier one; bounding boxes: every chunk's box is appended to the id's list (the
    first triple's second dict must be a ``defaultdict(list)``); sizes add up.  `offset` (the chunk's origin) is added to the
    coordinates and boxes of every triple but the first."""
    tot_rc, tot_bb, tot_size = prop_dicts[0][0], prop_dicts[0][1], prop_dicts[0][2]
    for el in prop_dicts[1:]:
        if len(el[0]) == 0:
            continue
        if offset is not None:
            for k in el[0]:
                el[0][k] = [el[0][k][ii] + offset[ii] for ii in range(3)]
        tot_rc.update(el[0])
        for k, v in el[1].items():
            bb = v if offset is None else [[v[0][ii] + offset[ii] for ii in range(3)], [v[1][ii] + offset[ii] for ii in range(3)]]
            tot_bb[k].append(bb)
        for k, v in el[2].items():
            if k in tot_size:
                tot_size[k] += v
            else:
                tot_size[k] = v


def merge_map_dicts(map_dicts):
    """sd_proc.py:1300-1322: merge ``subcell id -> cell id -> overlap voxels`` dictionaries IN PLACE into the first one (counts add
    up; an id that is new to the first dictionary brings its inner dictionary along, not a copy)."""
    tot_map = map_dicts[0]
    for el in map_dicts[1:]:
        for sc_id, sc_dc in el.items():
            if sc_id in tot_map:
                for cellsv_id, ol_vx_cnt in sc_dc.items():
                    if cellsv_id in tot_map[sc_id]:
                        tot_map[sc_id][cellsv_id] += ol_vx_cnt
                    else:
                        tot_map[sc_id][cellsv_id] = ol_vx_cnt
            else:
                tot_map[sc_id] = sc_dc


def convert_nvox2ratio_mapdict(map_dc):
    """sd_proc.py:1276-1285: overlap voxel counts -> fractions of each subcellular object's mapped voxels, in place."""
    for subcell_id, subcell_dc in map_dc.items():
        s = np.sum(list(subcell_dc.values()))
        for k, v in subcell_dc.items():
            map_dc[subcell_id][k] = subcell_dc[k] / s


def invert_mdc(mapping_dict):
    """sd_proc.py:1288-1297: ``subcell id -> cell id -> value`` turned into ``cell id -> subcell id -> value``."""
    mdc_inv = {}
    for subcell_id, subcell_dc in mapping_dict.items():
        for cell_id, v in subcell_dc.items():
            if cell_id not in mdc_inv:
                mdc_inv[cell_id] = {subcell_id: v}
            else:
                mdc_inv[cell_id][subcell_id] = v
    return mdc_inv


def _boundary_ids(vol) -> np.ndarray:
    """Ids on the six faces of an (x,y,z) device volume: objects that are "not purely inside this chunk" (sd_proc.py:626-629)."""
    import torch
    faces = [vol[0], vol[-1], vol[:, 0], vol[:, -1], vol[:, :, 0], vol[:, :, -1]]
    return torch.unique(torch.cat([f.reshape(-1) for f in faces])).cpu().numpy().view(np.uint64) if vol.dtype == torch.int64 \
        else torch.unique(torch.cat([f.reshape(-1) for f in faces])).cpu().numpy()


def map_subcell_extract_props(kd_seg_path: str, kd_organelle_paths: Dict[str, str], n_folders_fs: int = 1000,
                              n_folders_fs_sc: int = 1000, n_chunk_jobs: Optional[int] = None, n_cores: int = 1,
                              cube_of_interest_bb: Optional[Sequence] = None, chunk_size: Optional[Sequence[int]] = None,
                              log=None, overwrite=False, min_obj_vx: Optional[dict] = None, device=None):
    """Step 1 of the reference's function of this name (sd_proc.py:273-787; the per-chunk loop is
    ``_map_subcell_extract_props_thread``, :617-678): over a regular chunk grid (``fit_box_size=True``) load the cell segmentation
    and every organelle segmentation (``load_seg(...).swapaxes(0, 2)``), extract per-chunk properties and overlap counts on the
    GPU, drop objects that lie purely inside a chunk and are smaller than ``config['cell_objects']['min_obj_vx'][name]`` (for
    organelles also from the overlap dictionary), and merge everything with the chunk's origin added.

    Returns ``(cell_props, organelle_props, organelle_maps)``: ``[rc, bb, size]`` of the cell segmentation, ``{name: [rc, bb, size]}``
    and ``{name: {subcell id: {cell id: voxels}}}`` -- the dictionaries the reference's workers pickle for step 2.
    The segmentations go to the device as they are loaded (8 B per voxel and volume over PCIe); the label volumes are read once."""
    import torch
    from .. import global_params
    from ..extraction.find_object_properties import map_subcell_extract_props as native
    from ..knossos import ChunkDataset
    kd = kd_factory(kd_seg_path)
    kd_subcells = {k: kd_factory(v) for k, v in kd_organelle_paths.items()}
    for k, kd_sc in kd_subcells.items():
        if not np.array_equal(kd_sc.boundary, kd.boundary):
            raise ValueError("Data shape of subcellular structures '{}' differs from cell segmentation data. {} vs. {}".format(
                k, kd_sc.boundary, kd.boundary))
    if min_obj_vx is None:
        min_obj_vx = global_params.config['cell_objects']['min_obj_vx']
    if chunk_size is None:
        chunk_size = [512, 512, 512]
    chunk_size = np.asarray(chunk_size, dtype=np.int64)
    if cube_of_interest_bb is None:
        cube_of_interest_bb = [np.zeros(3, dtype=np.int64), np.asarray(kd.boundary, dtype=np.int64)]
    size = np.asarray(cube_of_interest_bb[1]) - np.asarray(cube_of_interest_bb[0])
    offset0 = np.asarray(cube_of_interest_bb[0], dtype=np.int64)
    cd = ChunkDataset()
    cd.initialize(kd, size, chunk_size, '', box_coords=offset0, fit_box_size=True)
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    names = list(kd_organelle_paths.keys())
    n_subcell = len(names)
    cpd_lst = [{}, defaultdict(list), {}]
    scpd_lst = [[{}, defaultdict(list), {}] for _ in range(n_subcell)]
    scmd_lst = [{} for _ in range(n_subcell)]

    def load(k, offset):
        a = np.ascontiguousarray(k.load_seg(size=chunk_size, offset=offset, mag=1).swapaxes(0, 2))
        return torch.from_numpy(a.view(np.int64)).to(device)
    for ch_id in sorted(cd.chunk_dict):
        offset = np.asarray(cd.chunk_dict[ch_id].coordinates, dtype=np.int64)
        subs = [load(kd_subcells[n], offset) for n in names]
        cell_d = load(kd, offset)
        obj_ids_bdry = {n: _boundary_ids(s) for n, s in zip(names, subs)}
        cell_prop_dicts, subcell_prop_dicts, subcell_mapping_dicts = native(cell_d, subs)
        # objects purely inside this chunk and below the size threshold are dropped (:640-650, :657-670)
        if min_obj_vx.get('sv', 1) > 1:
            inside = set(cell_prop_dicts[0].keys()).difference(set(_boundary_ids(cell_d).tolist()))
            for ix in inside:
                if cell_prop_dicts[2][ix] < min_obj_vx['sv']:
                    del cell_prop_dicts[0][ix], cell_prop_dicts[1][ix], cell_prop_dicts[2][ix]
        merge_prop_dicts([cpd_lst, cell_prop_dicts], offset)
        subcell_prop_dicts = [[subcell_prop_dicts[0][ii], subcell_prop_dicts[1][ii], subcell_prop_dicts[2][ii]]
                              for ii in range(n_subcell)]
        for ii, organelle in enumerate(names):
            if min_obj_vx.get(organelle, 1) > 1:
                inside = set(subcell_prop_dicts[ii][0].keys()).difference(set(obj_ids_bdry[organelle].tolist()))
                for ix in inside:
                    if subcell_prop_dicts[ii][2][ix] < min_obj_vx[organelle]:
                        del subcell_prop_dicts[ii][0][ix], subcell_prop_dicts[ii][1][ix]
                        del subcell_prop_dicts[ii][2][ix]
                        if ix in subcell_mapping_dicts[ii]:
                            del subcell_mapping_dicts[ii][ix]
            merge_map_dicts([scmd_lst[ii], subcell_mapping_dicts[ii]])
            merge_prop_dicts([scpd_lst[ii], subcell_prop_dicts[ii]], offset)
    return cpd_lst, {n: scpd_lst[i] for i, n in enumerate(names)}, {n: scmd_lst[i] for i, n in enumerate(names)}
