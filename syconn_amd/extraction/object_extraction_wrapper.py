"""Probability maps -> globally unique object segmentation in KnossosDatasets
(/root/reference/syconn/extraction/object_extraction_wrapper.py:23-377: ``calculate_chunk_numbers_for_box``,
``from_probabilities_to_kd``).  The reference runs six batch-job stages that hand h5 files of every
chunk to each other (connected components -> max labels -> unique labels -> stitch list -> merge list -> apply merge list ->
export); here a chunk's label volume is produced on the GPU and stays a device tensor until its stitched uint64 volume is written
into the target KnossosDataset's overlay cubes.  Per-chunk arithmetic: ``object_extraction_steps`` (HIP library, no CPU fallback).
Not reproduced: `transform_func`, the membrane hook on h5 chunk files (`membrane_filename`; `membrane_kd_path` is), `swapdata`."""
from typing import Dict, List, Optional

import numpy as np
import torch

from . import object_extraction_steps as oes
from ..handler.basics import kd_factory


def calculate_chunk_numbers_for_box(cset, offset, size):
    """Chunks of `cset` that intersect the box (`offset`, `size`) -- contract of object_extraction_wrapper.py:23-55: the list of their
    numbers ordered by chunk origin with x slowest and z fastest, and the mapping chunk number -> position in that list.  The box is
    widened to whole chunks first (down at its origin, up at its far side); origins the grid does not hold are left out (a box that
    ends one voxel beyond a dataset whose extent is a multiple of the chunk size names a row of chunks that does not exist).  Unlike
    the reference, `offset` and `size` are not modified."""
    cs = np.asarray(cset.chunk_size, dtype=np.int64)
    lo = np.asarray(offset, dtype=np.int64) // cs * cs
    hi = -(-(np.asarray(offset, dtype=np.int64) + np.asarray(size, dtype=np.int64)) // cs) * cs
    numbers = np.fromiter(cset.chunk_dict.keys(), dtype=np.int64, count=len(cset.chunk_dict))
    origins = np.asarray([cset.chunk_dict[int(n)].coordinates for n in numbers], dtype=np.int64).reshape(-1, 3)
    # the reference steps from the box's widened origin in chunk-size strides: only origins on that lattice can match
    inside = np.all((origins >= lo) & (origins < hi) & ((origins - lo) % cs == 0), axis=1)
    numbers, origins = numbers[inside], origins[inside]
    order = np.lexsort((origins[:, 2], origins[:, 1], origins[:, 0]))
    chunk_list = [int(n) for n in numbers[order]]
    return chunk_list, {n: i for i, n in enumerate(chunk_list)}


def generate_subcell_kd_from_proba(subcell_names: List[str], chunk_size=None, transf_func_kd_overlay=None,
                                   load_cellorganelles_from_kd_overlaycubes: bool = False, cube_of_interest_bb=None,
                                   cube_shape=None, log=None, overwrite=False, **kwargs):
    """object_extraction_wrapper.py:58-150: connected-component segmentation of the sub-cellular structures `subcell_names` (e.g.
    ['mi', 'vc', 'sj']) as KnossosDatasets at ``config.kd_organelle_seg_paths[co]``; sources are the probability-map datasets
    ``config.kd_<co>_path``, thresholds ``config['cell_objects']['probathresholds'][co]``; the chunk grid covers the cell segmentation
    dataset ``config.kd_seg_path`` in `chunk_size` chunks ([512, 512, 512]).  Existing targets need ``overwrite=True``
    (``FileExistsError`` otherwise)."""
    import os
    import shutil
    from .. import global_params
    from ..knossos import ChunkDataset, KnossosDataset
    conf = global_params.config
    if chunk_size is None:
        chunk_size = [512, 512, 512]
    if cube_shape is None:
        cube_shape = (256, 256, 256)
    kd = kd_factory(conf.kd_seg_path)
    if cube_of_interest_bb is None:
        cube_of_interest_bb = [np.zeros(3, dtype=np.int32), np.asarray(kd.boundary)]
    size = np.asarray(cube_of_interest_bb[1]) - np.asarray(cube_of_interest_bb[0]) + 1
    offset = np.asarray(cube_of_interest_bb[0])
    cd_dir = "{}/chunkdatasets/{}/".format(conf.working_dir, "_".join(subcell_names))
    if os.path.isdir(cd_dir):
        if not overwrite:
            raise FileExistsError(f'Could not start generation of sub-cellular objects "{subcell_names}" ChunkDataset because it '
                                  f'already exists at "{cd_dir}" and overwrite was not set to True.')
        shutil.rmtree(cd_dir)
    cd = ChunkDataset()
    cd.initialize(kd, np.asarray(kd.boundary), chunk_size, cd_dir, box_coords=[0, 0, 0], fit_box_size=True)
    prob_kd_path_dict = {co: getattr(conf, 'kd_{}_path'.format(co)) for co in subcell_names}
    prob_threshs = []
    seg_paths = {co: "{}/knossosdatasets/{}_seg/".format(conf.working_dir, co) for co in subcell_names}
    for co in subcell_names:
        prob_threshs.append(conf['cell_objects']["probathresholds"][co])
        path = seg_paths[co]
        if os.path.isdir(path):
            if not overwrite:
                raise FileExistsError(f'Could not start generation of sub-cellular object "{co}" KnossosDataset because it already '
                                      f'exists at "{path}" and overwrite was not set to True.')
            shutil.rmtree(path)
        target_kd = KnossosDataset()
        target_kd._cube_shape = tuple(cube_shape)
        target_kd.initialize_without_conf(path, kd.boundary, np.array(conf['scaling'], dtype=np.float32), kd.experiment_name, mags=[1],
                                          create_pyk_conf=True, create_knossos_conf=False)
    if load_cellorganelles_from_kd_overlaycubes:      # no thresholds needed
        prob_threshs = None
    # (the box of interest restricts the chunks as the reference does: calculate_chunk_numbers_for_box)
    res = from_probabilities_to_kd(seg_paths, cd, "_".join(subcell_names), prob_kd_path_dict=prob_kd_path_dict, thresholds=prob_threshs,
                                   hdf5names=list(subcell_names), size=size, offset=offset,
                                   load_from_kd_overlaycubes=load_cellorganelles_from_kd_overlaycubes,
                                   transf_func_kd_overlay=transf_func_kd_overlay, log=log, **kwargs)
    shutil.rmtree(cd_dir, ignore_errors=True)
    return res


def from_probabilities_to_kd(target_kd_paths: Optional[Dict[str, str]], cset, filename: str, hdf5names: List[str],
                             prob_kd_path_dict: Optional[Dict[str, str]] = None, load_from_kd_overlaycubes: bool = False,
                             transf_func_kd_overlay=None, log=None, overlap="auto", sigmas: Optional[list] = None,
                             thresholds: Optional[list] = None, debug: bool = False, swapdata: bool = False,
                             offset: Optional[np.ndarray] = None, size: Optional[np.ndarray] = None, suffix: str = "",
                             transform_func=None, func_kwargs: Optional[dict] = None, n_cores: Optional[int] = None,
                             overlap_thresh: Optional[int] = 0, stitch_overlap=None, membrane_filename: str = None,
                             membrane_kd_path: str = None, hdf5_name_membrane: str = None, n_chunk_jobs: int = None,
                             device=None, labels_on_device_bytes: int = 128 << 30, morph_ops: Optional[dict] = None,
                             min_seed_vx: Optional[dict] = None, scaling=None):
    """The reference's signature (object_extraction_wrapper.py:153-173).  `target_kd_paths`: name -> already initialised target
    KnossosDataset (``initialize_without_conf``); `cset`: the chunk grid; `prob_kd_path_dict`: name -> KnossosDataset with the
    uint8 probability map; `thresholds` one per name (fractions <= 1 are scaled by 255, :251-253).  Returns a dict with the
    intermediate results the reference pickles next to the ChunkDataset: ``cc_info_list``, ``overlap_info``, ``max_labels``,
    ``stitch_list``, ``merge_dict`` / ``merge_list_dict`` (per name).

    Extensions: `morph_ops` / `min_seed_vx` / `scaling` override ``config['cell_objects']['extract_morph_op']``, ``['min_seed_vx']`` and
    ``config['scaling']`` (the reference reads them from the working directory's config inside its worker).
    `labels_on_device_bytes`: the int32 component labels of all chunks are kept in HBM up to this budget (a 2048 x 2048 x 512
    dataset in 512^3 chunks with three organelles: 29 GB of 288), beyond it in host memory."""
    unsupported = {'transform_func': transform_func, 'membrane_filename': membrane_filename, 'swapdata': swapdata}
    for k, v in unsupported.items():
        if v:
            raise NotImplementedError(f'from_probabilities_to_kd: `{k}` is not part of the dense-prediction consumers built here')
    if prob_kd_path_dict is None:
        raise NotImplementedError('from_probabilities_to_kd: source data inside the ChunkDataset (h5 files) is not built')
    kd_keys = list(prob_kd_path_dict.keys())
    assert len(kd_keys) == len(hdf5names)
    for kd_key in kd_keys:
        assert kd_key in hdf5names
    if size is not None and offset is not None:
        chunk_list, chunk_translator = calculate_chunk_numbers_for_box(cset, offset, size)
    else:
        chunk_list = [ii for ii in range(len(cset.chunk_dict))]
        chunk_translator = {ii: ii for ii in chunk_list}
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    # (the kept int32 labels may take at most half of what is free in HBM right now, whatever budget was asked for)
    labels_on_device_bytes = min(int(labels_on_device_bytes), torch.cuda.mem_get_info(device)[0] // 2)

    # ---- connected components per chunk (object_segmentation), labels kept
    cc_info_list, overlap_info, _, labels = oes.object_segmentation(
        cset, hdf5names, prob_kd_path_dict, thresholds, overlap=overlap, chunk_list=chunk_list, with_properties=False,
        device=device, sigmas=sigmas, keep_labels=True, labels_on_device_bytes=labels_on_device_bytes, morph_ops=morph_ops,
        min_seed_vx=min_seed_vx, scaling=scaling, load_from_kd_overlaycubes=load_from_kd_overlaycubes,
        transf_func_kd_overlay=transf_func_kd_overlay, membrane_kd_path=membrane_kd_path)
    if stitch_overlap is None:
        stitch_overlap = overlap_info[1]
    else:
        stitch_overlap = np.asarray(stitch_overlap)
        overlap_info[1] = stitch_overlap
    if not np.all(stitch_overlap <= overlap_info[0]):
        raise ValueError("Stitch overlap ({}) has to be <= than chunk overlap ({}).".format(overlap_info[1], overlap_info[0]))
    overlap = overlap_info[0]

    # ---- max labels (:296-312)
    nb_cc = {name: np.zeros(len(chunk_list), dtype=np.int64) for name in hdf5names}
    for nb_chunk, name, n in cc_info_list:
        nb_cc[name][chunk_translator[nb_chunk]] = n
    offsets, max_labels = {}, {}
    for name in hdf5names:
        offsets[name], max_labels[name] = oes.label_offsets(nb_cc[name])

    cs = np.asarray(cset.chunk_size, dtype=np.int64)
    box0 = np.asarray(cset.box_coords if cset.box_coords is not None else np.zeros(3), dtype=np.int64)
    grid_pos = {n: tuple(int(v) for v in (np.asarray(cset.chunk_dict[n].coordinates, dtype=np.int64) - box0) // cs) for n in chunk_list}

    def unique(n, name):                                   # make_unique_labels, on the fly (the uint64 volume is transient)
        return oes.make_unique_labels(labels[(n, name)].to(device), int(offsets[name][chunk_translator[n]]))

    stitch_list, merge_dict, merge_list_dict = {}, {}, {}
    targets = {name: kd_factory(target_kd_paths[name]) for name in hdf5names} if target_kd_paths else {}
    for name in hdf5names:
        # ---- stitch list (:330-337): every chunk against its +x, +y, +z neighbours.  A chunk's unique-label volume is made ONCE and
        # dropped at the end of its iteration; what waits for a later neighbour is only the face slab that neighbour shares with it
        # (2 * stitch_overlap planes): at most one row of slabs per dimension is alive, whatever the size of the dataset.
        by_pos = {p: n for n, p in grid_pos.items()}
        # (`overlap_thresh > 0`, object_extraction_steps.py:597-615: a pair is kept only if the two objects coincide in more than 10 % of
        # their voxels -- then the slab that waits is everything the two chunk volumes share, 2 * overlap planes, and with it the
        # chunk's object sizes)
        waiting, pairs = {}, set()                      # (chunk, dim) -> its +dim face slab, until the +dim neighbour has been made
        width = overlap if overlap_thresh else stitch_overlap
        sizes_of = {}
        for n in sorted(chunk_list, key=lambda k: grid_pos[k]):      # (any -d neighbour then precedes its +d neighbour)
            p = grid_pos[n]
            vol = unique(n, name)
            if overlap_thresh:
                sizes_of[n] = oes.object_sizes(vol)
            for d in range(3):
                before = by_pos.get(tuple(p[k] - (1 if k == d else 0) for k in range(3)))
                if before is not None and (before, d) in waiting:
                    mine = oes.face_slab(vol, d, False, overlap, width)
                    if overlap_thresh:
                        pairs |= oes.overlapping_pairs(waiting.pop((before, d)), mine, d, overlap, stitch_overlap, sizes_of[before], sizes_of[n])
                    else:
                        pairs |= oes.slab_pairs(waiting.pop((before, d)), mine)
                if by_pos.get(tuple(p[k] + (1 if k == d else 0) for k in range(3))) is not None:
                    waiting[(n, d)] = oes.face_slab(vol, d, True, overlap, width)
            del vol
            for m in [k for k in sizes_of if not any((k, d) in waiting for d in range(3)) and k != n]:
                del sizes_of[m]                     # (no neighbour is waiting for this chunk any more)
        assert not waiting, 'chunk order: a +neighbour was processed before its -neighbour'
        stitch_list[name] = sorted(pairs)
        # ---- merge list (:343-347) and its application + export (:352-366)
        merge_dict[name], merge_list_dict[name] = oes.make_merge_list(stitch_list[name], max_labels[name])
        lut = torch.from_numpy(merge_list_dict[name].view(np.int64)).to(device)
        for n in chunk_list:
            chunk = cset.chunk_dict[n]
            stitched = oes.apply_merge_list(unique(n, name), chunk.size, lut)                # (x,y,z) uint64, chunk.size
            if name in targets:
                zyx = stitched.permute(2, 1, 0).contiguous().cpu().numpy().view(np.uint64)
                targets[name].save_seg(offset=[int(v) for v in chunk.coordinates], mags=[1], data=zyx, data_mag=1)
        del lut
    for kd in targets.values():
        if hasattr(kd, 'flush'):
            kd.flush()
    return {'cc_info_list': cc_info_list, 'overlap_info': overlap_info, 'max_labels': max_labels, 'stitch_list': stitch_list,
            'merge_dict': merge_dict, 'merge_list_dict': merge_list_dict, 'chunk_list': chunk_list}
