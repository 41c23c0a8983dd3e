"""Device implementation of the first stage of SyConn's probability-map -> object segmentation
(/root/reference/syconn/extraction/object_extraction_steps.py:204-366, ``_object_segmentation_thread``): threshold ->
morphological operations -> connected components of one chunk (:316-317, :354-358), and -- when the operation list contains
``'binary_erosion'``, which is what the default config selects for mi / sj / vc (config.yml:130-136) -- the watershed branch
(:319-352): erosion seeds -> ``scipy.ndimage.label`` -> ``min_seed_vx`` filter with the reference's id hole filling ->
anisotropic Euclidean distance transform -> marker-based priority flood inside the mask.  Morphology follows /root/reference/syconn/proc/image.py (``apply_morphological_operations`` :485-507,
``_multi_mop_findobjects`` :357-438, ``get_aniso_struct`` :522-539); labels are numbered exactly like
``scipy.ndimage.label``.  Compute is ``sd_object_segmentation`` of the HIP library; there is no CPU fallback.

Parity: everything up to and including the relabelled marker volume is scipy / numpy in the reference and is reproduced bit
for bit (goldens from the reference's own code, tests/golden/g9_objseg.npz, g10_objseg_ws.npz).  The distance transform
(vigra) and the flood (skimage) restate the published algorithms of packages that are absent from the reference tree and
from this image: parity-UNPINNED (oracle/objseg_ref.py), like the optional Gaussian pre-smoothing (`sigmas`, vigra; SyConn's own
pipeline passes none): `gaussian_threshold` / `sd_gaussian_threshold`.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib as L

_MOPS = {'binary_opening': L.SD_MOP_OPENING, 'binary_closing': L.SD_MOP_CLOSING, 'binary_dilation': L.SD_MOP_DILATION,
         'binary_erosion': L.SD_MOP_EROSION}


def get_aniso_struct(scaling: Union[tuple, np.ndarray]) -> np.ndarray:
    """image.py:522-539: kernel for the morphology operations, cross-like with anisotropic dilation in the xy plane;
    (5, 5, 3) in (x, y, z)."""
    aniso = int(scaling[2] // scaling[0])
    assert scaling[1] // scaling[0] == 1
    assert aniso >= 1
    struct = np.zeros((5, 5, 3), dtype=bool)
    struct[2, 2, :] = True
    for dx in range(-2, 3):
        for dy in range(-2, 3):
            if abs(dx) + abs(dy) <= aniso:
                struct[2 + dx, 2 + dy, 1] = True
    return struct


def _count_subsequent_mops(mops: Sequence[str]) -> Tuple[List[str], List[int]]:
    """image.py:510-519: runs of the same operation become one operation with `iterations`."""
    names, counts = [], []
    for m in mops:
        if names and names[-1] == m:
            counts[-1] += 1
        else:
            names.append(m)
            counts.append(1)
    return names, counts


def object_segmentation_first_stage(prob, threshold: float, morph_ops: Sequence[str] = (), scaling=(10, 10, 20),
                                    structure: Optional[np.ndarray] = None, return_mask: bool = False, device=None,
                                    return_device: bool = False, min_seed_vx: int = 0, return_markers: bool = False):
    """One probability map of a chunk -> ``(labels int32 (x,y,z), max_label)`` (+ the binary volume after the morphology if
    `return_mask`, + the relabelled watershed markers if `return_markers`).  `prob`: uint8 (x,y,z) numpy array or device
    tensor; `threshold` in uint8 units as the reference compares it (``tmp_data > threshold``; 0: `prob` is already a 0/1
    mask); `morph_ops`: names of ``scipy.ndimage`` binary operations as in ``config['cell_objects']['extract_morph_op']``;
    `structure` defaults to ``get_aniso_struct(scaling)``.  With ``'binary_erosion'`` in `morph_ops` the watershed branch
    runs (object_extraction_steps.py:319-352): `min_seed_vx` = ``config['cell_objects']['min_seed_vx'][name]``, `scaling` is
    the pixel pitch of the distance transform, `return_mask` returns tmp_data (the mask the flood is confined to).
    `return_device`: leave the results on the GPU (int32 label tensor, 1-element count tensor[, uint8 mask][, int32 markers])
    -- the label volume is 4 bytes per voxel, and its consumer (``find_object_properties`` / ``segstats``, which take device
    tensors) does not need it on the host."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise RuntimeError('syconn_amd: no MI355X visible to PyTorch-ROCm; this package has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    L.check(lib.sd_init(device.index or 0), 'sd_init')
    morph_ops = list(morph_ops)
    for m in morph_ops:
        if m not in _MOPS:
            raise NotImplementedError(f"Only erosion or dilation allowed. Attempted to use morphological operation '{m}'.")
    watershed = 'binary_erosion' in morph_ops
    first_erosion_ix = morph_ops.index('binary_erosion') if watershed else len(morph_ops)      # :320
    pre, seed = morph_ops[:first_erosion_ix], morph_ops[first_erosion_ix:]
    names, counts = _count_subsequent_mops(pre) if pre else ([], [])
    snames, scounts = _count_subsequent_mops(seed) if seed else ([], [])
    if isinstance(prob, np.ndarray):
        if prob.dtype != np.uint8:
            raise TypeError('probability maps are uint8 (KnossosDataset raw data)')
        p = torch.from_numpy(np.ascontiguousarray(prob)).to(device)
    else:
        if prob.dtype != torch.uint8:
            raise TypeError('probability maps are uint8 (KnossosDataset raw data)')
        p = prob.to(device).contiguous()
    if p.dim() != 3:
        raise ValueError('expected a 3D (x, y, z) volume')
    X, Y, Z = (int(v) for v in p.shape)
    st = np.ascontiguousarray(get_aniso_struct(np.asarray(scaling)) if structure is None else structure).astype(np.uint8)
    pmax = max([c for n, c in zip(names + snames, counts + scounts) if n in ('binary_closing', 'binary_dilation')], default=0)
    labels = torch.empty((X, Y, Z), dtype=torch.int32, device=device)
    max_label = torch.zeros(1, dtype=torch.int32, device=device)
    mask = torch.empty((X, Y, Z), dtype=torch.uint8, device=device) if return_mask else None
    markers = torch.empty((X, Y, Z), dtype=torch.int32, device=device) if (return_markers and watershed) else None
    n = len(names)
    ops_a = (C.c_int32 * max(n, 1))(*[_MOPS[m] for m in names])
    it_a = (C.c_int32 * max(n, 1))(*counts)
    stream = torch.cuda.current_stream(device).cuda_stream
    if watershed:
        ws_bytes = lib.sd_objseg_watershed_workspace_bytes(X, Y, Z, pmax)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        ns = len(snames)
        sops_a = (C.c_int32 * ns)(*[_MOPS[m] for m in snames])
        sit_a = (C.c_int32 * ns)(*scounts)
        pitch = (C.c_int32 * 3)(*[int(np.uint32(v)) for v in np.asarray(scaling)[:3]])          # scaling.astype(np.uint32), :350
        L.check(lib.sd_object_segmentation_watershed(
            p.data_ptr(), X, Y, Z, float(threshold), ops_a, it_a, n, sops_a, sit_a, ns, st.ctypes.data_as(C.c_void_p),
            *[int(s) for s in st.shape], int(min_seed_vx), pitch, labels.data_ptr(), max_label.data_ptr(),
            markers.data_ptr() if markers is not None else None, None, mask.data_ptr() if mask is not None else None,
            ws.data_ptr(), ws_bytes, stream), 'sd_object_segmentation_watershed')
    else:
        ws_bytes = lib.sd_objseg_workspace_bytes(X, Y, Z, pmax)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        L.check(lib.sd_object_segmentation(p.data_ptr(), X, Y, Z, float(threshold), ops_a, it_a, n,
                                           st.ctypes.data_as(C.c_void_p), *[int(s) for s in st.shape], labels.data_ptr(),
                                           max_label.data_ptr(), mask.data_ptr() if mask is not None else None, ws.data_ptr(),
                                           ws_bytes, stream), 'sd_object_segmentation')
    extra = ((mask,) if return_mask else ()) + ((markers,) if markers is not None else ())
    if return_device:
        return (labels, max_label) + extra
    return (labels.cpu().numpy(), int(max_label.item())) + tuple(e.cpu().numpy() for e in extra)


def gaussian_threshold(prob, sigma, threshold: float, device=None, return_device: bool = False, return_smoothed: bool = False):
    """``gaussianSmoothing(tmp_data, sigma)`` followed by ``tmp_data > threshold`` (object_extraction_steps.py:296-297, 316-317)
    for one uint8 (x,y,z) probability map: `sigma` a number or one per axis (x,y,z) in voxels, `threshold` in uint8 units ->
    uint8 0/1 mask (x,y,z) [+ the float32 smoothed map].  vigra's algorithm restated (`sd_gaussian_threshold`, parity unpinned)."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise RuntimeError('syconn_amd: no MI355X visible to PyTorch-ROCm; this package has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    L.check(lib.sd_init(device.index or 0), 'sd_init')
    p = torch.from_numpy(np.ascontiguousarray(prob)) if isinstance(prob, np.ndarray) else prob
    if p.dtype != torch.uint8:
        raise TypeError('probability maps are uint8 (KnossosDataset raw data)')
    p = p.to(device).contiguous()
    if p.dim() != 3:
        raise ValueError('expected a 3D (x, y, z) volume')
    sg = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (3,)) if np.ndim(sigma) == 0 else np.asarray(sigma, dtype=np.float64)
    if sg.shape != (3,) or np.any(sg < 0) or not np.all(np.isfinite(sg)):
        raise ValueError('sigma: one non-negative number, or one per axis (x, y, z)')
    X, Y, Z = (int(v) for v in p.shape)
    mask = torch.empty((X, Y, Z), dtype=torch.uint8, device=device)
    sm = torch.empty((X, Y, Z), dtype=torch.float32, device=device) if return_smoothed else None
    ws_bytes = lib.sd_gauss_workspace_bytes(X, Y, Z)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    sg_a = (C.c_double * 3)(*[float(v) for v in sg])
    L.check(lib.sd_gaussian_threshold(p.data_ptr(), X, Y, Z, sg_a, float(threshold), mask.data_ptr(),
                                      sm.data_ptr() if sm is not None else None, ws.data_ptr(), ws_bytes,
                                      torch.cuda.current_stream(device).cuda_stream), 'sd_gaussian_threshold')
    out = (mask,) + ((sm,) if sm is not None else ())
    if not return_device:
        out = tuple(o.cpu().numpy() for o in out)
    return out if len(out) > 1 else out[0]


def marker_flood(d2, markers, mask, device=None, return_device: bool = False):
    """``skimage.segmentation.watershed(-distance, markers, mask=mask)`` as `_object_segmentation_thread` calls it
    (object_extraction_steps.py:351) for ``distance ** 2 == d2`` (int32 >= 0): (x, y, z) arrays or device tensors -> int32 labels
    and the largest label (`sd_marker_flood`)."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise RuntimeError('syconn_amd: no MI355X visible to PyTorch-ROCm; this package has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    L.check(lib.sd_init(device.index or 0), 'sd_init')

    def dev(a, dt):
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        return t.to(device=device, dtype=dt).contiguous()
    g, mk, m = dev(d2, torch.int32), dev(markers, torch.int32), dev(mask, torch.uint8)
    if g.dim() != 3 or g.shape != mk.shape or g.shape != m.shape:
        raise ValueError('expected three 3D (x, y, z) volumes of one shape')
    X, Y, Z = (int(v) for v in g.shape)
    labels = torch.empty((X, Y, Z), dtype=torch.int32, device=device)
    max_label = torch.zeros(1, dtype=torch.int32, device=device)
    ws_bytes = lib.sd_objseg_watershed_workspace_bytes(X, Y, Z, 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    L.check(lib.sd_marker_flood(g.data_ptr(), mk.data_ptr(), m.data_ptr(), X, Y, Z, labels.data_ptr(), max_label.data_ptr(),
                                ws.data_ptr(), ws_bytes, torch.cuda.current_stream(device).cuda_stream), 'sd_marker_flood')
    if return_device:
        return labels, max_label
    return labels.cpu().numpy(), int(max_label.item())


# ---------------------------------------------------------------------------------------------------------------------------
# chunk driver (thin: no job machinery, no h5 files)
def auto_overlap(morph_ops: dict, scaling, sigmas=None) -> np.ndarray:
    """The ``overlap == "auto"`` rule of ``object_segmentation`` (object_extraction_steps.py:152-166): 4 sigma of the largest
    Gaussian (none by default), at least twice the number of erosions laterally enlarged by the anisotropy."""
    max_sigma = np.zeros(3) if sigmas is None else np.array([np.max(sigmas)] * 3)
    overlap = np.ceil(max_sigma * 4)
    aniso = scaling[2] // scaling[0]
    n_erosions = 0
    for v in morph_ops.values():
        n_erosions = max(n_erosions, 2 * aniso * int(np.sum(np.array(v) == 'binary_erosion')))
    return np.max([overlap, [n_erosions, n_erosions, n_erosions // aniso]], axis=0).astype(np.int32)


def object_segmentation(cset, hdf5names: Sequence[str], prob_kd_path_dict: dict, thresholds: Sequence[float],
                        overlap="auto", chunk_list: Optional[Sequence[int]] = None, morph_ops: Optional[dict] = None,
                        min_seed_vx: Optional[dict] = None, scaling=None, with_properties: bool = True, device=None,
                        sigmas=None, keep_labels: bool = False, labels_on_device_bytes: int = 128 << 30,
                        load_from_kd_overlaycubes: bool = False, transf_func_kd_overlay: Optional[dict] = None,
                        membrane_kd_path: Optional[str] = None):
    """``object_segmentation`` + ``_object_segmentation_thread`` (object_extraction_steps.py:42-201, 204-366) for the branch
    SyConn's pipeline takes after the dense prediction (object_extraction_wrapper.py:58-150: probability maps in
    KnossosDatasets ``prob_kd_path_dict``, ``load_raw``): per chunk of `cset` load size + 2 * overlap around the chunk from every
    dataset (x,y,z like ``kd.load_raw(...).swapaxes(0, 2)``), threshold (uint8 scale; fractions <= 1 are scaled by 255 as in
    ``from_probabilities_to_kd``, :251-253), apply ``config['cell_objects']['extract_morph_op'][name]`` and label.

    Returns ``(results, [overlap, stitch_overlap], props)``: `results` = the reference's ``[chunk.number, hdf5_name,
    max_label]`` rows; `props[(chunk.number, hdf5_name)]`` = ``find_object_properties`` of that label volume (rep_coords,
    bounding_box, sizes in chunk-local (x,y,z) incl. the overlap margin) computed from the label volume while it is STILL ON THE
    DEVICE -- the int32 labels (4 bytes per voxel) never cross PCIe; the reference writes them to an h5 file per chunk and reads
    them back for the statistics.  `sigmas` (object_extraction_steps.py:77-81, 135-138, 296-298: a vigra ``gaussianSmoothing`` of the
    probability map before the threshold, one sigma or (x,y,z) triple per name; SyConn's pipeline passes none): maps with a
    non-zero sigma are smoothed and thresholded on the device (`gaussian_threshold`; vigra's algorithm restated, parity
    unpinned).  `keep_labels`: a fourth result ``labels[(chunk.number, hdf5_name)]`` = the int32 (x,y,z) label volume incl. the
    overlap margin (what the reference writes to ``*_connected_components.h5``), a device tensor while all of them fit in
    `labels_on_device_bytes`, else a host tensor -- the input of the stitching steps (``from_probabilities_to_kd``).
    `load_from_kd_overlaycubes` (object_extraction_steps.py:254-270): the source datasets hold SEGMENTATION (overlay cubes, ``load_seg``)
    instead of probability maps; ``transf_func_kd_overlay[name]`` (a callable on the (x,y,z) uint64 array) is applied if given, no
    threshold is (:316), every non-zero voxel is foreground.  `membrane_kd_path` (:309-314, experimental in the reference): for the
    names 'p4' and 'vc' voxels whose membrane probability exceeds ``255 * .4`` are cleared before the threshold.
    Not reproduced: the membrane hook on h5 chunk files (`membrane_filename`), `swapdata`, source data inside the ChunkDataset."""
    from .. import global_params
    from ..knossos import KnossosDataset
    from .find_object_properties import find_object_properties
    conf = global_params.config
    morph_ops = conf['cell_objects']['extract_morph_op'] if morph_ops is None else morph_ops
    min_seed_vx = conf['cell_objects']['min_seed_vx'] if min_seed_vx is None else min_seed_vx
    scaling = np.array(conf['scaling'] if scaling is None else scaling)
    if sigmas is not None:
        if len(sigmas) != len(hdf5names):
            raise Exception("Number of thresholds, sigmas and HDF5 names does not match!")      # (the reference's check, :137-139)
    if isinstance(overlap, str) and overlap == "auto":
        overlap = auto_overlap(morph_ops, scaling, sigmas)
    overlap = np.asarray(overlap, dtype=np.int64)
    stitch_overlap = np.max([overlap.copy(), [1, 1, 1]], axis=0)
    thresholds = np.zeros(len(hdf5names)) if thresholds is None else np.array(thresholds, dtype=np.float64)
    if len(thresholds) and thresholds[0] <= 1.:
        thresholds = thresholds * 255
    from ..handler.basics import kd_factory
    kds = {k: kd_factory(path) for k, path in prob_kd_path_dict.items()}
    kd_bar = kd_factory(membrane_kd_path) if membrane_kd_path is not None else None
    chunk_ids = list(cset.chunk_dict.keys()) if chunk_list is None else list(chunk_list)
    results, props, kept, kept_bytes = [], {}, {}, 0
    for nb in chunk_ids:
        chunk = cset.chunk_dict[nb]
        box_offset = np.array(chunk.coordinates) - overlap
        size = np.array(chunk.size) + 2 * overlap
        for i, name in enumerate(hdf5names):
            if load_from_kd_overlaycubes:                                                    # :254-267
                data_k = kds[name].load_seg(size=size, offset=box_offset, mag=1).swapaxes(0, 2)
                if transf_func_kd_overlay is not None:
                    data_k = transf_func_kd_overlay[name](data_k)
                tmp_data = np.ascontiguousarray(np.asarray(data_k) != 0).astype(np.uint8)      # (labelled / eroded as a binary volume)
            else:
                tmp_data = np.ascontiguousarray(kds[name].load_raw(size=size, offset=box_offset, mag=1).swapaxes(0, 2))
            ops = list(morph_ops.get(name, [])) if name in morph_ops else []
            seed = int(min_seed_vx.get(name, 0)) if name in min_seed_vx else 0
            thr = 0.0 if load_from_kd_overlaycubes else float(thresholds[i])                 # :316 (no threshold on overlay input)
            if name in ("p4", "vc") and kd_bar is not None:                                  # :309-314
                if sigmas is not None and float(np.sum(sigmas[i])) != 0.0:
                    raise NotImplementedError('membrane masking behind a Gaussian pre-smoothing is not built')
                membrane_data = kd_bar.load_raw(size=size, offset=box_offset, mag=1).swapaxes(0, 2)
                tmp_data[membrane_data > 255 * .4] = 0
                del membrane_data
            if sigmas is not None and float(np.sum(sigmas[i])) != 0.0:                      # :296-297
                tmp_data = gaussian_threshold(tmp_data, sigmas[i], thr, device=device, return_device=True)
                thr = 0.0                                                                    # (a 0/1 mask from here on)
            labels, max_label = object_segmentation_first_stage(tmp_data, thr, ops, scaling, device=device,
                                                                return_device=True, min_seed_vx=seed)
            if with_properties:
                props[(chunk.number, name)] = find_object_properties(labels)
            results.append([chunk.number, name, int(max_label.item())])
            if keep_labels:
                kept_bytes += labels.numel() * 4
                kept[(chunk.number, name)] = labels if kept_bytes <= labels_on_device_bytes else labels.cpu()
    if keep_labels:
        return results, [overlap, stitch_overlap], props, kept
    return results, [overlap, stitch_overlap], props


# ---------------------------------------------------------------------------------------------------------------------------
# Globally unique objects: the steps of ``from_probabilities_to_kd`` behind the per-chunk first stage
# (/root/reference/syconn/extraction/object_extraction_wrapper.py:296-352; object_extraction_steps.py:369-736).  The reference keeps
# every intermediate in one h5 file per chunk and step; here a chunk's label volume stays a device tensor from the connected
# components to the stitched uint64 volume.  Pinned by tests/golden/g11_stitch.npz (the reference's own thread functions, executed
# with in-memory stand-ins for their h5 / ChunkDataset I/O).
def label_offsets(n_components: Sequence[int]) -> Tuple[np.ndarray, int]:
    """object_extraction_wrapper.py:300-312: ``max_nb_dict`` -- what is added to the labels of chunk i -- is the number of components
    of all chunks before it (in chunk-list order); ``max_labels`` = the total.  Returns (offsets int64[n], max_label)."""
    nb = np.asarray(n_components, dtype=np.int64)
    off = np.zeros(len(nb), dtype=np.int64)
    if len(nb) > 1:
        off[1:] = np.cumsum(nb[:-1])
    return off, int(off[-1] + nb[-1]) if len(nb) else 0


def make_unique_labels(labels: torch.Tensor, offset: int) -> torch.Tensor:
    """``_make_unique_labels_thread`` (object_extraction_steps.py:425-443) for one chunk: int32 (x,y,z) component labels on the
    device -> uint64 (stored as int64 bits) with `offset` added to every non-zero label (`sd_labels_make_unique`)."""
    lib = L.load()
    if not (labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous()):
        raise TypeError('expected a contiguous int32 device tensor (the labels of object_segmentation_first_stage(return_device=True))')
    out = torch.empty(labels.shape, dtype=torch.int64, device=labels.device)
    with torch.cuda.device(labels.device):
        L.check(lib.sd_labels_make_unique(labels.data_ptr(), labels.numel(), int(offset), out.data_ptr(),
                                          torch.cuda.current_stream(labels.device).cuda_stream), 'sd_labels_make_unique')
    return out


def labels_box(vol: torch.Tensor, lo, size, lut: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Contiguous copy of the box ``vol[lo : lo + size]`` of a uint64 (int64 bits) (x,y,z) device volume, optionally mapped through a
    look-up table (`sd_labels_box_lut`); an id beyond the table raises."""
    lib = L.load()
    if not (vol.is_cuda and vol.dtype == torch.int64 and vol.is_contiguous() and vol.dim() == 3):
        raise TypeError('expected a contiguous 3D int64 / uint64 device tensor')
    lo, size = [int(v) for v in lo], [int(v) for v in size]
    out = torch.empty(size, dtype=torch.int64, device=vol.device)
    status = torch.zeros(1, dtype=torch.int32, device=vol.device) if lut is not None else None
    with torch.cuda.device(vol.device):
        L.check(lib.sd_labels_box_lut(vol.data_ptr(), *[int(v) for v in vol.shape], *lo, *size,
                                      lut.data_ptr() if lut is not None else None, lut.numel() if lut is not None else 0,
                                      out.data_ptr(), status.data_ptr() if status is not None else None,
                                      torch.cuda.current_stream(vol.device).cuda_stream), 'sd_labels_box_lut')
    if status is not None and int(status.item()):
        raise ValueError('apply_merge_list: a label exceeds the merge list (max_label too small)')
    return out


def stitch_pairs(chunk_a: torch.Tensor, chunk_b: torch.Tensor, dim: int, overlap, stitch_overlap, overlap_thresh: float = 0) -> set:
    """The inner part of ``_make_stitch_list_thread`` (object_extraction_steps.py:555-615) for one chunk and its neighbour in +`dim`:
    both are unique-label volumes of size chunk + 2 * overlap; the slab of `stitch_overlap` voxels on either side of the common
    chunk face is ``a[-overlap - stitch : -overlap + stitch]`` in the first and ``b[overlap - stitch : overlap + stitch]`` in the
    second (``cut_array_in_one_dim``, proc/general.py:45-82) -- the same voxels, labelled twice.  Returns the set of
    ``tuple(sorted((id_a, id_b)))`` over the voxels where both are non-zero; with ``overlap_thresh > 0`` (:597-615) only the pairs whose
    objects coincide in more than 10 % of their voxels (`overlapping_pairs`).  The co-occurrence table is the device native behind
    ``map_subcell_C`` (`sd_segstats_scan`: one pass over the two slabs)."""
    if any(int(chunk_a.shape[d]) != int(chunk_b.shape[d]) for d in range(3) if d != dim):
        raise ValueError('neighbouring chunks differ in the face extents')
    if not overlap_thresh:
        return slab_pairs(face_slab(chunk_a, dim, True, overlap, stitch_overlap), face_slab(chunk_b, dim, False, overlap, stitch_overlap))
    return overlapping_pairs(face_slab(chunk_a, dim, True, overlap, overlap), face_slab(chunk_b, dim, False, overlap, overlap), dim, overlap,
                             stitch_overlap, object_sizes(chunk_a), object_sizes(chunk_b))


def face_slab(chunk: torch.Tensor, dim: int, high: bool, overlap, width) -> torch.Tensor:
    """The ``2 * width[dim]`` planes around the chunk face in +`dim` (`high`: the slab this chunk shares with its +`dim` neighbour,
    ``[-overlap - width, -overlap + width)``) or in -`dim` (``[overlap - width, overlap + width)``), as a contiguous copy.
    ``width = stitch_overlap``: all ``stitch_pairs`` reads of a chunk (so a chunk's volume need not outlive the loop iteration that
    made it); ``width = overlap``: everything the chunk shares with that neighbour (the ``overlap_thresh`` test)."""
    ol, so = int(overlap[dim]), int(width[dim])
    if so > ol or so < 1:
        raise ValueError('stitch overlap has to be >= 1 and <= the chunk overlap')
    n = int(chunk.shape[dim])
    lo, size = [0, 0, 0], [int(v) for v in chunk.shape]
    lo[dim], size[dim] = (n - ol - so) if high else (ol - so), 2 * so
    if lo[dim] < 0 or lo[dim] + 2 * so > n:
        raise ValueError('chunk smaller than its overlap')
    return labels_box(chunk, lo, size)


def slab_pairs(slab_a: torch.Tensor, slab_b: torch.Tensor, counts: bool = False):
    """``tuple(sorted((id_a, id_b)))`` over the voxels where both slabs (the same voxels, labelled by two chunks) are non-zero;
    `counts`: ``{(id_a, id_b): voxels}`` instead (ids in argument order)."""
    from .find_object_properties import segstats
    if tuple(slab_a.shape) != tuple(slab_b.shape):
        raise ValueError('neighbouring chunks differ in the face extents')
    r = segstats(slab_a, [slab_b], want_props=False)
    ids_b, ids_a, n = r.pairs[0]
    if counts:
        return {(int(x), int(y)): int(c) for x, y, c in zip(ids_a.tolist(), ids_b.tolist(), n.tolist())}
    return {(int(min(x, y)), int(max(x, y))) for x, y in zip(ids_a.tolist(), ids_b.tolist())}


def object_sizes(vol: torch.Tensor) -> dict:
    """id -> voxels of a unique-label volume (``len(np.nonzero(cc_data == id))`` of object_extraction_steps.py:598-600, for every id)."""
    from .find_object_properties import segstats
    ids, _, size, _ = segstats(vol).cell
    return dict(zip(ids.tolist(), size.tolist()))


def overlapping_pairs(wide_a: torch.Tensor, wide_b: torch.Tensor, dim: int, overlap, stitch_overlap, sizes_a: dict, sizes_b: dict) -> set:
    """``overlap_thresh > 0`` (object_extraction_steps.py:597-615).  The reference takes every pair of ids that touch inside the stitch
    slab, builds a cKDTree over the GLOBAL voxel coordinates of the one object and asks for every voxel of the other whether its
    distance is zero: ``match_vx`` = voxels where the first chunk carries the one id and the second chunk the other, over everything
    the two chunk volumes share (their ``2 * overlap`` planes around the face); the pair is kept iff ``2 * match_vx / (size_a + size_b)``
    exceeds 0.1 (object sizes inside the chunk volumes, the constant is the reference's).  Here: one co-occurrence pass over the two
    ``2 * overlap`` slabs (`wide_a` / `wide_b` = ``face_slab(..., width=overlap)``) gives every ``match_vx`` at once, the candidates
    are the pairs of the stitch slab in its middle."""
    ol, so = int(overlap[dim]), int(stitch_overlap[dim])
    if so > ol or so < 1:
        raise ValueError('stitch overlap has to be >= 1 and <= the chunk overlap')
    matches = slab_pairs(wide_a, wide_b, counts=True)
    if so == ol:
        cands = set(matches)
    else:
        lo, size = [0, 0, 0], [int(v) for v in wide_a.shape]
        lo[dim], size[dim] = ol - so, 2 * so
        cands = set(slab_pairs(labels_box(wide_a, lo, size), labels_box(wide_b, lo, size), counts=True))
    keep = set()
    for a, b in cands:
        if 2.0 * float(matches[(a, b)]) / (sizes_a[a] + sizes_b[b]) > 0.1:
            keep.add((min(a, b), max(a, b)))
    return keep


def make_stitch_list(chunks: dict, grid_pos: dict, overlap, stitch_overlap, overlap_thresh: float = 0) -> list:
    """``make_stitch_list`` / ``_make_stitch_list_thread`` (object_extraction_steps.py:446-617) for one label name: `chunks` maps a
    chunk number to its unique-label device volume, `grid_pos` a chunk number to its (ix, iy, iz) position in the chunk grid; every
    chunk is compared with its neighbours in +x, +y, +z (the upper half of the 6-neighbourhood, :548-554).  Returns the list of
    id pairs that belong to one object."""
    by_pos = {tuple(int(v) for v in p): n for n, p in grid_pos.items() if n in chunks}
    pairs = set()
    for n, vol in chunks.items():
        p = tuple(int(v) for v in grid_pos[n])
        for dim in range(3):
            q = list(p)
            q[dim] += 1
            m = by_pos.get(tuple(q))
            if m is not None:
                pairs |= stitch_pairs(vol, chunks[m], dim, overlap, stitch_overlap, overlap_thresh)
    return sorted(pairs)


def make_merge_list(stitch_list: Sequence, max_label: int) -> Tuple[dict, np.ndarray]:
    """``make_merge_list`` (object_extraction_steps.py:620-655) for one label name: ids connected through `stitch_list` pairs collapse
    to ONE id of their component.  The reference takes ``list(component)[0]`` of networkx's set -- an arbitrary member; here it is the
    smallest id (same partition, canonical representative).  Returns (merge_dict: id -> representative for every id that occurs in
    a pair, merge_list: uint64[max_label + 1] with ``merge_list[id]`` = representative, identity elsewhere).  Union-find on the
    host: the list has one entry per pair of touching objects, not per voxel."""
    ml = np.arange(int(max_label) + 1, dtype=np.uint64)
    if len(stitch_list) == 0:
        return {}, ml
    pr = np.asarray(stitch_list, dtype=np.int64).reshape(-1, 2)
    if pr.min() < 1 or pr.max() > max_label:
        raise ValueError('stitch list holds ids outside 1..max_label')
    parent = np.arange(int(max_label) + 1, dtype=np.int64)

    def find(a):
        r = a
        while parent[r] != r:
            r = parent[r]
        while parent[a] != r:
            parent[a], a = r, parent[a]
        return r
    for a, b in pr.tolist():
        ra, rb = find(a), find(b)
        if ra != rb:
            if ra < rb:
                parent[rb] = ra
            else:
                parent[ra] = rb
    ids = np.unique(pr)
    merge_dict = {int(i): int(find(int(i))) for i in ids}
    for i, r in merge_dict.items():
        ml[i] = r
    return merge_dict, ml


def apply_merge_list(chunk_vol: torch.Tensor, chunk_size, merge_list) -> torch.Tensor:
    """``_apply_merge_list_thread`` (object_extraction_steps.py:717-731) for one chunk: crop the overlap margin
    (``offset = (shape - chunk.size) // 2``) and map every id through the merge list; uint64 (int64 bits) device tensor of
    ``chunk_size``."""
    shape = np.asarray(chunk_vol.shape, dtype=np.int64)
    size = np.asarray(chunk_size, dtype=np.int64)
    off = (shape - size) // 2
    lut = merge_list if isinstance(merge_list, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(merge_list).view(np.int64)).to(chunk_vol.device)
    return labels_box(chunk_vol, off, shape - 2 * off, lut)
