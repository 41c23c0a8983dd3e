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
                        sigmas=None):
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
    unpinned).  Not reproduced: the membrane hooks, `swapdata`, overlay-cube input."""
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
    thresholds = np.array(thresholds, dtype=np.float64)
    if len(thresholds) and thresholds[0] <= 1.:
        thresholds = thresholds * 255
    kds = {}
    for k, path in prob_kd_path_dict.items():
        kds[k] = KnossosDataset()
        kds[k].initialize_from_knossos_path(path)
    chunk_ids = list(cset.chunk_dict.keys()) if chunk_list is None else list(chunk_list)
    results, props = [], {}
    for nb in chunk_ids:
        chunk = cset.chunk_dict[nb]
        box_offset = np.array(chunk.coordinates) - overlap
        size = np.array(chunk.size) + 2 * overlap
        for i, name in enumerate(hdf5names):
            tmp_data = np.ascontiguousarray(kds[name].load_raw(size=size, offset=box_offset, mag=1).swapaxes(0, 2))
            ops = list(morph_ops.get(name, [])) if name in morph_ops else []
            seed = int(min_seed_vx.get(name, 0)) if name in min_seed_vx else 0
            thr = float(thresholds[i])
            if sigmas is not None and float(np.sum(sigmas[i])) != 0.0:                      # :296-297
                tmp_data = gaussian_threshold(tmp_data, sigmas[i], thr, device=device, return_device=True)
                thr = 0.0                                                                    # (a 0/1 mask from here on)
            labels, max_label = object_segmentation_first_stage(tmp_data, thr, ops, scaling, device=device,
                                                                return_device=True, min_seed_vx=seed)
            if with_properties:
                props[(chunk.number, name)] = find_object_properties(labels)
            results.append([chunk.number, name, int(max_label.item())])
    return results, [overlap, stitch_overlap], props
