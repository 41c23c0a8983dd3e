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
from this image: parity-UNPINNED (oracle/objseg_ref.py).  Gaussian pre-smoothing (`sigmas`, vigra; unused by default) is not offered.
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
