"""Device implementation of the first stage of SyConn's probability-map -> object segmentation
(/root/reference/syconn/extraction/object_extraction_steps.py:204-366, ``_object_segmentation_thread``): threshold ->
morphological operations -> connected components of one chunk, for the branches without watershed seeds
(:316-317, :354-358).  Morphology follows /root/reference/syconn/proc/image.py (``apply_morphological_operations`` :485-507,
``_multi_mop_findobjects`` :357-438, ``get_aniso_struct`` :522-539); labels are numbered exactly like
``scipy.ndimage.label``.  Compute is ``sd_object_segmentation`` of the HIP library; there is no CPU fallback.

The watershed branch (``'binary_erosion'`` in the operation list -> erosion seeds, vigra distance transform, skimage
watershed, :319-352) is not implemented and raises ``NotImplementedError``.
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
                                    return_device: bool = False):
    """One probability map of a chunk -> ``(labels int32 (x,y,z), max_label)`` (+ the binary volume after the morphology if
    `return_mask`).  `prob`: uint8 (x,y,z) numpy array or device tensor; `threshold` in uint8 units as the reference
    compares it (``tmp_data > threshold``; 0: `prob` is already a 0/1 mask); `morph_ops`: names of ``scipy.ndimage``
    binary operations as in ``config['cell_objects']['extract_morph_op']``; `structure` defaults to
    ``get_aniso_struct(scaling)``.  `return_device`: leave the results on the GPU (int32 label tensor, 1-element count
    tensor[, uint8 mask]) -- the label volume is 4 bytes per voxel, and its consumer (``find_object_properties`` /
    ``segstats``, which take device tensors) does not need it on the host."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise RuntimeError('syconn_amd: no MI355X visible to PyTorch-ROCm; this package has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    L.check(lib.sd_init(device.index or 0), 'sd_init')
    morph_ops = list(morph_ops)
    for m in morph_ops:
        if m not in _MOPS:
            raise NotImplementedError(f"Only erosion or dilation allowed. Attempted to use morphological operation '{m}'.")
    if 'binary_erosion' in morph_ops:
        raise NotImplementedError('binary_erosion selects the watershed branch of _object_segmentation_thread '
                                  '(object_extraction_steps.py:319-352), which is not implemented on the device')
    names, counts = _count_subsequent_mops(morph_ops) if morph_ops else ([], [])
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
    pmax = max([c for n, c in zip(names, counts) if n != 'binary_opening'], default=0)
    ws_bytes = lib.sd_objseg_workspace_bytes(X, Y, Z, pmax)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    labels = torch.empty((X, Y, Z), dtype=torch.int32, device=device)
    max_label = torch.zeros(1, dtype=torch.int32, device=device)
    mask = torch.empty((X, Y, Z), dtype=torch.uint8, device=device) if return_mask else None
    n = len(names)
    ops_a = (C.c_int32 * max(n, 1))(*[_MOPS[m] for m in names])
    it_a = (C.c_int32 * max(n, 1))(*counts)
    L.check(lib.sd_object_segmentation(p.data_ptr(), X, Y, Z, float(threshold), ops_a, it_a, n,
                                       st.ctypes.data_as(C.c_void_p), *[int(s) for s in st.shape], labels.data_ptr(),
                                       max_label.data_ptr(), mask.data_ptr() if mask is not None else None, ws.data_ptr(),
                                       ws_bytes, torch.cuda.current_stream(device).cuda_stream), 'sd_object_segmentation')
    if return_device:
        return (labels, max_label) + ((mask,) if return_mask else ())
    out = (labels.cpu().numpy(), int(max_label.item()))
    if return_mask:
        out = out + (mask.cpu().numpy(),)
    return out
