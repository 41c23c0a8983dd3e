"""ORACLE (test infrastructure, NOT product code): CPU restatement of the first stage of SyConn's probability-map ->
object segmentation, non-watershed branches:

    /root/reference/syconn/extraction/object_extraction_steps.py
        :316-317   tmp_data = np.array(tmp_data > threshold, dtype=np.uint8)
        :354-356   mop_data = apply_morphological_operations(tmp_data.copy(), morph_ops, mop_kwargs=dict(structure=struct))
                   this_labels_data, max_label = scipy.ndimage.label(mop_data)
        :357-358   (no morphology configured)  this_labels_data, max_label = scipy.ndimage.label(tmp_data)
    /root/reference/syconn/proc/image.py
        :485-507   apply_morphological_operations (runs of equal operations become `iterations`)
        :357-438   _multi_mop_findobjects        (per-object bounding box, zero padding by `iterations` for closing /
                                                  dilation, write-back masks)
        :522-539   get_aniso_struct              (5x5x3 structuring element from the voxel scaling)

The watershed branch (:319-352: erosion seeds + vigra distance transform + skimage watershed) is DEFERRED: vigra and
skimage are absent from this image and its result cannot be pinned.

Only ``tests/``, ``tools/`` probes and ``bench.py``'s ``cpu_baseline`` leg may import this.

PINNED: ``tests/golden/make_golden_objseg.py`` lifts the reference's own ``apply_morphological_operations`` /
``_multi_mop_findobjects`` / ``get_aniso_struct`` by AST, executes them here with scipy (installed) and stores
inputs / outputs in ``tests/golden/g9_objseg.npz``; ``tests/test_objseg.py`` checks this restatement against them.

The restatement spells out what those functions do to a BINARY volume (one object, id 1), so that the device
implementation has a precise target:
  * the operation works on the bounding box of the foreground only (nothing happens without foreground);
  * closing / dilation: the box is zero-padded by `iterations` voxels per side -- LESS than the reach of the 5x5x3
    element in x/y -- the dilations are clipped to that padded window and the erosions treat everything outside it as
    background, then the result is cropped to the box and replaces it;
  * opening: erosions (outside the box = background) then dilations clipped to the box; only former foreground is
    rewritten (opening is anti-extensive, so this equals the result inside the box);
  * labels: 6-connected components numbered in raster order of their first voxel (scipy.ndimage.label), int32.
"""
from typing import List, Sequence, Tuple

import numpy as np
from scipy import ndimage


def get_aniso_struct_ref(scaling) -> np.ndarray:
    """image.py:522-539: (5,5,3) element in (x,y,z): the z-neighbours of the centre plus, in the centre plane, the
    4-connected cross dilated ``scaling[2] // scaling[0]`` times (clipped to 5x5)."""
    aniso = int(scaling[2] // scaling[0])
    assert scaling[1] // scaling[0] == 1 and aniso >= 1
    plane = np.zeros((5, 5), bool)
    for dx in range(-2, 3):
        for dy in range(-2, 3):
            plane[dx + 2, dy + 2] = abs(dx) + abs(dy) <= aniso
    centre = np.zeros((5, 5), bool)
    centre[2, 2] = True
    return np.stack([centre, plane, centre], axis=2)


def _offsets(struct: np.ndarray) -> np.ndarray:
    c = np.array(struct.shape) // 2
    return np.argwhere(struct) - c


def _erode(a: np.ndarray, offs: np.ndarray) -> np.ndarray:
    """one binary erosion of `a` (bool), everything outside the array = background."""
    out = np.ones_like(a)
    p = int(np.abs(offs).max())
    ap = np.pad(a, p)
    for o in offs:
        out &= ap[p + o[0]:p + o[0] + a.shape[0], p + o[1]:p + o[1] + a.shape[1], p + o[2]:p + o[2] + a.shape[2]]
    return out


def _dilate(a: np.ndarray, offs: np.ndarray) -> np.ndarray:
    """one binary dilation of `a` clipped to the array."""
    out = np.zeros_like(a)
    p = int(np.abs(offs).max())
    ap = np.pad(a, p)
    for o in offs:      # symmetric elements only (the reference's is): reflection == identity
        out |= ap[p - o[0]:p - o[0] + a.shape[0], p - o[1]:p - o[1] + a.shape[1], p - o[2]:p - o[2] + a.shape[2]]
    return out


def count_subsequent_mops(mops: Sequence[str]) -> Tuple[List[str], List[int]]:
    """image.py:510-519."""
    names, cnt = [mops[0]], [1]
    for m in mops[1:]:
        if m == names[-1]:
            cnt[-1] += 1
        else:
            names.append(m)
            cnt.append(1)
    return names, cnt


def apply_morphological_operations_ref(vol: np.ndarray, morph_ops: Sequence[str], struct: np.ndarray) -> np.ndarray:
    """image.py:485-507 + :357-438 on a binary uint8 volume (values 0 / 1)."""
    if len(morph_ops) == 0:
        return vol
    vol = vol.copy()
    offs = _offsets(struct.astype(bool))
    for mop, n in zip(*count_subsequent_mops(list(morph_ops))):
        fg = np.argwhere(vol == 1)
        if fg.size == 0:
            continue
        lo, hi = fg.min(axis=0), fg.max(axis=0) + 1
        box = tuple(slice(int(l), int(h)) for l, h in zip(lo, hi))
        mask = vol[box] == 1
        if mop in ('binary_closing', 'binary_dilation'):
            a = np.pad(mask, n)
            for _ in range(n):
                a = _dilate(a, offs)
            if mop == 'binary_closing':
                for _ in range(n):
                    a = _erode(a, offs)
            res = a[n:-n, n:-n, n:-n]
            proc = mask | (vol[box] == 0)
            vol[box][proc] = res[proc].astype(vol.dtype)
        elif mop in ('binary_opening', 'binary_erosion'):
            a = mask
            for _ in range(n):
                a = _erode(a, offs)
            if mop == 'binary_opening':
                for _ in range(n):
                    a = _dilate(a, offs)
            vol[box][mask] = a[mask].astype(vol.dtype)
        else:
            raise NotImplementedError(f"Only erosion or dilation allowed. Attempted to use morphological operation '{mop}'.")
    return vol


def object_segmentation_ref(prob: np.ndarray, threshold: float, morph_ops: Sequence[str], scaling) -> Tuple[np.ndarray, int]:
    """object_extraction_steps.py:316-317, 354-358 for one probability map (x,y,z) uint8: -> (labels int32, max_label)."""
    tmp = np.array(prob > threshold, dtype=np.uint8) if threshold != 0 else prob
    if 'binary_erosion' in morph_ops:
        raise NotImplementedError('watershed branch (object_extraction_steps.py:319-352) is deferred')
    if len(morph_ops):
        tmp = apply_morphological_operations_ref(tmp.copy(), morph_ops, get_aniso_struct_ref(np.asarray(scaling)))
    labels, max_label = ndimage.label(tmp)
    return labels, int(max_label)
