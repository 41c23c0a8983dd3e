"""ORACLE (test infrastructure, NOT product code): CPU restatement of the first stage of SyConn's probability-map ->
object segmentation:

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

        :319-352   watershed branch ('binary_erosion' in the operation list; the default config's mi / sj / vc lists):
                   tmp_data = ops before the first erosion; markers = scipy.ndimage.label(ops from the first erosion on);
                   min_seed_vx filter + id hole filling + relabel_vol (block_processing_C.pyx:161-169);
                   distance = vigra distanceTransform(tmp_data, background=False, pixel_pitch=scaling);
                   labels = skimage.segmentation.watershed(-distance, markers, mask=tmp_data)

PARITY of the watershed branch: the marker volume (everything up to relabel_vol) is scipy / numpy in the reference and PINNED
(``seed_markers_ref`` against ``tests/golden/g10_objseg_ws.npz``, produced by the reference's own statements).  vigra and
skimage are absent from /root/reference and from this image (environment.yml pins vigra 1.11.1 / scikit-image 0.18/0.19):
``distance_transform_ref`` restates "exact Euclidean distance of every foreground voxel to the nearest background voxel of the
array, with pixel pitch" (computed with scipy's exact EDT), ``watershed_ref`` restates skimage's ``watershed_raveled`` (heap of
(value, age); a neighbour is labelled when it is PUSHED; neighbour order of the sorted raveled offsets).  One point skimage
leaves to its heap's internals is fixed here: marker voxels of equal value all enter with age 0 and are ordered by raster
index.  **PARITY UNPINNED** for these two steps.

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
        raise ValueError('watershed branch: use object_segmentation_watershed_ref (needs min_seed_vx)')
    if len(morph_ops):
        tmp = apply_morphological_operations_ref(tmp.copy(), morph_ops, get_aniso_struct_ref(np.asarray(scaling)))
    labels, max_label = ndimage.label(tmp)
    return labels, int(max_label)


# ---- watershed branch -----------------------------------------------------------------------------------------------------
def relabel_vol_ref(vol: np.ndarray, label_map: dict) -> None:
    """block_processing_C.pyx:161-169: in place, every voxel whose value is a key of `label_map` takes the mapped value."""
    keys = np.fromiter(label_map.keys(), dtype=np.int64, count=len(label_map))
    vals = np.fromiter(label_map.values(), dtype=np.int64, count=len(label_map))
    if keys.size == 0:
        return
    lut = np.arange(int(max(vol.max(), keys.max())) + 1, dtype=np.int64)
    lut[keys] = vals
    vol[...] = lut[vol].astype(vol.dtype)


def seed_markers_ref(tmp_data: np.ndarray, morph_ops: Sequence[str], struct: np.ndarray, min_seed_vx: int):
    """object_extraction_steps.py:319-347 -> (tmp_data after the pre-erosion operations, markers uint32)."""
    ops = list(morph_ops)
    first_erosion_ix = ops.index('binary_erosion')
    tmp_data = apply_morphological_operations_ref(tmp_data.copy(), ops[:first_erosion_ix], struct)
    markers = apply_morphological_operations_ref(tmp_data.copy(), ops[first_erosion_ix:], struct)
    markers = ndimage.label(markers)[0].astype(np.uint32)
    if min_seed_vx > 1:
        ixs, cnt = np.unique(markers, return_counts=True)
        m = (ixs != 0) & (cnt < min_seed_vx)
        ixs_del = np.sort(ixs[m])
        ixs_keep = np.sort(ixs[~m])
        label_m = {ix_del: 0 for ix_del in ixs_del}
        ii = len(ixs_keep) - 1
        for ix_del in ixs_del:            # the smallest freed ids go to the largest surviving ids, while they are smaller
            if ii < 0 or (ix_del > ixs_keep[ii]) or (ixs_keep[ii] == 0):
                break
            label_m[ixs_keep[ii]] = ix_del
            ii -= 1
        relabel_vol_ref(markers, label_m)
    return tmp_data, markers


def distance_transform_ref(mask: np.ndarray, pixel_pitch) -> Tuple[np.ndarray, np.ndarray]:
    """vigra distanceTransform(mask, background=False, pixel_pitch) restated: float32 Euclidean distance of every foreground
    voxel to the nearest background voxel INSIDE the array (the border is not background); also returns the exact squared
    distances (int64; pitches are integers)."""
    fgm = mask != 0
    if fgm.all():
        d2 = np.full(mask.shape, 0x3f000000, dtype=np.int64)
    else:
        d = ndimage.distance_transform_edt(fgm, sampling=[float(v) for v in pixel_pitch])
        d2 = np.rint(d * d).astype(np.int64)
    return np.sqrt(d2.astype(np.float32)), d2


def watershed_ref(d2: np.ndarray, markers: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """skimage.segmentation.watershed(-distance, markers, mask=mask) restated (watershed_raveled, connectivity 1, no
    compactness, no watershed line), with value = -distance compared through the exact squared distance `d2`."""
    import heapq
    sh = mask.shape
    out = np.where(mask != 0, markers, 0).astype(np.int32).ravel()
    m = (mask != 0).ravel()
    key = (-d2).ravel()
    sx, sy = sh[1] * sh[2], sh[2]
    heap = [(int(key[i]), 0, int(i)) for i in np.flatnonzero(out)]       # all markers enter with age 0
    heapq.heapify(heap)
    age = 0
    while heap:
        _, _, idx = heapq.heappop(heap)
        x, r = divmod(idx, sx)
        y, z = divmod(r, sy)
        for ok, q in ((x > 0, idx - sx), (y > 0, idx - sy), (z > 0, idx - 1), (z + 1 < sh[2], idx + 1),
                      (y + 1 < sh[1], idx + sy), (x + 1 < sh[0], idx + sx)):
            if not ok or not m[q] or out[q]:
                continue
            age += 1
            out[q] = out[idx]                                             # labelled at push time
            heapq.heappush(heap, (int(key[q]), age, int(q)))
    return out.reshape(sh)


def object_segmentation_watershed_ref(prob: np.ndarray, threshold: float, morph_ops: Sequence[str], scaling, min_seed_vx: int):
    """:316-352 for one probability map -> (labels int32, max_label, tmp_data, markers)."""
    tmp = np.array(prob > threshold, dtype=np.uint8) if threshold != 0 else prob
    struct = get_aniso_struct_ref(np.asarray(scaling))
    tmp, markers = seed_markers_ref(tmp, morph_ops, struct, min_seed_vx)
    _, d2 = distance_transform_ref(tmp, np.asarray(scaling).astype(np.uint32))
    labels = watershed_ref(d2, markers, tmp)
    return labels, int(labels.max()) if labels.size else 0, tmp, markers


def gaussian_kernel_ref(sigma: float) -> np.ndarray:
    """Taps of vigra's ``Kernel1D.initGaussian(sigma)`` (published algorithm; vigra is absent from the reference tree and this
    image -> parity UNPINNED): radius int(3 sigma + 0.5), at least 1; samples of exp(-t^2 / (2 sigma^2)); normalised to sum 1."""
    r = max(1, int(3.0 * sigma + 0.5))
    t = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 * t * t / (sigma * sigma))
    return w / w.sum()


def gaussian_smoothing_ref(data: np.ndarray, sigma) -> np.ndarray:
    """``vigra.gaussianSmoothing(data.astype(float32), sigma)`` as /root/reference/syconn/extraction/object_extraction_steps.py:
    296-297 calls it, restated with scipy: separable, axes in order, reflective border without repeating the edge
    (BORDER_TREATMENT_REFLECT == scipy's 'mirror'), sums in double, every pass stored as float32; sigma 0 skips an axis."""
    from scipy.ndimage import correlate1d
    sg = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (data.ndim,))
    out = data.astype(np.float32)
    for ax, s in enumerate(sg):
        if s > 0:
            out = correlate1d(out, gaussian_kernel_ref(float(s)), axis=ax, output=np.float32, mode='mirror')
    return out
