"""First stage of probability map -> object segmentation.

CPU part: the oracle restatement against goldens produced by the REFERENCE'S OWN morphology helpers
(/root/reference/syconn/proc/image.py:357-438, 485-539, lifted and executed by tests/golden/make_golden_objseg.py with the
inline threshold / scipy.ndimage.label statements of object_extraction_steps.py:316-317, 354-358).
GPU part (`-m gpu`): the HIP path through the C ABI -- binary volume after the morphology, label volume and label count
bit-exact against those goldens and against the oracle on larger random volumes."""
import os

import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle.objseg_ref import (apply_morphological_operations_ref, count_subsequent_mops, get_aniso_struct_ref,
                               object_segmentation_ref)

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g9_objseg.npz'))
NAMES = [str(n) for n in G['names']]


def _case(n):
    return (G[f'{n}_prob'], float(G[f'{n}_thr']), [str(o) for o in G[f'{n}_ops']], G[f'{n}_scaling'], G[f'{n}_struct'],
            G[f'{n}_mask'], G[f'{n}_labels'], int(G[f'{n}_max_label']))


@pytest.mark.parametrize('name', NAMES)
def test_oracle_matches_reference_goldens(name):
    prob, thr, ops, scaling, struct, mask, labels, max_label = _case(name)
    assert np.array_equal(get_aniso_struct_ref(scaling).astype(np.uint8), struct)
    tmp = np.array(prob > thr, dtype=np.uint8)
    got_mask = apply_morphological_operations_ref(tmp, ops, struct.astype(bool)) if ops else tmp
    assert np.array_equal(got_mask, mask)
    lab, mx = object_segmentation_ref(prob, thr, ops, scaling)
    assert lab.dtype == np.int32 and np.array_equal(lab, labels) and mx == max_label


def test_goldens_are_meaningful():
    """the fixture exercises what it claims: morphology changes the mask, several components, the closing quirk (the
    reference pads by `iterations` < reach of the element, so a closing can REMOVE voxels on the faces of the bounding box)"""
    assert int(G['plain_big_max_label']) > 100 and int(G['open_close_max_label']) > 5
    prob, thr = G['aniso3_prob'], float(G['aniso3_thr'])
    assert int(G['aniso3_mask'].sum()) < int((prob > thr).sum())
    assert count_subsequent_mops(['a', 'a', 'b', 'a']) == (['a', 'b', 'a'], [2, 1, 1])


def _blobs(shape, seed, sigma, thr_q):
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), sigma) + 0.02 * rng.random(shape)
    v = (v - v.min()) / max(v.max() - v.min(), 1e-9) if v.size > 1 else np.ones(shape)
    return (v * 255).astype(np.uint8), float(np.quantile(v * 255, thr_q)) - (1.0 if v.size == 1 else 0.0)


# ---- HIP path ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_gpu_object_segmentation_matches_reference_goldens(gpu, name):
    from syconn_amd.extraction.object_extraction_steps import get_aniso_struct, object_segmentation_first_stage
    prob, thr, ops, scaling, struct, mask, labels, max_label = _case(name)
    assert np.array_equal(get_aniso_struct(scaling).astype(np.uint8), struct)
    lab, mx, m = object_segmentation_first_stage(prob, thr, ops, scaling, return_mask=True)
    assert np.array_equal(m, mask), 'binary volume after the morphology differs'
    assert mx == max_label and lab.dtype == np.int32 and np.array_equal(lab, labels)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,sigma,q,ops,scaling', [
    ((96, 80, 72), 1.5, 0.80, [], (10, 10, 20)),
    ((70, 90, 50), 1.3, 0.75, ['binary_opening', 'binary_closing'], (10, 10, 20)),
    ((64, 64, 33), 1.2, 0.85, ['binary_closing', 'binary_opening'], (10, 10, 20)),           # the 'sj' recipe of the docs
    ((40, 37, 129), 2.0, 0.70, ['binary_closing', 'binary_closing', 'binary_opening'], (10, 10, 10)),
    ((1, 1, 1), 1.0, 0.0, ['binary_closing'], (10, 10, 20)),
    ((3, 200, 2), 1.0, 0.6, ['binary_opening'], (10, 10, 20))])
def test_gpu_object_segmentation_equals_oracle(gpu, shape, sigma, q, ops, scaling):
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    prob, thr = _blobs(shape, 5, sigma, q)
    want, want_max = object_segmentation_ref(prob, thr, ops, scaling)
    lab, mx = object_segmentation_first_stage(prob, thr, ops, scaling)
    assert mx == want_max and np.array_equal(lab, want)


@pytest.mark.gpu
def test_gpu_object_segmentation_worst_case_components_and_errors(gpu):
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    # 3D checkerboard: every foreground voxel its own component (no 6-neighbour), ids strictly in raster order
    x, y, z = np.indices((32, 30, 34))
    prob = (((x + y + z) & 1) * 255).astype(np.uint8)
    lab, mx = object_segmentation_first_stage(prob, 127.5)
    assert mx == int((prob > 0).sum())
    assert np.array_equal(lab[prob > 0], np.arange(1, mx + 1, dtype=np.int32))
    # one snake filling the volume: long union-find chains
    full = np.full((16, 40, 64), 255, np.uint8)
    full[:, 1::2, :-1] = 0
    full[:, 1::4, :] = 0
    full[:, 1::4, -1] = 255
    full[:, 3::4, 0] = 255
    lab, mx = object_segmentation_first_stage(full, 1.0)
    want, want_max = ndimage.label(full > 1.0)
    assert mx == want_max and np.array_equal(lab, want)
    # the default config's operation lists (config.yml:130-136) run: 'binary_erosion' selects the watershed branch
    lab, mx = object_segmentation_first_stage(prob, 100.0, ['binary_opening', 'binary_closing', 'binary_erosion'], min_seed_vx=10)
    assert mx == 0 and not lab.any()          # (a checkerboard does not survive the opening)
    with pytest.raises(NotImplementedError):
        object_segmentation_first_stage(prob, 100.0, ['binary_fill_holes'])
    with pytest.raises(TypeError):
        object_segmentation_first_stage(prob.astype(np.float32), 100.0)


@pytest.mark.gpu
def test_gpu_labels_stay_on_device_and_feed_the_statistics(gpu):
    """object segmentation -> label-volume statistics without leaving the GPU (int32 labels as SD_U32 volume): sizes and
    bounding boxes of the components equal what the oracle computes from the oracle's labels."""
    from oracle.objprops_ref import find_object_properties_np
    from syconn_amd.extraction.find_object_properties import find_object_properties
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    prob, thr = _blobs((48, 56, 40), 9, 1.4, 0.8)
    lab_dev, mx_dev = object_segmentation_first_stage(prob, thr, ['binary_closing', 'binary_opening'], return_device=True)
    assert lab_dev.is_cuda and lab_dev.dtype == torch.int32
    want, want_max = object_segmentation_ref(prob, thr, ['binary_closing', 'binary_opening'], (10, 10, 20))
    assert int(mx_dev.item()) == want_max
    assert find_object_properties(lab_dev) == find_object_properties_np(want.astype(np.uint32))


@pytest.mark.gpu
def test_gpu_full_chunk_size_against_scipy_and_properties(gpu):
    """A chunk of the size SyConn segments at once (512^3, object_extraction_wrapper.py:92-93 `chunk_size = [512]*3`, here
    384x512x512 to keep the CPU side at a few seconds): labels bit-identical to scipy.ndimage.label of the same mask, and
    size-independent properties of the morphology: an opening only removes voxels, applying the same opening to its own
    result changes nothing (idempotence), label count matches the label maximum."""
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    rng = np.random.default_rng(12)
    small = ndimage.gaussian_filter(rng.random((96, 128, 128)).astype(np.float32), 1.2)
    prob = np.kron(small, np.ones((4, 4, 4), np.float32))
    prob = ((prob - prob.min()) / (prob.max() - prob.min()) * 255).astype(np.uint8)
    prob[::7, ::5, ::3] = 255                                    # specks: many tiny components and run fragments
    thr = float(np.quantile(prob[::4, ::4, ::4], 0.85))
    lab, mx, mask = object_segmentation_first_stage(prob, thr, [], return_mask=True)
    want, want_max = ndimage.label(prob > thr)
    assert mx == want_max and int(lab.max()) == mx and np.array_equal(lab, want)
    lab_o, mx_o, mask_o = object_segmentation_first_stage(prob, thr, ['binary_opening'], return_mask=True)
    assert not np.any(mask_o & ~mask.astype(bool)) and int(mask_o.sum()) < int(mask.sum())
    lab_oo, mx_oo, mask_oo = object_segmentation_first_stage(mask_o, 0.0, ['binary_opening'], return_mask=True)
    assert np.array_equal(mask_oo, mask_o) and mx_oo == mx_o and np.array_equal(lab_oo, lab_o)
    assert np.array_equal(lab_o, ndimage.label(mask_o)[0])


# ---- watershed branch (object_extraction_steps.py:319-352; the default config's mi / sj / vc operation lists) -------------
GW = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g10_objseg_ws.npz'))
WNAMES = [str(n) for n in GW['names']]


def _wcase(n):
    return (GW[f'{n}_prob'], float(GW[f'{n}_thr']), [str(o) for o in GW[f'{n}_ops']], GW[f'{n}_scaling'],
            int(GW[f'{n}_min_seed']), GW[f'{n}_pre_mask'], GW[f'{n}_markers'])


@pytest.mark.parametrize('name', WNAMES)
def test_oracle_seed_markers_match_reference_goldens(name):
    """PINNED part of the watershed branch: tmp_data after the pre-erosion operations and the marker volume after the
    min_seed_vx filter + id hole filling + relabel_vol, against outputs of the reference's own statements (:320-347)."""
    from oracle.objseg_ref import seed_markers_ref
    prob, thr, ops, scaling, min_seed, pre, markers = _wcase(name)
    tmp = np.array(prob > thr, dtype=np.uint8)
    got_pre, got_markers = seed_markers_ref(tmp, ops, get_aniso_struct_ref(scaling), min_seed)
    assert np.array_equal(got_pre, pre)
    assert got_markers.dtype == np.uint32 and np.array_equal(got_markers, markers)


def test_watershed_goldens_are_meaningful():
    """the fixture exercises the id hole filling (kept ids are dense although seeds in between were deleted), a case where every
    seed is deleted, one without filter, and the four default operation lists"""
    for n, want_max in (('sj_default', 5), ('vc_default', 7), ('no_filter', 15), ('all_deleted', 0)):
        ids = np.unique(GW[f'{n}_markers'])
        assert int(ids.max()) == want_max and np.array_equal(ids, np.arange(want_max + 1))
    assert [str(o) for o in GW['mi_default_ops']].count('binary_erosion') == 4


def test_oracle_watershed_properties():
    """UNPINNED part (vigra / skimage absent): properties any correct marker flood inside a mask has -- labels only inside the
    mask, only marker ids, every mask component that holds a marker is filled completely, a component with one marker carries
    that id everywhere, markers keep their own label -- and the distance transform against a brute-force search."""
    from oracle.objseg_ref import distance_transform_ref, object_segmentation_watershed_ref
    for name in ('sj_default', 'iso', 'no_filter'):
        prob, thr, ops, scaling, min_seed, _, _ = _wcase(name)
        labels, mx, tmp, markers = object_segmentation_watershed_ref(prob, thr, ops, scaling, min_seed)
        assert labels.dtype == np.int32 and mx == labels.max()
        assert not labels[tmp == 0].any() and set(np.unique(labels)) <= set(np.unique(markers).tolist())
        assert np.array_equal(labels[markers > 0], markers[markers > 0].astype(np.int32))
        comp, nc = ndimage.label(tmp)
        for c in range(1, nc + 1):
            ids = np.unique(markers[comp == c])
            ids = ids[ids > 0]
            got = np.unique(labels[comp == c])
            if len(ids) == 0:
                assert np.array_equal(got, [0])
            else:
                assert np.array_equal(got, ids.astype(got.dtype))
    mask = (_blobs((14, 12, 10), 3, 1.5, 0.6)[0] > 120).astype(np.uint8)
    pitch = np.array((10, 10, 20))
    dist, d2 = distance_transform_ref(mask, pitch)
    bg = np.argwhere(mask == 0)
    for v in np.argwhere(mask != 0)[::7]:
        assert d2[tuple(v)] == (((bg - v) * pitch) ** 2).sum(1).min()
    assert dist.dtype == np.float32 and not d2[mask == 0].any()


@pytest.mark.gpu
@pytest.mark.parametrize('name', WNAMES)
def test_gpu_watershed_markers_match_reference_goldens(gpu, name):
    """device seeds: tmp_data and the relabelled marker volume bit-exact against the reference's own output; the final label
    volume bit-exact against the oracle's restatement of the (unpinned) distance transform + flood"""
    from oracle.objseg_ref import object_segmentation_watershed_ref
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    prob, thr, ops, scaling, min_seed, pre, markers = _wcase(name)
    lab, mx, m, mk = object_segmentation_first_stage(prob, thr, ops, scaling, return_mask=True, min_seed_vx=min_seed,
                                                     return_markers=True)
    assert np.array_equal(m, pre), 'tmp_data (mask before the erosions) differs'
    assert np.array_equal(mk.astype(np.uint32), markers), 'marker volume differs'
    want, want_max, _, _ = object_segmentation_watershed_ref(prob, thr, ops, scaling, min_seed)
    assert lab.dtype == np.int32 and mx == want_max and np.array_equal(lab, want)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,sigma,q,ops,scaling,min_seed', [
    ((96, 80, 72), 2.5, 0.70, ['binary_opening', 'binary_closing', 'binary_erosion'], (10, 10, 20), 10),
    ((80, 96, 48), 3.0, 0.60, ['binary_opening', 'binary_closing'] + ['binary_erosion'] * 4, (10, 10, 20), 50),
    ((64, 64, 64), 2.0, 0.65, ['binary_erosion', 'binary_erosion'], (10, 10, 10), 2),
    ((40, 37, 129), 2.0, 0.55, ['binary_closing', 'binary_erosion'], (9, 9, 20), 1),
    ((1, 1, 1), 1.0, 0.0, ['binary_erosion'], (10, 10, 20), 10),
    ((3, 200, 2), 1.0, 0.6, ['binary_erosion'], (10, 10, 20), 1)])
def test_gpu_watershed_equals_oracle(gpu, shape, sigma, q, ops, scaling, min_seed):
    """larger volumes with many touching blobs (several markers per mask component: the priority flood decides): markers, label
    volume and label count equal the oracle bit for bit"""
    from oracle.objseg_ref import object_segmentation_watershed_ref
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    prob, thr = _blobs(shape, 7, sigma, q)
    want, want_max, tmp, markers = object_segmentation_watershed_ref(prob, thr, ops, scaling, min_seed)
    lab, mx, m, mk = object_segmentation_first_stage(prob, thr, ops, scaling, return_mask=True, min_seed_vx=min_seed,
                                                     return_markers=True)
    assert np.array_equal(m, tmp) and np.array_equal(mk.astype(np.uint32), markers)
    assert mx == want_max and np.array_equal(lab, want)
    if np.prod(shape) > 1000:
        comp, nc = ndimage.label(tmp)
        multi = sum(1 for c in range(1, nc + 1) if len(np.unique(markers[comp == c])) > 2)
        print(f'{shape}: {nc} mask components, {multi} with several markers, {want_max} labels')


def test_auto_overlap_rule_and_default_config():
    """`overlap == "auto"` of object_segmentation (object_extraction_steps.py:152-166) with the default operation lists
    (config.yml:130-136: mi has 4 erosions, aniso 2) and the defaults the chunk driver reads from the config"""
    from syconn_amd.extraction.object_extraction_steps import auto_overlap
    from syconn_amd.handler.config import DEFAULTS
    co = DEFAULTS['cell_objects']
    assert auto_overlap(co['extract_morph_op'], [10, 10, 20]).tolist() == [16, 16, 8]
    assert auto_overlap({'sj': co['extract_morph_op']['sj']}, [10, 10, 20]).tolist() == [4, 4, 2]
    assert co['min_seed_vx']['mi'] == 50 and abs(co['probathresholds']['sj'] - 0.19047619) < 1e-12


@pytest.mark.gpu
def test_gpu_chunk_driver_kd_to_labels_to_properties(gpu, tmp_path):
    """KnossosDataset probability maps -> per chunk threshold / default-config morphology / watershed labels on the device ->
    find_object_properties from the device-resident labels (object_extraction_wrapper.py:58-150 -> object_segmentation ->
    _object_segmentation_thread): rows and properties equal the oracle run on the same chunk boxes."""
    from oracle.objprops_ref import find_object_properties_np
    from oracle.objseg_ref import object_segmentation_watershed_ref
    from syconn_amd.extraction.object_extraction_steps import object_segmentation
    from syconn_amd.handler.config import DEFAULTS
    from syconn_amd.knossos import ChunkDataset, KnossosDataset
    co = DEFAULTS['cell_objects']
    scaling = (10, 10, 20)
    paths, vols = {}, {}
    for i, name in enumerate(('sj', 'vc')):
        prob, _ = _blobs((96, 80, 48), 30 + i, 2.2, 0.5)                 # (x,y,z)
        vols[name] = prob
        kd = KnossosDataset()
        kd.initialize_without_conf(str(tmp_path / name), boundary=prob.shape, scale=scaling, experiment_name=name, mags=[1])
        kd.save_raw(offset=(0, 0, 0), mags=[1], data=np.ascontiguousarray(prob.swapaxes(0, 2)), data_mag=1)
        paths[name] = str(tmp_path / name)
    cset = ChunkDataset()
    cset.initialize(None, (96, 80, 48), (48, 40, 48), str(tmp_path / 'cd'), box_coords=[0, 0, 0], fit_box_size=True)
    thr = [0.45, 0.5]
    results, (overlap, stitch), props = object_segmentation(cset, ['sj', 'vc'], paths, thr, morph_ops=co['extract_morph_op'],
                                                            min_seed_vx=co['min_seed_vx'], scaling=scaling)
    assert overlap.tolist() == [16, 16, 8] and len(results) == 2 * len(cset.chunk_dict) == 8
    n_obj = 0
    for nb, chunk in cset.chunk_dict.items():
        lo = np.array(chunk.coordinates) - overlap
        size = np.array(chunk.size) + 2 * overlap
        for i, name in enumerate(('sj', 'vc')):
            box = np.zeros(size, np.uint8)                              # kd.load_raw: zeros outside the dataset
            a, b = np.maximum(lo, 0), np.minimum(lo + size, vols[name].shape)
            box[a[0] - lo[0]:b[0] - lo[0], a[1] - lo[1]:b[1] - lo[1], a[2] - lo[2]:b[2] - lo[2]] = \
                vols[name][a[0]:b[0], a[1]:b[1], a[2]:b[2]]
            want, want_max, _, _ = object_segmentation_watershed_ref(box, thr[i] * 255, co['extract_morph_op'][name], scaling,
                                                                     co['min_seed_vx'][name])
            assert [nb, name, want_max] in results
            assert props[(nb, name)] == find_object_properties_np(want.astype(np.uint32))
            n_obj += want_max
    assert n_obj > 8
    # the same driver with Gaussian pre-smoothing of one map (object_extraction_steps.py:296-297): overlap = 4 sigma (auto rule),
    # rows of the smoothed map = the oracle's segmentation of the oracle-smoothed box; the map with sigma 0 takes the plain path
    from oracle.objseg_ref import gaussian_smoothing_ref, object_segmentation_ref
    ops = {'sj': ['binary_opening', 'binary_closing'], 'vc': ['binary_opening']}
    sig, ov2 = [[2.0, 2.0, 1.0], [0, 0, 0]], np.array([8, 8, 4])
    boxes = {nb: (np.array(ch.coordinates) - ov2, np.array(ch.size) + 2 * ov2) for nb, ch in cset.chunk_dict.items()}
    smoothed = {nb: gaussian_smoothing_ref(_box(vols['sj'], lo, size), sig[0]) for nb, (lo, size) in boxes.items()}
    # a threshold no smoothed value sits on (then the masks are EQUAL, whatever the last bit of a sum is)
    allv = np.sort(np.concatenate([sm.ravel() for sm in smoothed.values()]))
    allv = allv[(allv > 100) & (allv < 140)]
    k = int(np.argmax(np.diff(allv)))
    thr_sj = float(0.5 * (float(allv[k]) + float(allv[k + 1])))          # the middle of the widest gap between smoothed values
    assert allv[k + 1] - allv[k] > 1e-3
    res2, _, _ = object_segmentation(cset, ['sj', 'vc'], paths, [thr_sj, 127.5], overlap=tuple(ov2), morph_ops=ops, min_seed_vx={},
                                     scaling=scaling, sigmas=sig, with_properties=False)
    assert len(res2) == 8
    for nb, (lo, size) in boxes.items():
        n_sj = object_segmentation_ref((smoothed[nb] > thr_sj).astype(np.uint8), 0, ops['sj'], scaling)[1]
        assert [nb, 'sj', n_sj] in res2 and n_sj > 0
        assert [nb, 'vc', object_segmentation_ref(_box(vols['vc'], lo, size), 127.5, ops['vc'], scaling)[1]] in res2


def _box(vol, lo, size):
    box = np.zeros(size, np.uint8)
    a, b = np.maximum(lo, 0), np.minimum(lo + size, vol.shape)
    box[a[0] - lo[0]:b[0] - lo[0], a[1] - lo[1]:b[1] - lo[1], a[2] - lo[2]:b[2] - lo[2]] = vol[a[0]:b[0], a[1]:b[1], a[2]:b[2]]
    return box


# ---- the marker flood on its own (sd_marker_flood: level-synchronous form, csrc/sd_objseg.hip::k_ws_flood) -------------------------------
def _flood_case(rng, sh, kind, n_mk):
    mask = rng.random(sh) < rng.uniform(0.5, 1.0)
    if kind == 'ties':                                          # few distinct levels: plateaus, first-in first-out decides
        d2 = rng.integers(1, 4, sh)
    elif kind == 'rough':                                       # rough landscape: most pushes go uphill (cascades)
        d2 = rng.integers(0, 50, sh)
    elif kind == 'flat':
        d2 = np.ones(sh, np.int64)
    else:                                                       # what the product feeds it: an exact squared distance transform
        d2 = np.rint(ndimage.distance_transform_edt(mask, sampling=(1, 1, 2)) ** 2)
    d2 = np.where(mask, d2, 0).astype(np.int32)
    markers = np.zeros(sh, np.int32)
    pts = np.flatnonzero(mask.ravel())
    for lab, p in enumerate(rng.choice(pts, min(n_mk, pts.size), replace=False), start=1):
        markers.ravel()[p] = lab
        for q in (p + 1, p + sh[2]):                            # some markers of several voxels
            if lab % 2 and q < mask.size and mask.ravel()[q] and markers.ravel()[q] == 0:
                markers.ravel()[q] = lab
    return d2, markers, mask.astype(np.uint8)


@pytest.mark.gpu
def test_gpu_marker_flood_equals_sequential_restatement(gpu):
    """random landscapes (plateaus, uphill cascades, distance transforms; 2..40 markers, ragged masks): the level-synchronous
    device flood labels every voxel like the sequential priority flood of the oracle"""
    from oracle.objseg_ref import watershed_ref
    from syconn_amd.extraction.object_extraction_steps import marker_flood
    rng = np.random.default_rng(11)
    n = 0
    for case in range(90):
        sh = tuple(int(v) for v in rng.integers(2, 15, 3)) if case < 75 else tuple(int(v) for v in rng.integers(20, 41, 3))
        kind = ('ties', 'rough', 'edt')[case % 3]
        d2, markers, mask = _flood_case(rng, sh, kind, int(rng.integers(2, 7)) if case < 75 else 40)
        want = watershed_ref(d2.astype(np.int64), markers, mask)
        got, mx = marker_flood(d2, markers, mask)
        assert np.array_equal(got, want), (case, sh, kind, int((got != want).sum()))
        assert mx == want.max()
        n += int((want > 0).sum() - (markers * (mask > 0) > 0).sum())
    assert n > 50000                                            # voxels the floods had to decide


@pytest.mark.gpu
def test_gpu_marker_flood_large_generations(gpu):
    """one level, one mask component, three markers in a 40 x 150 x 150 block: generations are whole breadth-first shells (beyond the
    8192 elements a generation may hold in LDS -> the global-memory sort), ties decided by queue order only"""
    from oracle.objseg_ref import watershed_ref
    from syconn_amd.extraction.object_extraction_steps import marker_flood
    rng = np.random.default_rng(3)
    d2, markers, mask = _flood_case(rng, (40, 150, 150), 'flat', 3)      # shells of > 8192 voxels
    mask[:] = 1
    d2[:] = 1
    want = watershed_ref(d2.astype(np.int64), markers, mask)
    got, mx = marker_flood(d2, markers, mask)
    assert mx == 3 and np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_watershed_level_synchronous_equals_sequential_kernel(gpu, monkeypatch):
    """the product path against the sequential device restatement (SD_WS_SEQUENTIAL=1) on volumes the Python oracle is too slow
    for: 192^3 of touching ellipsoids (many mask components with 2..6 markers, deleted seeds -> cascades) and a smooth random
    field (few huge components)"""
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    n = 192
    sph = np.zeros((n, n, n), np.uint8)
    r2 = np.random.default_rng(5)
    for _ in range(400):
        c = r2.integers(12, n - 12, 3)
        r = int(r2.integers(4, 12))
        for cc in (c, np.clip(c + r2.integers(-r, r + 1, 3) * 1.4, 12, n - 13).astype(int))[:1 + (r2.random() < 0.5)]:
            lo, hi = np.maximum(cc - r, 0), np.minimum(cc + r + 1, n)
            g = np.ogrid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
            sph[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]][((g[0] - cc[0]) ** 2 + (g[1] - cc[1]) ** 2 + ((g[2] - cc[2]) * 1.6) ** 2) <= r * r] = 255
    field, thr = _blobs((n, n, n), 3, 3.0, 0.7)
    ops = ['binary_opening', 'binary_closing', 'binary_erosion']
    for prob, t, o, ms in ((sph, 127.5, ops, 10), (sph, 127.5, ops + ['binary_erosion'] * 2, 40), (field, thr, ops, 10)):
        monkeypatch.delenv('SD_WS_SEQUENTIAL', raising=False)
        lab, mx, mk = object_segmentation_first_stage(prob, t, o, (10, 10, 20), min_seed_vx=ms, return_markers=True)
        monkeypatch.setenv('SD_WS_SEQUENTIAL', '1')
        want, wmx = object_segmentation_first_stage(prob, t, o, (10, 10, 20), min_seed_vx=ms)
        monkeypatch.delenv('SD_WS_SEQUENTIAL')
        assert mx == wmx and mx > 50 and np.array_equal(lab, want)
        assert int(((lab > 0) & (mk == 0)).sum()) > 100000


def test_level_synchronous_formulation_equals_sequential_flood():
    """CPU: the formulation the device flood implements (generations of one level claimed at once, cascades by min-propagation,
    pushes ordered by block only -- tools/experiments/ws_levelsync_proto.py, in-block order shuffled) against the sequential
    restatement of skimage's priority flood on tie-heavy, cascade-heavy and distance-transform landscapes"""
    import importlib.util
    from oracle.objseg_ref import watershed_ref
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'experiments', 'ws_levelsync_proto.py')
    spec = importlib.util.spec_from_file_location('ws_levelsync_proto', path)
    proto = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(proto)
    rng = np.random.default_rng(21)
    for case in range(45):
        sh = tuple(int(v) for v in rng.integers(3, 11, 3))
        d2, markers, mask = _flood_case(rng, sh, ('ties', 'rough', 'edt')[case % 3], int(rng.integers(2, 6)))
        want = watershed_ref(d2.astype(np.int64), markers, mask)
        got = proto.flood_levelsync(d2.astype(np.int64), markers, mask, rng)
        assert np.array_equal(got, want), (case, sh)


@pytest.mark.gpu
def test_gpu_watershed_branch_random_configurations(gpu):
    """40 random small configurations of the watershed branch -- operation lists with 1..4 erosions behind random openings /
    closings / dilations, min_seed_vx 0..40, isotropic and anisotropic pitches, ragged shapes: mask, markers, labels and label
    count equal the oracle's bit for bit"""
    from oracle.objseg_ref import object_segmentation_watershed_ref
    from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage
    rng = np.random.default_rng(2024)
    pre_choices = ([], ['binary_opening'], ['binary_closing'], ['binary_opening', 'binary_closing'], ['binary_dilation'],
                   ['binary_closing', 'binary_closing'])
    flooded = 0
    for case in range(40):
        shape = tuple(int(v) for v in rng.integers(6, 49, 3))
        prob, thr = _blobs(shape, 100 + case, float(rng.uniform(1.2, 3.0)), float(rng.uniform(0.45, 0.75)))
        ops = list(pre_choices[int(rng.integers(len(pre_choices)))]) + ['binary_erosion'] * int(rng.integers(1, 5))
        if rng.random() < 0.25:
            ops += ['binary_dilation']                          # operations after the erosions act on the seeds
        scaling = [(10, 10, 20), (10, 10, 10), (9, 9, 20), (4, 4, 35)][int(rng.integers(4))]
        min_seed = int(rng.integers(0, 41))
        want, want_max, tmp, markers = object_segmentation_watershed_ref(prob, thr, ops, scaling, min_seed)
        lab, mx, m, mk = object_segmentation_first_stage(prob, thr, ops, scaling, return_mask=True, min_seed_vx=min_seed,
                                                         return_markers=True)
        assert np.array_equal(m, tmp) and np.array_equal(mk.astype(np.uint32), markers), (case, shape, ops, scaling, min_seed)
        assert mx == want_max and np.array_equal(lab, want), (case, shape, ops, scaling, min_seed)
        flooded += int(((want > 0) & (markers == 0)).sum())
    assert flooded > 20000


@pytest.mark.gpu
@pytest.mark.parametrize('shape,sigma', [((40, 52, 31), 1.0), ((33, 20, 47), (1.5, 0.7, 2.2)), ((17, 64, 9), (0.0, 3.0, 0.0)),
                                         ((5, 6, 70), (4.0, 0.1, 1.0)), ((1, 30, 30), 2.0)])
def test_gpu_gaussian_threshold_equals_the_restatement(gpu, shape, sigma):
    """`sd_gaussian_threshold` (object_extraction_steps.py:296-297 + 316-317) against oracle/objseg_ref.gaussian_smoothing_ref:
    smoothed map equal to float32 rounding (sums in double on both sides, in different orders), masks equal except where the
    smoothed value sits on the threshold; windows longer than the axis (repeated mirroring), skipped axes, one-voxel axes."""
    from oracle.objseg_ref import gaussian_smoothing_ref
    from syconn_amd.extraction.object_extraction_steps import gaussian_threshold
    rng = np.random.default_rng(11)
    g = ndimage.gaussian_filter(rng.random(shape), 1.2)
    vol = ((g - g.min()) / (g.max() - g.min()) * 255).astype(np.uint8)
    thr = float(np.median(vol))
    mask, sm = gaussian_threshold(vol, sigma, thr, device=gpu, return_smoothed=True)
    ref = gaussian_smoothing_ref(vol, sigma)
    assert sm.dtype == np.float32 and sm.shape == vol.shape
    assert float(np.abs(sm - ref).max()) <= 6.2e-5, float(np.abs(sm - ref).max())      # (two float32 ulps at 255)
    clear = np.abs(ref - thr) > 1e-3
    assert np.array_equal(mask[clear], (ref > thr).astype(np.uint8)[clear]) and 0.05 < mask.mean() < 0.95
    assert set(np.unique(mask)) <= {0, 1}


@pytest.mark.gpu
def test_gpu_smoothed_mask_feeds_the_segmentation(gpu, tmp_path):
    """`gaussian_threshold` -> `object_segmentation_first_stage(mask, 0)`: the labels equal the oracle's segmentation of the
    oracle-smoothed mask."""
    from oracle.objseg_ref import gaussian_smoothing_ref, object_segmentation_ref
    from syconn_amd.extraction.object_extraction_steps import (gaussian_threshold, object_segmentation,
                                                               object_segmentation_first_stage)
    rng = np.random.default_rng(5)
    g = ndimage.gaussian_filter(rng.random((48, 40, 24)), 1.5)
    vol = ((g - g.min()) / (g.max() - g.min()) * 255).astype(np.uint8)
    sigma, ops = (1.0, 1.0, 0.5), ['binary_opening', 'binary_closing']
    ref = gaussian_smoothing_ref(vol, sigma)
    # a threshold no smoothed value sits on (then the masks must be EQUAL, whatever the last bit of a sum is)
    thr = next(t for t in float(np.percentile(vol, 70)) + np.arange(0.0, 1.0, 0.0137) if np.abs(ref - t).min() > 2e-3)
    want, n_want = object_segmentation_ref((ref > thr).astype(np.uint8), 0, ops, (10, 10, 20))
    mask = gaussian_threshold(vol, sigma, thr, device=gpu)
    got, n_got = object_segmentation_first_stage(mask, 0.0, ops, (10, 10, 20), device=gpu)
    assert n_got == n_want and n_want > 3 and np.array_equal(got, want)

