"""Globally unique objects across chunks (the steps behind the per-chunk first stage):
make_unique_labels / make_stitch_list / make_merge_list / apply_merge_list against tests/golden/g11_stitch.npz -- outputs of the
reference's own thread functions (tests/golden/make_golden_stitch.py) -- and from_probabilities_to_kd end to end."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'g11_stitch.npz'))
NAMES = [str(n) for n in G['names']]


def _grid(name):
    grid = tuple(int(v) for v in G[f'{name}_grid'])
    pos, n = {}, 0
    for x in range(grid[0]):
        for y in range(grid[1]):
            for z in range(grid[2]):
                pos[n] = (x, y, z)
                n += 1
    return pos


@pytest.mark.parametrize('name', NAMES)
def test_label_offsets_and_merge_list_host(name):
    """object_extraction_wrapper.py:300-312 and make_merge_list (object_extraction_steps.py:620-655): the same offsets, and the same
    partition of the ids with the smallest member as the representative."""
    from syconn_amd.extraction.object_extraction_steps import label_offsets, make_merge_list
    off, mx = label_offsets(G[f'{name}_nb_cc'])
    assert np.array_equal(off, G[f'{name}_offsets']) and mx == int(G[f'{name}_max_label'])
    md, ml = make_merge_list([tuple(p) for p in G[f'{name}_pairs'].tolist()], mx)
    assert ml.dtype == np.uint64 and np.array_equal(ml, G[f'{name}_canon'])
    assert set(md) == set(np.unique(G[f'{name}_pairs']).tolist()) and all(ml[k] == v for k, v in md.items())
    with pytest.raises(ValueError):
        make_merge_list([(1, mx + 1)], mx)


def test_calculate_chunk_numbers_for_box_host():
    from syconn_amd.extraction.object_extraction_wrapper import calculate_chunk_numbers_for_box
    from syconn_amd.knossos import ChunkDataset
    cd = ChunkDataset()
    cd.initialize(None, (40, 30, 20), (10, 10, 10), '/tmp/x/', box_coords=[0, 0, 0], fit_box_size=True)
    lst, tr = calculate_chunk_numbers_for_box(cd, np.array([12, 0, 5]), np.array([10, 10, 10]))
    want = [n for n, c in cd.chunk_dict.items() if 10 <= c.coordinates[0] < 30 and c.coordinates[1] == 0 and c.coordinates[2] < 20]
    assert lst == want and all(tr[n] == i for i, n in enumerate(lst))


def test_calculate_chunk_numbers_for_box_against_the_reference_function():
    """tests/golden/g13_chunk_numbers.npz: outputs of the reference's own calculate_chunk_numbers_for_box (its triple loop over the widened
    box) on a stand-in chunk set; here the same lists come from lattice arithmetic over the chunk origins."""
    from syconn_amd.extraction.object_extraction_wrapper import calculate_chunk_numbers_for_box
    from syconn_amd.knossos import ChunkDataset
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g13_chunk_numbers.npz'))
    for i in range(int(g['n_cases'])):
        cd = ChunkDataset()
        cd.initialize(None, g[f'c{i}_box'], g[f'c{i}_chunk'], '/tmp/x/', box_coords=[0, 0, 0], fit_box_size=True)
        off, size = g[f'c{i}_offset'].copy(), g[f'c{i}_size'].copy()
        lst, tr = calculate_chunk_numbers_for_box(cd, off, size)
        assert lst == g[f'c{i}_list'].tolist(), i
        assert tr == dict(zip(g[f'c{i}_tr_keys'].tolist(), g[f'c{i}_tr_vals'].tolist()))
        assert np.array_equal(off, g[f'c{i}_offset']) and np.array_equal(size, g[f'c{i}_size'])        # (arguments untouched)


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_unique_stitch_apply_against_reference(gpu, name):
    """Device kernels (sd_labels_make_unique, sd_labels_box_lut, the co-occurrence table of sd_segstats_scan) against the outputs of
    _make_unique_labels_thread, _make_stitch_list_thread and _apply_merge_list_thread: bit-exact unique volumes, the same set of
    id pairs, the same stitched volumes up to the choice of the representative (golden: canonicalised to the smallest member)."""
    from syconn_amd.extraction import object_extraction_steps as oes
    pos = _grid(name)
    ol, so, cs = G[f'{name}_overlap'], G[f'{name}_stitch_overlap'], G[f'{name}_chunk_size']
    uniq = {}
    for n in pos:
        lab = torch.from_numpy(G[f'{name}_labels_{n}']).to(gpu)
        uniq[n] = oes.make_unique_labels(lab, int(G[f'{name}_offsets'][n]))
        assert np.array_equal(uniq[n].cpu().numpy().view(np.uint64), G[f'{name}_unique_{n}'])
    pairs = oes.make_stitch_list(uniq, pos, ol, so)
    assert pairs == [tuple(p) for p in G[f'{name}_pairs'].tolist()]
    _, ml = oes.make_merge_list(pairs, int(G[f'{name}_max_label']))
    for n in pos:
        st = oes.apply_merge_list(uniq[n], cs, ml)
        assert tuple(st.shape) == tuple(cs) and np.array_equal(st.cpu().numpy().view(np.uint64), G[f'{name}_stitched_canon_{n}'])
    last = max(pos)
    if int(uniq[last].max()) > 1:                # a merge list that is too short for the ids of a chunk is refused
        with pytest.raises(ValueError):
            oes.apply_merge_list(uniq[last], cs, ml[:int(uniq[last].max())])


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_stitch_list_with_overlap_threshold_against_reference(gpu, name):
    """``overlap_thresh > 0`` (object_extraction_steps.py:597-615): the reference keeps a touching pair only if a cKDTree over the
    global voxel coordinates finds more than 10 % of the two objects' voxels coincident; tests/golden/g14_stitch_thresh.npz holds the
    pair lists its own function gives on the inputs of g11.  Here: co-occurrence counts over the 2 * overlap planes two chunk volumes
    share + the chunks' object sizes, on the device."""
    from syconn_amd.extraction import object_extraction_steps as oes
    g14 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g14_stitch_thresh.npz'))
    pos = _grid(name)
    ol, so = G[f'{name}_overlap'], G[f'{name}_stitch_overlap']
    uniq = {n: oes.make_unique_labels(torch.from_numpy(G[f'{name}_labels_{n}']).to(gpu), int(G[f'{name}_offsets'][n])) for n in pos}
    want = [tuple(p) for p in g14[f'{name}_pairs_thresh'].tolist()]
    assert oes.make_stitch_list(uniq, pos, ol, so, overlap_thresh=1) == want
    assert set(want) <= set(oes.make_stitch_list(uniq, pos, ol, so))


@pytest.mark.gpu
def test_from_probabilities_to_kd_gives_the_connected_components_of_the_whole_volume(gpu, tmp_path):
    """End to end on a KnossosDataset: per-chunk components (no morphology) + stitching = the 6-connected components of the
    thresholded WHOLE volume, as a partition of the voxels; ids are globally unique; the target dataset holds them."""
    import scipy.ndimage
    from scipy import ndimage
    from syconn_amd.extraction.object_extraction_wrapper import from_probabilities_to_kd
    from syconn_amd.knossos import ChunkDataset, KnossosDataset
    shape_xyz = (96, 80, 40)
    rng = np.random.default_rng(7)
    v = ndimage.gaussian_filter(rng.random(shape_xyz[::-1]), 2.5)
    prob = (255 * (v - v.min()) / (v.max() - v.min())).astype(np.uint8)          # (z,y,x)
    src = KnossosDataset()
    src.initialize_without_conf(str(tmp_path / 'prob'), boundary=shape_xyz, scale=(10., 10., 20.), experiment_name='p', mags=[1])
    src.save_raw(offset=(0, 0, 0), mags=[1], data=prob, data_mag=1)
    tgt = KnossosDataset()
    tgt.initialize_without_conf(str(tmp_path / 'seg'), boundary=shape_xyz, scale=(10., 10., 20.), experiment_name='s', mags=[1])
    cd = ChunkDataset()
    cd.initialize(src, np.array(shape_xyz), (48, 40, 20), str(tmp_path / 'cd') + '/', box_coords=[0, 0, 0], fit_box_size=True)
    thr = float(np.quantile(prob, 0.7))
    res = from_probabilities_to_kd({'obj': str(tmp_path / 'seg')}, cd, 'obj', ['obj'], prob_kd_path_dict={'obj': str(tmp_path / 'prob')},
                                   thresholds=[thr], overlap=np.array([2, 2, 2]), device=gpu,
                                   morph_ops={'obj': []}, min_seed_vx={'obj': 0}, scaling=(10, 10, 20))
    seg = KnossosDataset().initialize_from_knossos_path(str(tmp_path / 'seg')).load_seg(size=shape_xyz, offset=(0, 0, 0), mag=1)
    want, n_want = scipy.ndimage.label(prob > thr)
    assert seg.dtype == np.uint64 and np.array_equal(seg > 0, want > 0)
    # the same partition: every reference component carries exactly one id and vice versa
    a, b = want[want > 0].astype(np.int64), seg[seg > 0].astype(np.int64)
    pairs = np.unique(np.stack([a, b], 1), axis=0)
    assert len(pairs) == n_want == len(np.unique(b)) and len(np.unique(pairs[:, 0])) == len(np.unique(pairs[:, 1]))
    assert n_want >= 4 and len(res['stitch_list']['obj']) > 0 and res['max_labels']['obj'] > n_want
    # the same call with the reference's overlap test switched on: the driver's slab bookkeeping (only face slabs and object sizes of
    # a chunk wait for its later neighbours) gives what the all-volumes-in-memory form gives
    from syconn_amd.extraction import object_extraction_steps as oes
    res_t = from_probabilities_to_kd(None, cd, 'obj', ['obj'], prob_kd_path_dict={'obj': str(tmp_path / 'prob')}, thresholds=[thr],
                                     overlap=np.array([2, 2, 2]), device=gpu, morph_ops={'obj': []}, min_seed_vx={'obj': 0},
                                     scaling=(10, 10, 20), overlap_thresh=1)
    _, (ol_used, so_used), _, labels = oes.object_segmentation(cd, ['obj'], {'obj': str(tmp_path / 'prob')}, [thr], overlap=np.array([2, 2, 2]),
                                                              with_properties=False, device=gpu, keep_labels=True, morph_ops={'obj': []},
                                                              min_seed_vx={'obj': 0}, scaling=(10, 10, 20))
    nb = np.zeros(len(cd.chunk_dict), dtype=np.int64)
    for n_chunk, _, n_cc in res['cc_info_list']:
        nb[n_chunk] = n_cc
    offs, _ = oes.label_offsets(nb)
    uniq = {n: oes.make_unique_labels(labels[(n, 'obj')].to(gpu), int(offs[n])) for n in cd.chunk_dict}
    gpos = {n: tuple(int(v) for v in np.asarray(c.coordinates) // np.array([48, 40, 20])) for n, c in cd.chunk_dict.items()}
    assert res_t['stitch_list']['obj'] == oes.make_stitch_list(uniq, gpos, ol_used, so_used, overlap_thresh=1)
    assert set(res_t['stitch_list']['obj']) <= set(res['stitch_list']['obj'])


@pytest.mark.gpu
def test_overlay_cube_input_and_membrane_mask(gpu, tmp_path):
    """object_extraction_steps.py:254-270 (`load_from_kd_overlaycubes` + `transf_func_kd_overlay`: a segmentation as the source, no
    threshold) and :309-314 (`membrane_kd_path`: 'vc' voxels with a membrane probability above 255 * .4 are cleared before the
    threshold): per chunk the labels scipy gives for the volume prepared as the reference's inline statements prepare it."""
    import scipy.ndimage
    from scipy import ndimage
    from syconn_amd.extraction import object_extraction_steps as oes
    from syconn_amd.knossos import ChunkDataset, KnossosDataset
    shape_xyz = (48, 40, 24)
    rng = np.random.default_rng(11)

    def field(sigma):
        v = ndimage.gaussian_filter(rng.random(shape_xyz[::-1]), sigma)
        return (255 * (v - v.min()) / (v.max() - v.min())).astype(np.uint8)                 # (z,y,x)
    prob, memb = field(2.0), field(3.0)
    seg = (ndimage.label(field(2.0) > 140)[0] % 5).astype(np.uint64)                          # a label volume with ids 0..4

    def kd_of(name, data, seg_data=False):
        kd = KnossosDataset()
        kd.initialize_without_conf(str(tmp_path / name), boundary=shape_xyz, scale=(10., 10., 20.), experiment_name=name, mags=[1])
        (kd.save_seg if seg_data else kd.save_raw)(offset=(0, 0, 0), mags=[1], data=data, data_mag=1)
        if hasattr(kd, 'flush'):
            kd.flush()
        return str(tmp_path / name)
    p_prob, p_memb, p_seg = kd_of('vc', prob), kd_of('bar', memb), kd_of('ov', seg, True)
    cd = ChunkDataset()
    cd.initialize(None, np.array(shape_xyz), (24, 20, 24), str(tmp_path / 'cd') + '/', box_coords=[0, 0, 0], fit_box_size=True)
    ol = np.array([2, 2, 1])
    kw = dict(overlap=ol, morph_ops={'vc': [], 'ov': []}, min_seed_vx={}, scaling=(10, 10, 20), with_properties=False, device=gpu,
              keep_labels=True)

    def padded(a_zyx):
        a = np.zeros(tuple(np.array(shape_xyz) + 2 * ol), a_zyx.dtype)
        a[ol[0]:-ol[0], ol[1]:-ol[1], ol[2]:-ol[2]] = a_zyx.swapaxes(0, 2)
        return a

    def chunk_box(a, ch):
        c = ch.coordinates
        return a[c[0]:c[0] + 24 + 2 * ol[0], c[1]:c[1] + 20 + 2 * ol[1], c[2]:c[2] + 24 + 2 * ol[2]]
    thr = 120.0
    # membrane hook: tmp_data[membrane_data > 255 * .4] = 0, then tmp_data > threshold
    rows, _, _, labs = oes.object_segmentation(cd, ['vc'], {'vc': p_prob}, [thr], membrane_kd_path=p_memb, **kw)
    pp, pm = padded(prob), padded(memb)
    n_masked = 0
    for n, ch in cd.chunk_dict.items():
        tmp = chunk_box(pp, ch).copy()
        tmp[chunk_box(pm, ch) > 255 * .4] = 0
        want, nmax = scipy.ndimage.label(tmp > thr)
        assert np.array_equal(labs[(n, 'vc')].cpu().numpy(), want) and rows[n][2] == nmax
        n_masked += int(((chunk_box(pp, ch) > thr) & (chunk_box(pm, ch) > 255 * .4)).sum())
    assert n_masked > 0
    # overlay-cube input through a transform function: the voxels with id 3, labelled without a threshold
    rows, _, _, labs = oes.object_segmentation(cd, ['ov'], {'ov': p_seg}, None, load_from_kd_overlaycubes=True,
                                               transf_func_kd_overlay={'ov': lambda d: d == 3}, **kw)
    ps = padded(seg)
    tot = 0
    for n, ch in cd.chunk_dict.items():
        want, nmax = scipy.ndimage.label(chunk_box(ps, ch) == 3)
        assert np.array_equal(labs[(n, 'ov')].cpu().numpy(), want) and rows[n][2] == nmax
        tot += nmax
    assert tot > 0


@pytest.mark.gpu
def test_generate_subcell_kd_from_proba_from_the_working_directory(gpu, tmp_path):
    """object_extraction_wrapper.py:58-150: sources, thresholds and targets come from the working directory's config; an existing
    target needs overwrite=True; the result is the connected components of the thresholded map."""
    import scipy.ndimage
    from scipy import ndimage
    from syconn_amd import global_params
    from syconn_amd.extraction.object_extraction_wrapper import generate_subcell_kd_from_proba
    from syconn_amd.handler.config import generate_default_conf
    from syconn_amd.knossos import KnossosDataset
    shape_xyz = (64, 48, 32)
    rng = np.random.default_rng(3)
    v = ndimage.gaussian_filter(rng.random(shape_xyz[::-1]), 2.0)
    prob = (255 * (v - v.min()) / (v.max() - v.min())).astype(np.uint8)
    wd = str(tmp_path / 'wd')

    def kd_of(name, data, seg=False):
        kd = KnossosDataset()
        kd.initialize_without_conf(str(tmp_path / name), boundary=shape_xyz, scale=(10., 10., 20.), experiment_name=name, mags=[1])
        (kd.save_seg if seg else kd.save_raw)(offset=(0, 0, 0), mags=[1], data=data, data_mag=1)
        if hasattr(kd, 'flush'):
            kd.flush()
        return str(tmp_path / name)
    p_seg, p_mi = kd_of('cellseg', np.ones(shape_xyz[::-1], np.uint64), True), kd_of('mi', prob)
    thr = 0.62
    generate_default_conf(wd, scaling=(10, 10, 20), kd_seg=p_seg,
                          key_value_pairs=[('paths', {'kd_seg': p_seg, 'kd_mi': p_mi}), ('process_cell_organelles', ['mi']),
                                           ('cell_objects', {'probathresholds': {'mi': thr}, 'extract_morph_op': {'mi': []},
                                                             'min_seed_vx': {'mi': 0}})])
    global_params.wd = wd
    with pytest.raises(ValueError):      # (no erosion, no sigma: the automatic chunk overlap is 0 < the stitch overlap of 1, as in the reference)
        generate_subcell_kd_from_proba(['mi'], chunk_size=[32, 24, 32], device=gpu)
    ol = np.array([2, 2, 2])
    res = generate_subcell_kd_from_proba(['mi'], chunk_size=[32, 24, 32], device=gpu, overlap=ol, overwrite=True)
    seg = KnossosDataset().initialize_from_pyknossos_path(
        [str(tmp_path / 'wd' / 'knossosdatasets' / 'mi_seg' / f) for f in os.listdir(tmp_path / 'wd' / 'knossosdatasets' / 'mi_seg')
         if f.endswith('.pyk.conf')][0]).load_seg(size=shape_xyz, offset=(0, 0, 0), mag=1)
    want, n_want = scipy.ndimage.label(prob > thr * 255)
    assert np.array_equal(seg > 0, want > 0) and len(np.unique(seg[seg > 0])) == n_want >= 2
    pairs = np.unique(np.stack([want[want > 0], seg[seg > 0].astype(np.int64)], 1), axis=0)
    assert len(pairs) == n_want and len(res['chunk_list']) == 4
    with pytest.raises(FileExistsError):
        generate_subcell_kd_from_proba(['mi'], chunk_size=[32, 24, 32], device=gpu, overlap=ol)
    generate_subcell_kd_from_proba(['mi'], chunk_size=[32, 24, 32], device=gpu, overlap=ol, overwrite=True)
