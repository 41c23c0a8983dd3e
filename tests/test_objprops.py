"""Label-volume statistics (/root/reference/syconn/extraction/find_object_properties_C.pyx).

CPU part: the oracle against the reference's own known-answer test (tests/test_segmentation_analysis.py:19-52, stored as
tests/golden/g8_objprops.npz) and the vectorised oracle against the literal restatement.
GPU part (`-m gpu`): the HIP path through the C ABI against the oracle, bit-exact (integer work)."""
import os

import numpy as np
import pytest

from oracle.objprops_ref import (find_object_properties_loops, find_object_properties_np,
                                 map_subcell_extract_props_loops, map_subcell_extract_props_np)

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g8_objprops.npz'))


def _check_like_reference_test(vol, ids, counts, lo, hi, result):
    """The assertions of /root/reference/tests/test_segmentation_analysis.py:29-52."""
    repcoord_dc, bb_dc, cnt_dc = result
    assert 0 not in repcoord_dc and 0 not in bb_dc and 0 not in cnt_dc, 'Background properties must not be extracted.'
    assert set(cnt_dc) == set(int(i) for i in ids)
    for i, e in enumerate(ids.tolist()):
        assert cnt_dc[e] == int(counts[i]), 'Count of the voxels not working.'
        ll = repcoord_dc[e]
        assert int(vol[ll[0], ll[1], ll[2]]) == e, "Object voxel dictionary dosen't match."
        assert np.all(lo[i] == bb_dc[e][0]) and np.all(hi[i] == bb_dc[e][1]), 'Bounding box dictionary mismatch.'


@pytest.mark.parametrize('name', ['sample', 'toy'])
@pytest.mark.parametrize('impl', [find_object_properties_loops, find_object_properties_np])
def test_oracle_matches_reference_known_answers(name, impl):
    if name == 'toy' and impl is find_object_properties_loops:
        vol = G['toy_vol'][:12, :12, :12]            # the pure-Python loops are for small inputs
        from tests.golden.make_golden_objprops import expectations
        _check_like_reference_test(vol, *expectations(vol), impl(vol))
        return
    _check_like_reference_test(G[f'{name}_vol'], G[f'{name}_ids'], G[f'{name}_counts'], G[f'{name}_bb_lo'],
                               G[f'{name}_bb_hi'], impl(G[f'{name}_vol']))


def _random_case(seed, shape, nid, dtype=np.uint64, coherent=False):
    rng = np.random.default_rng(seed)
    if coherent:      # blocky supervoxel-like labels
        small = rng.integers(0, nid, [max(1, s // 5 + 1) for s in shape])
        vol = np.kron(small, np.ones((5, 5, 5), dtype=np.int64))[:shape[0], :shape[1], :shape[2]]
    else:
        vol = rng.integers(0, nid, shape)
    return vol.astype(dtype)


def test_vectorised_oracle_equals_literal_loops():
    for seed, shape, nid, coh in ((1, (7, 9, 11), 12, False), (2, (6, 5, 17), 5, True), (3, (1, 1, 3), 3, False)):
        cell = _random_case(seed, shape, nid, coherent=coh)
        subs = np.stack([_random_case(seed + 10 + k, shape, 4, coherent=coh) for k in range(2)])
        assert find_object_properties_loops(cell) == find_object_properties_np(cell)
        assert map_subcell_extract_props_loops(cell, subs) == map_subcell_extract_props_np(cell, subs)
    # representative coordinate = first voxel of the raster scan (find_object_properties_C.pyx:46)
    v = np.zeros((3, 3, 3), np.uint64)
    v[2, 0, 1] = v[1, 2, 2] = v[1, 2, 0] = 9
    assert find_object_properties_np(v)[0][9] == [1, 2, 0]


# ---- HIP path ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', ['sample', 'toy'])
def test_gpu_find_object_properties_reference_known_answers(gpu, name):
    from syconn_amd.extraction.find_object_properties import find_object_properties
    res = find_object_properties(G[f'{name}_vol'])
    _check_like_reference_test(G[f'{name}_vol'], G[f'{name}_ids'], G[f'{name}_counts'], G[f'{name}_bb_lo'],
                               G[f'{name}_bb_hi'], res)
    assert res == find_object_properties_np(G[f'{name}_vol'])          # incl. the exact representative coordinates


@pytest.mark.gpu
@pytest.mark.parametrize('shape,nid,dtype,coh', [((50, 50, 50), 1000, np.uint64, False), ((33, 47, 129), 40, np.uint32, True),
                                                ((5, 3, 64), 7, np.uint64, True), ((2, 3, 1), 3, np.uint32, False),
                                                ((64, 96, 200), 3000, np.uint64, True)])
def test_gpu_objprops_and_mapping_equal_oracle(gpu, shape, nid, dtype, coh):
    from syconn_amd.extraction.find_object_properties import (find_object_properties, map_subcell_C,
                                                              map_subcell_extract_props)
    cell = _random_case(11, shape, nid, dtype, coh)
    if dtype == np.uint64:
        cell[cell == 3] = np.uint64(2 ** 63 + 12345)               # ids beyond 32 and 63 bits
    subs = np.stack([_random_case(20 + k, shape, max(2, nid // 3), dtype, coh) for k in range(3)])
    assert find_object_properties(cell) == find_object_properties_np(cell)
    want = map_subcell_extract_props_np(cell, subs)
    got = map_subcell_extract_props(cell, subs)
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    assert map_subcell_C(cell, subs) == want[2]


@pytest.mark.gpu
@pytest.mark.parametrize('shape,nid,dtype,coh', [((3, 5, 4), 3, np.uint64, False), ((1, 1, 4), 2, np.uint32, False), ((3, 1, 12), 4, np.uint64, True),
                                                ((7, 3, 260), 9, np.uint32, False), ((9, 4, 256), 30, np.uint64, True),
                                                ((37, 21, 72), 200, np.uint64, True), ((6, 10, 1024), 5, np.uint32, True)])
def test_gpu_four_voxels_per_lane_form_equals_one_voxel_form_and_oracle(gpu, shape, nid, dtype, coh, monkeypatch):
    """Rows of a multiple of 4 voxels run the pass with 4 voxels per lane (`k_segstats_scan<L, true>`): same tables as the one-voxel
    form (`SD_SEGSTATS_V1=1`) and as the numpy oracle -- row ends inside a wave, a volume that ends inside a wave, runs that start
    at every in-lane position, uint32 and uint64 labels, ids beyond 63 bits."""
    from syconn_amd.extraction.find_object_properties import map_subcell_C, map_subcell_extract_props
    assert shape[2] % 4 == 0
    cell = _random_case(31, shape, nid, dtype, coh)
    if dtype == np.uint64:
        cell[cell == 1] = np.uint64(2 ** 63 + 77)
    subs = np.stack([_random_case(40 + k, shape, max(2, nid // 2), dtype, coh and k != 1) for k in range(3)])
    want = map_subcell_extract_props_np(cell, subs)
    got4 = map_subcell_extract_props(cell, subs)
    assert got4[0] == want[0] and got4[1] == want[1] and got4[2] == want[2]
    assert map_subcell_C(cell, subs) == want[2]
    monkeypatch.setenv('SD_SEGSTATS_V1', '1')
    got1 = map_subcell_extract_props(cell, subs)
    assert got1[0] == want[0] and got1[1] == want[1] and got1[2] == want[2]


@pytest.mark.gpu
def test_gpu_objprops_edge_cases(gpu):
    from syconn_amd.extraction.find_object_properties import find_object_properties, map_subcell_extract_props, segstats
    empty = np.zeros((4, 5, 6), np.uint64)
    assert find_object_properties(empty) == ({}, {}, {})
    c, s, m = map_subcell_extract_props(empty, empty[None])
    assert c == [{}, {}, {}] and s == [[{}], [{}], [{}]] and m == [{}]
    # every voxel its own object + a table that starts far too small: overflow is detected and the pass repeated
    vol = (np.arange(20 * 20 * 20, dtype=np.uint64) + 1).reshape(20, 20, 20)
    r = segstats(vol, [vol], cap_obj=1024, cap_pair=1024)
    assert len(r.cell[0]) == 8000 and np.array_equal(r.cell[0], np.arange(1, 8001, dtype=np.uint64))
    assert np.all(r.cell[2] == 1) and np.array_equal(r.cell[1], np.arange(8000))
    assert len(r.pairs[0][0]) == 8000 and np.all(r.pairs[0][2] == 1)
    with pytest.raises(AssertionError):
        map_subcell_extract_props(np.zeros((2, 2, 2), np.uint64), np.zeros((1, 2, 2, 3), np.uint64))
    with pytest.raises(TypeError):
        find_object_properties(np.zeros((2, 2, 2), np.int16))


@pytest.mark.gpu
def test_gpu_objprops_full_chunk_size_properties(gpu):
    """256x512x512 uint64 supervoxel-like volume with three organelle volumes (one pass over 4 volumes of 0.5 GiB each):
    size-independent properties -- sizes sum to the number of non-zero voxels, overlap counts of an organelle id sum to its
    voxels inside cells, bounding boxes contain the representative coordinate -- and equality with the numpy oracle on a
    corner the oracle finishes in seconds."""
    from syconn_amd.extraction.find_object_properties import segstats, find_object_properties
    rng = np.random.default_rng(21)
    shape = (256, 512, 512)
    small = rng.integers(0, 3000, (17, 33, 33))
    cell = np.kron(small, np.ones((16, 16, 16), np.int64))[:shape[0], :shape[1], :shape[2]].astype(np.uint64)
    subs = []
    for k in range(3):
        s = rng.integers(0, 40, (33, 65, 65))
        s[s > 6] = 0
        subs.append(np.kron(s, np.ones((8, 8, 8), np.int64))[:shape[0], :shape[1], :shape[2]].astype(np.uint64) * (k + 1))
    r = segstats(cell, subs)
    assert int(r.cell[2].sum()) == int(np.count_nonzero(cell))
    for k in range(3):
        ids, first, size, bb = r.sub[k]
        assert int(size.sum()) == int(np.count_nonzero(subs[k]))
        s_ids, c_ids, cnt = r.pairs[k]
        both = (subs[k] != 0) & (cell != 0)
        assert int(cnt.sum()) == int(np.count_nonzero(both))
        rc = np.stack(np.unravel_index(first, shape), axis=1)
        assert np.all(rc >= bb[:, 0]) and np.all(rc < bb[:, 1])
    corner = cell[:64, :128, :128]
    assert find_object_properties(np.ascontiguousarray(corner)) == find_object_properties_np(np.ascontiguousarray(corner))
