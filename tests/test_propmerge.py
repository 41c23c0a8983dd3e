"""Dataset-wide merging of per-chunk object properties and overlap counts (the chunk driver of
map_subcell_extract_props) against tests/golden/g12_propmerge.npz: outputs of the reference's own merge_prop_dicts / merge_map_dicts /
convert_nvox2ratio_mapdict / invert_mdc (tests/golden/make_golden_propmerge.py)."""
import copy
import os
import pickle
import sys
from collections import defaultdict

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'g12_propmerge.npz'))
NAMES = [str(n) for n in G['names']]


def _case(name):
    exp = pickle.loads(G[f'{name}_expected'].tobytes())
    mov = pickle.loads(G[f'{name}_min_obj_vx'].tobytes())
    orgs = [str(o) for o in G[f'{name}_organelles']]
    return exp, mov, orgs, G[f'{name}_cell'], {o: G[f'{name}_sub_{o}'] for o in orgs}, G[f'{name}_chunk_size']


def _plain(t):
    return [{int(k): v for k, v in t[0].items()}, {int(k): v for k, v in t[1].items()}, {int(k): int(v) for k, v in t[2].items()}]


@pytest.mark.parametrize('name', NAMES)
def test_merge_functions_host(name):
    """The host merge functions fed with per-chunk dictionaries of the CPU oracle: the reference's totals."""
    from oracle.objprops_ref import map_subcell_extract_props_np
    from syconn_amd.proc.sd_proc import convert_nvox2ratio_mapdict, invert_mdc, merge_map_dicts, merge_prop_dicts
    exp, mov, orgs, cell, subs, cs = _case(name)
    shape = np.array(cell.shape)
    grid = -(-shape // cs)
    pad = lambda a: np.pad(a, [(0, int(grid[i] * cs[i] - shape[i])) for i in range(3)])
    cell_p, sub_p = pad(cell), {k: pad(v) for k, v in subs.items()}
    cpd = [{}, defaultdict(list), {}]
    scpd = [[{}, defaultdict(list), {}] for _ in orgs]
    scmd = [{} for _ in orgs]

    def faces(a):
        return set(np.unique(np.concatenate([a[0].flat, a[-1].flat, a[:, 0].flat, a[:, -1].flat, a[:, :, 0].flat, a[:, :, -1].flat])).tolist())
    for x in range(0, int(grid[0] * cs[0]), int(cs[0])):
        for y in range(0, int(grid[1] * cs[1]), int(cs[1])):
            for z in range(0, int(grid[2] * cs[2]), int(cs[2])):
                off = np.array([x, y, z])
                sl = tuple(slice(int(off[i]), int(off[i] + cs[i])) for i in range(3))
                cd, sd, md = map_subcell_extract_props_np(cell_p[sl], np.stack([sub_p[o][sl] for o in orgs]))
                if mov['sv'] > 1:
                    for ix in set(cd[0]) - faces(cell_p[sl]):
                        if cd[2][ix] < mov['sv']:
                            del cd[0][ix], cd[1][ix], cd[2][ix]
                merge_prop_dicts([cpd, cd], off)
                for i, o in enumerate(orgs):
                    t = [sd[0][i], sd[1][i], sd[2][i]]
                    if mov[o] > 1:
                        for ix in set(t[0]) - faces(sub_p[o][sl]):
                            if t[2][ix] < mov[o]:
                                del t[0][ix], t[1][ix], t[2][ix]
                                md[i].pop(ix, None)
                    merge_map_dicts([scmd[i], md[i]])
                    merge_prop_dicts([scpd[i], t], off)
    assert _plain(cpd) == exp['cell']
    for i, o in enumerate(orgs):
        assert _plain(scpd[i]) == exp['sub'][o]
        m = {int(a): {int(b): int(c) for b, c in d.items()} for a, d in scmd[i].items()}
        assert m == exp['maps'][o] and invert_mdc(m) == exp['inverted'][o]
        r = copy.deepcopy(m)
        convert_nvox2ratio_mapdict(r)
        assert {a: {b: float(c) for b, c in d.items()} for a, d in r.items()} == exp['ratio'][o]


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_map_subcell_extract_props_over_knossos_datasets(gpu, name, tmp_path):
    """The chunk driver on the device natives, reading cell and organelle segmentations from KnossosDataset overlay cubes."""
    from syconn_amd.knossos import KnossosDataset
    from syconn_amd.proc.sd_proc import map_subcell_extract_props
    exp, mov, orgs, cell, subs, cs = _case(name)

    def write(path, vol_xyz):
        kd = KnossosDataset()
        kd.initialize_without_conf(str(path), boundary=vol_xyz.shape, scale=(10., 10., 20.), experiment_name='seg', mags=[1])
        kd.save_seg(offset=(0, 0, 0), mags=[1], data=np.ascontiguousarray(vol_xyz.swapaxes(0, 2)), data_mag=1)
        if hasattr(kd, 'flush'):
            kd.flush()
        return str(path)
    p_cell = write(tmp_path / 'cell', cell)
    p_sub = {o: write(tmp_path / o, subs[o]) for o in orgs}
    cpd, scpd, scmd = map_subcell_extract_props(p_cell, p_sub, chunk_size=cs, min_obj_vx=mov, device=gpu)
    assert _plain(cpd) == exp['cell']
    for o in orgs:
        assert _plain(scpd[o]) == exp['sub'][o]
        assert {int(a): {int(b): int(c) for b, c in d.items()} for a, d in scmd[o].items()} == exp['maps'][o]
    # the merged tables themselves and the array forms of convert_nvox2ratio_mapdict / invert_mdc
    cell_t, sub_t, map_t = map_subcell_extract_props(p_cell, p_sub, chunk_size=cs, min_obj_vx=mov, device=gpu, as_tables=True)
    assert np.all(np.diff(cell_t.ids.astype(np.int64)) > 0) and cell_t.box_begin[-1] == len(cell_t.boxes)
    for o in orgs:
        m = map_t[o]
        assert m.inverted() == exp['inverted'][o]
        r = m.ratios()
        got = {}
        for a, b, c in zip(m.sub_ids.tolist(), m.cell_ids.tolist(), r.tolist()):
            got.setdefault(a, {})[b] = c
        assert got == exp['ratio'][o]


@pytest.mark.gpu
@pytest.mark.parametrize('n,n_ids,seed', [(1, 1, 0), (1000, 37, 1), (200000, 5000, 2), (70000, 70000, 3)])
def test_propmerge_kernels_against_numpy(gpu, n, n_ids, seed):
    """sd_propmerge_objects / sd_propmerge_pairs on random records: stable order inside an id, sums, last representative."""
    import torch
    from syconn_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(seed)
    pool = np.unique(rng.integers(1, 2 ** 62, n_ids * 2, dtype=np.int64).astype(np.uint64))[:n_ids]
    rng.shuffle(pool)
    n_ids = len(pool)
    ids = pool[rng.integers(0, n_ids, n)]
    sizes = rng.integers(1, 10 ** 9, n, dtype=np.int64)
    rc = rng.integers(0, 2 ** 20, (n, 3), dtype=np.int32)
    bb = rng.integers(0, 2 ** 20, (n, 6), dtype=np.int32)
    dev = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(gpu)
    d_ids, d_sz, d_rc, d_bb = dev(ids), dev(sizes), dev(rc), dev(bb)
    uniq, tot = torch.empty(n, dtype=torch.int64, device=gpu), torch.empty(n, dtype=torch.int64, device=gpu)
    lrc, beg = torch.empty((n, 3), dtype=torch.int32, device=gpu), torch.empty(n, dtype=torch.int32, device=gpu)
    bbs, cnt = torch.empty((n, 6), dtype=torch.int32, device=gpu), torch.zeros(1, dtype=torch.int64, device=gpu)
    tb = lib.sd_propmerge_temp_bytes(n)
    tmp = torch.empty(tb, dtype=torch.uint8, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.sd_propmerge_objects(d_ids.data_ptr(), d_sz.data_ptr(), d_rc.data_ptr(), d_bb.data_ptr(), n, uniq.data_ptr(), tot.data_ptr(),
                                     lrc.data_ptr(), beg.data_ptr(), bbs.data_ptr(), cnt.data_ptr(), tmp.data_ptr(), tb, st))
    u = int(cnt.item())
    order = np.argsort(ids, kind='stable')
    sid = ids[order]
    heads = np.flatnonzero(np.concatenate(([True], sid[1:] != sid[:-1])))
    assert u == len(heads)
    assert np.array_equal(uniq[:u].cpu().numpy().view(np.uint64), sid[heads])
    assert np.array_equal(tot[:u].cpu().numpy(), np.add.reduceat(sizes[order], heads))
    assert np.array_equal(beg[:u].cpu().numpy(), heads.astype(np.int32))
    assert np.array_equal(bbs.cpu().numpy(), bb[order])
    last = np.concatenate((heads[1:], [n])) - 1
    assert np.array_equal(lrc[:u].cpu().numpy(), rc[order][last])
    # pairs
    cells = pool[rng.integers(0, max(1, n_ids // 7), n)]
    d_c = dev(cells)
    o_s, o_c, o_n = (torch.empty(n, dtype=torch.int64, device=gpu) for _ in range(3))
    L.check(lib.sd_propmerge_pairs(d_ids.data_ptr(), d_c.data_ptr(), d_sz.data_ptr(), n, o_s.data_ptr(), o_c.data_ptr(), o_n.data_ptr(),
                                   cnt.data_ptr(), tmp.data_ptr(), tb, st))
    u = int(cnt.item())
    order = np.lexsort((cells, ids))
    a, b = ids[order], cells[order]
    heads = np.flatnonzero(np.concatenate(([True], (a[1:] != a[:-1]) | (b[1:] != b[:-1]))))
    assert u == len(heads)
    assert np.array_equal(o_s[:u].cpu().numpy().view(np.uint64), a[heads]) and np.array_equal(o_c[:u].cpu().numpy().view(np.uint64), b[heads])
    assert np.array_equal(o_n[:u].cpu().numpy(), np.add.reduceat(sizes[order], heads))


@pytest.mark.gpu
def test_chunk_driver_on_device_volumes_equals_whole_volume_statistics(gpu):
    """Objects that never touch a chunk face are filtered per chunk; with the filter off the merged sizes / overlap counts of a chunked
    pass equal one pass over the whole volume (device-resident label volumes through `chunk_loader`, ragged last chunks)."""
    import torch
    from scipy import ndimage
    from syconn_amd.extraction.find_object_properties import segstats
    from syconn_amd.proc.sd_proc import map_subcell_extract_props

    class KD:                       # the driver asks a dataset only for its extent
        boundary = np.array([70, 50, 44])
    rng = np.random.default_rng(5)
    def labels(q):
        lab, _ = ndimage.label(ndimage.gaussian_filter(rng.random(tuple(KD.boundary)), 1.5) > q)
        return (lab.astype(np.uint64) * 7919) % 100003 * (lab > 0)
    vols = {'sv': labels(0.5), 'mi': labels(0.53), 'vc': labels(0.55)}
    dvols = {k: torch.from_numpy(v.view(np.int64)).to(gpu) for k, v in vols.items()}

    def loader(name, off, size):
        out = torch.zeros(tuple(int(s) for s in size), dtype=torch.int64, device=gpu)
        hi = np.minimum(off + size, KD.boundary)
        n = hi - off
        out[:n[0], :n[1], :n[2]] = dvols[name][off[0]:hi[0], off[1]:hi[1], off[2]:hi[2]]
        return out
    import syconn_amd.proc.sd_proc as sp
    orig = sp.kd_factory
    sp.kd_factory = lambda p: KD()
    try:
        cell_t, sub_t, map_t = map_subcell_extract_props('', {'mi': '', 'vc': ''}, chunk_size=(32, 32, 16), min_obj_vx={'sv': 1, 'mi': 1, 'vc': 1},
                                                         device=gpu, as_tables=True, chunk_loader=loader)
    finally:
        sp.kd_factory = orig
    whole = segstats(vols['sv'], [vols['mi'], vols['vc']], device=gpu)
    assert np.array_equal(cell_t.ids, whole.cell[0]) and np.array_equal(cell_t.sizes, whole.cell[2])
    for i, o in enumerate(['mi', 'vc']):
        assert np.array_equal(sub_t[o].ids, whole.sub[i][0]) and np.array_equal(sub_t[o].sizes, whole.sub[i][2])
        lo = np.minimum.reduceat(sub_t[o].boxes[:, 0], sub_t[o].box_begin[:-1], axis=0)
        hi = np.maximum.reduceat(sub_t[o].boxes[:, 1], sub_t[o].box_begin[:-1], axis=0)
        assert np.array_equal(lo, whole.sub[i][3][:, 0]) and np.array_equal(hi, whole.sub[i][3][:, 1])
        s, c, n = whole.pairs[i]
        assert np.array_equal(map_t[o].sub_ids, s) and np.array_equal(map_t[o].cell_ids, c) and np.array_equal(map_t[o].counts, n)


@pytest.mark.gpu
def test_chunk_driver_repeats_the_pass_when_a_table_overflows(gpu):
    """The host reads a chunk's overflow flags two chunks late (it never waits for the chunk it has just queued): tables that start
    far too small for one chunk are detected, the whole pass is repeated with larger ones, and the result is the one of a pass that
    started large enough -- every voxel its own object in one of the chunks."""
    import torch
    import syconn_amd.proc.sd_proc as sp

    class KD:
        boundary = np.array([48, 16, 16])
    vol = np.zeros(tuple(KD.boundary), np.uint64)
    vol[16:32] = (np.arange(16 * 16 * 16, dtype=np.uint64) + 5).reshape(16, 16, 16)       # chunk 1 of 3: 4096 objects
    vol[40:44, 3:9, 2:7] = 3
    sub = (vol % np.uint64(7)) * np.uint64(11)
    dv, ds = torch.from_numpy(vol.view(np.int64)).to(gpu), torch.from_numpy(sub.view(np.int64)).to(gpu)

    def loader(name, off, size):
        src = dv if name == 'sv' else ds
        return src[off[0]:off[0] + size[0], off[1]:off[1] + size[1], off[2]:off[2] + size[2]].contiguous()
    orig = sp.kd_factory
    sp.kd_factory = lambda p: KD()
    try:
        kw = dict(chunk_size=(16, 16, 16), min_obj_vx={'sv': 1, 'mi': 1}, device=gpu, as_tables=True, chunk_loader=loader)
        small = sp.map_subcell_extract_props('', {'mi': ''}, table_capacity=1024, **kw)
        big = sp.map_subcell_extract_props('', {'mi': ''}, table_capacity=1 << 16, **kw)
    finally:
        sp.kd_factory = orig
    assert len(small[0]) == 4097 and np.array_equal(small[0].ids, big[0].ids) and np.array_equal(small[0].sizes, big[0].sizes)
    assert np.array_equal(small[0].boxes, big[0].boxes) and np.array_equal(small[0].rep_coords, big[0].rep_coords)
    for a, b in ((small[1]['mi'], big[1]['mi']),):
        assert np.array_equal(a.ids, b.ids) and np.array_equal(a.sizes, b.sizes) and np.array_equal(a.boxes, b.boxes)
    assert np.array_equal(small[2]['mi'].counts, big[2]['mi'].counts) and np.array_equal(small[2]['mi'].cell_ids, big[2]['mi'].cell_ids)
    assert int(small[0].sizes.sum()) == int(np.count_nonzero(vol))
