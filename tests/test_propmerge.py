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
