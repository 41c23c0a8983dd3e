"""Golden vectors for the dataset-wide merging of per-chunk object properties (the chunk driver):
``merge_prop_dicts``, ``merge_map_dicts``, ``convert_nvox2ratio_mapdict`` and ``invert_mdc`` are lifted by AST from
/root/reference/syconn/proc/sd_proc.py (:1248-1322) and executed here; the per-chunk loop around them
(``_map_subcell_extract_props_thread``, :617-678: boundary ids, the ``min_obj_vx`` filter, the merge calls) is applied as written
there, with the numpy restatement of the Cython native (oracle/objprops_ref.py, pinned by the reference's known-answer test) in
the place of ``map_subcell_extract_props_func``.  Only inputs and outputs are stored (dictionaries as pickled bytes).

    python tests/golden/make_golden_propmerge.py      ->  tests/golden/g12_propmerge.npz
"""
import ast
import os
import pickle
import sys
from collections import defaultdict
from typing import List, Optional

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.objprops_ref import map_subcell_extract_props_np      # noqa: E402

REF = '/root/reference/syconn'


def lift(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return [ns[n] for n in names]


def label_volume(shape, seed, sigma, q, n_max):
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), sigma)
    lab, n = ndimage.label(v > np.quantile(v, q))
    ids = rng.permutation(np.arange(1, n + 1)).astype(np.uint64) * 3 + 11          # sparse, unordered ids
    lut = np.concatenate([[0], ids]).astype(np.uint64)
    return lut[lab]


CASES = [
    # name, volume (x,y,z), chunk size, organelles {name: (seed, sigma, quantile)}, cell (seed, sigma, quantile), min_obj_vx
    ('two_organelles', (40, 36, 24), (20, 18, 12), {'mi': (1, 1.5, 0.8), 'vc': (2, 1.2, 0.85)}, (3, 1.6, 0.62),
     {'mi': 30, 'vc': 8, 'sv': 1}),
    ('ragged_sv_filter', (37, 29, 23), (16, 16, 16), {'sj': (4, 1.3, 0.8)}, (5, 1.4, 0.66), {'sj': 12, 'sv': 40}),
    ('one_chunk', (18, 16, 14), (32, 32, 32), {'mi': (6, 1.4, 0.75)}, (7, 2.5, 0.5), {'mi': 20, 'sv': 1}),
]


def main():
    ns = {'np': np, 'List': List, 'Optional': Optional}
    merge_prop_dicts, merge_map_dicts, nvox2ratio, invert_mdc = lift(
        f'{REF}/proc/sd_proc.py', ['merge_prop_dicts', 'merge_map_dicts', 'convert_nvox2ratio_mapdict', 'invert_mdc'], ns)
    out = {'names': np.array([c[0] for c in CASES])}
    for name, shape, cs, orgs, cell, min_obj_vx in CASES:
        shape, cs = np.array(shape), np.array(cs)
        cell_vol = label_volume(tuple(shape), *cell, 0)
        sub_vols = {k: label_volume(tuple(shape), *v, 0) for k, v in orgs.items()}
        grid = -(-shape // cs)
        padded = lambda a: np.pad(a, [(0, int(grid[i] * cs[i] - shape[i])) for i in range(3)])        # load_seg: zeros beyond the dataset
        cell_p, sub_p = padded(cell_vol), {k: padded(v) for k, v in sub_vols.items()}
        existing_oragnelles = list(orgs.keys())
        n_subcell = len(existing_oragnelles)
        cpd_lst = [{}, defaultdict(list), {}]
        scpd_lst = [[{}, defaultdict(list), {}] for _ in range(n_subcell)]
        scmd_lst = [{} for _ in range(n_subcell)]
        for x in range(0, int(grid[0] * cs[0]), int(cs[0])):
            for y in range(0, int(grid[1] * cs[1]), int(cs[1])):
                for z in range(0, int(grid[2] * cs[2]), int(cs[2])):
                    offset = np.array([x, y, z])
                    sl = tuple(slice(int(offset[i]), int(offset[i] + cs[i])) for i in range(3))
                    # ---- sd_proc.py:617-678, as written there ----
                    subcell_d = []
                    obj_ids_bdry = dict()
                    for organelle in existing_oragnelles:
                        subc_d = sub_p[organelle][sl]
                        obj_bdry = np.concatenate(
                            [subc_d[0].flat, subc_d[:, 0].flat, subc_d[:, :, 0].flat, subc_d[-1].flat,
                             subc_d[:, -1].flat, subc_d[:, :, -1].flat])
                        obj_bdry = np.unique(obj_bdry)
                        obj_ids_bdry[organelle] = obj_bdry
                        subcell_d.append(subc_d[None,])
                    subcell_d = np.concatenate(subcell_d)
                    cell_d = cell_p[sl]
                    cell_prop_dicts, subcell_prop_dicts, subcell_mapping_dicts = map_subcell_extract_props_np(cell_d, subcell_d)
                    if min_obj_vx['sv'] > 1:
                        obj_bdry = np.concatenate(
                            [cell_d[0].flat, cell_d[:, 0].flat, cell_d[:, :, 0].flat, cell_d[-1].flat,
                             cell_d[:, -1].flat, cell_d[:, :, -1].flat])
                        obj_bdry = set(np.unique(obj_bdry))
                        obj_inside = set(list(cell_prop_dicts[0].keys())).difference(obj_bdry)
                        for ix in obj_inside:
                            if cell_prop_dicts[2][ix] < min_obj_vx['sv']:
                                del cell_prop_dicts[0][ix], cell_prop_dicts[1][ix], cell_prop_dicts[2][ix]
                    merge_prop_dicts([cpd_lst, cell_prop_dicts], offset)
                    del cell_prop_dicts
                    subcell_prop_dicts = [[subcell_prop_dicts[0][ii], subcell_prop_dicts[1][ii],
                                           subcell_prop_dicts[2][ii]] for ii in range(n_subcell)]
                    for ii, organelle in enumerate(existing_oragnelles):
                        if min_obj_vx[organelle] > 1:
                            obj_bdry = obj_ids_bdry[organelle]
                            obj_inside = set(list(subcell_prop_dicts[ii][0].keys())).difference(obj_bdry)
                            for ix in obj_inside:
                                if subcell_prop_dicts[ii][2][ix] < min_obj_vx[organelle]:
                                    del subcell_prop_dicts[ii][0][ix], subcell_prop_dicts[ii][1][ix]
                                    del subcell_prop_dicts[ii][2][ix]
                                    if ix in subcell_mapping_dicts[ii]:
                                        del subcell_mapping_dicts[ii][ix]
                        merge_map_dicts([scmd_lst[ii], subcell_mapping_dicts[ii]])
                        merge_prop_dicts([scpd_lst[ii], subcell_prop_dicts[ii]], offset)
        plain = lambda t: [{int(k): v for k, v in t[0].items()}, {int(k): v for k, v in t[1].items()}, {int(k): int(v) for k, v in t[2].items()}]
        res = {'cell': plain(cpd_lst), 'sub': {k: plain(scpd_lst[i]) for i, k in enumerate(existing_oragnelles)},
               'maps': {k: {int(a): {int(b): int(c) for b, c in d.items()} for a, d in scmd_lst[i].items()}
                        for i, k in enumerate(existing_oragnelles)}}
        import copy
        ratio = copy.deepcopy(res['maps'])
        for k in ratio:
            nvox2ratio(ratio[k])
        res['ratio'] = {k: {a: {b: float(c) for b, c in d.items()} for a, d in ratio[k].items()} for k in ratio}
        res['inverted'] = {k: invert_mdc(res['maps'][k]) for k in res['maps']}
        out.update({f'{name}_cell': cell_vol, f'{name}_chunk_size': cs, f'{name}_min_obj_vx': np.frombuffer(pickle.dumps(min_obj_vx), np.uint8),
                    f'{name}_expected': np.frombuffer(pickle.dumps(res), np.uint8), f'{name}_organelles': np.array(existing_oragnelles)})
        for k, v in sub_vols.items():
            out[f'{name}_sub_{k}'] = v
        print(name, 'cell objects', len(res['cell'][2]), {k: len(res['sub'][k][2]) for k in res['sub']},
              'mapped', {k: len(res['maps'][k]) for k in res['maps']})
    np.savez_compressed(os.path.join(HERE, 'g12_propmerge.npz'), **out)


if __name__ == '__main__':
    main()
