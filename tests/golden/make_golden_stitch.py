"""Golden vectors for the steps that make per-chunk object labels globally unique (behind the first stage),
produced by the REFERENCE'S OWN code: ``_make_unique_labels_thread``, ``_make_stitch_list_thread``, ``make_merge_list`` and
``_apply_merge_list_thread`` are lifted by AST from /root/reference/syconn/extraction/object_extraction_steps.py (:425-443, :531-617,
:620-655, :706-736) and ``cut_array_in_one_dim`` from /root/reference/syconn/proc/general.py (:45-82), and executed here with real
networkx and with in-memory stand-ins for what they import besides numpy: ``compression.load_from_h5py / save_to_h5py`` (a dict
instead of h5 files) and ``chunky.load_dataset`` (a grid object whose ``get_neighbouring_chunks`` answers the six face neighbours,
-1 where there is none -- knossos_utils is absent, its enumeration order does not matter to the caller, which sorts by direction).
The label offsets are the inline statements of object_extraction_wrapper.py:296-312, applied as written.  Only inputs and outputs
are stored.

    python tests/golden/make_golden_stitch.py      ->  tests/golden/g11_stitch.npz
"""
import ast
import os
import pickle as pkl
import sys
import tempfile

import networkx as nx
import numpy as np
import scipy.ndimage
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/syconn'


def lift(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return [ns[n] for n in names]


class Store:                                   # compression.load_from_h5py / save_to_h5py on a dict
    def __init__(self):
        self.d = {}

    def load_from_h5py(self, path, hdf5names, *a, **k):
        return [self.d[(path, n)].copy() for n in hdf5names]

    def save_to_h5py(self, data, path, hdf5names, *a, **k):
        for arr, n in zip(data, hdf5names):
            self.d[(path, n)] = np.array(arr)


class Chunk:
    def __init__(self, number, coordinates, size, overlap, folder):
        self.number, self.coordinates, self.size, self.overlap, self.folder = number, np.array(coordinates), np.array(size), np.array(overlap), folder


class CSet:
    def __init__(self, grid, chunk_size, overlap, head):
        self.path_head_folder = head
        self.chunk_dict, self.pos = {}, {}
        n = 0
        for x in range(grid[0]):
            for y in range(grid[1]):
                for z in range(grid[2]):
                    self.chunk_dict[n] = Chunk(n, np.array([x, y, z]) * chunk_size, chunk_size, overlap, f'{head}chunk{n}/')
                    self.pos[n] = (x, y, z)
                    n += 1
        self.by_pos = {p: k for k, p in self.pos.items()}

    def get_neighbouring_chunks(self, chunk, chunklist=None, con_mode=7):
        dirs = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
        nb = []
        for d in dirs:
            k = self.by_pos.get(tuple(np.array(self.pos[chunk.number]) + d), -1)
            nb.append(k if (chunklist is None or k in chunklist) else -1)
        return np.array(nb), dirs


def blobs(shape, seed, sigma, level):
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), sigma)
    return v > np.quantile(v, level)


CASES = [
    # name, grid, chunk size, overlap, stitch overlap, seed, sigma, quantile
    ('grid222', (2, 2, 2), (24, 20, 18), (2, 2, 1), (2, 2, 1), 1, 2.0, 0.72),
    ('grid321_thin', (3, 2, 1), (16, 18, 20), (4, 4, 2), (1, 1, 1), 2, 1.6, 0.70),
    ('grid113', (1, 1, 3), (20, 22, 12), (2, 2, 2), (2, 2, 1), 3, 2.5, 0.65),
    ('single', (1, 1, 1), (14, 12, 10), (2, 2, 1), (1, 1, 1), 4, 1.5, 0.7),
    ('dense223', (2, 2, 3), (12, 12, 10), (3, 3, 2), (2, 2, 2), 5, 3.0, 0.45),
]


def main():
    ns = {'np': np, 'nx': nx, 'pkl': pkl}
    store = Store()
    ns['compression'] = store
    holder = {}

    class Chunky:
        @staticmethod
        def load_dataset(path):
            return holder['cset']
    ns['chunky'] = Chunky
    (cut,) = lift(f'{REF}/proc/general.py', ['cut_array_in_one_dim'], ns)
    uniq_t, stitch_t, merge_l, apply_t = lift(f'{REF}/extraction/object_extraction_steps.py',
                                              ['_make_unique_labels_thread', '_make_stitch_list_thread', 'make_merge_list',
                                               '_apply_merge_list_thread'], ns)
    out = {'names': np.array([c[0] for c in CASES])}
    tmp = tempfile.mkdtemp()
    for name, grid, cs, ol, so, seed, sigma, level in CASES:
        cs, ol, so = np.array(cs), np.array(ol), np.array(so)
        vol_shape = np.array(grid) * cs
        mask = blobs(tuple(vol_shape), seed, sigma, level)
        padded = np.zeros(tuple(vol_shape + 2 * ol), dtype=bool)             # zeros beyond the dataset, like kd.load_raw
        padded[ol[0]:ol[0] + vol_shape[0], ol[1]:ol[1] + vol_shape[1], ol[2]:ol[2] + vol_shape[2]] = mask
        cset = CSet(grid, cs, ol, f'{tmp}/{name}/')
        holder['cset'] = cset
        chunk_list = list(cset.chunk_dict)
        hdf5names, filename, suffix = ['obj'], 'seg', ''
        nb_cc = np.zeros(len(chunk_list), dtype=np.int32)
        for n, ch in cset.chunk_dict.items():
            c = ch.coordinates
            sub = padded[c[0]:c[0] + cs[0] + 2 * ol[0], c[1]:c[1] + cs[1] + 2 * ol[1], c[2]:c[2] + cs[2] + 2 * ol[2]]
            lab, nmax = scipy.ndimage.label(sub)                              # (what the first stage writes: *_connected_components.h5)
            store.save_to_h5py([lab.astype(np.int32)], ch.folder + filename + "_connected_components%s.h5" % suffix, hdf5names)
            nb_cc[n] = nmax
            out[f'{name}_labels_{n}'] = lab.astype(np.int32)
        # object_extraction_wrapper.py:300-312
        max_nb = np.zeros(len(chunk_list), dtype=np.int32)
        for nb_chunk in range(1, len(chunk_list)):
            max_nb[nb_chunk] = max_nb[nb_chunk - 1] + nb_cc[nb_chunk - 1]
        max_label = int(max_nb[-1] + nb_cc[-1])
        uniq_t([[cset.chunk_dict[n], filename, hdf5names, {'obj': int(max_nb[n])}, suffix] for n in chunk_list])      # (a Python int: numpy 2 refuses uint64 += np.int32, the reference's numpy 1 added it)
        res = stitch_t([cset.path_head_folder, chunk_list, filename, hdf5names, so, ol, suffix, chunk_list, 0])
        pairs = sorted(tuple(int(v) for v in p) for p in res['obj'])
        merge_dict, merge_list_dict = merge_l(hdf5names, {'obj': [tuple(p) for p in pairs]}, {'obj': max_label})
        mp = f'{tmp}/{name}_merge.pkl'
        with open(mp, 'wb') as f:
            pkl.dump(merge_list_dict, f)
        apply_t([[cset.chunk_dict[n] for n in chunk_list], filename, hdf5names, mp, suffix])
        ml = np.asarray(merge_list_dict['obj']).astype(np.uint64)
        # canonical representative = smallest id of the component (the reference's is an arbitrary member)
        canon = np.arange(max_label + 1, dtype=np.uint64)
        for comp in {}.fromkeys(ml.tolist()):
            members = np.nonzero(ml == comp)[0]
            canon[members] = members.min()
        assert np.array_equal(canon[ml.astype(np.int64)], canon)              # representatives are members of their component
        out.update({f'{name}_grid': np.array(grid), f'{name}_chunk_size': cs, f'{name}_overlap': ol, f'{name}_stitch_overlap': so,
                    f'{name}_nb_cc': nb_cc.astype(np.int64), f'{name}_offsets': max_nb.astype(np.int64), f'{name}_max_label': np.int64(max_label),
                    f'{name}_pairs': np.array(pairs, dtype=np.int64).reshape(-1, 2), f'{name}_canon': canon})
        for n, ch in cset.chunk_dict.items():
            u = store.d[(ch.folder + filename + "_unique_components%s.h5" % suffix, 'obj')]
            st = store.d[(ch.folder + filename + "_stitched_components%s.h5" % suffix, 'obj')]
            assert u.dtype == np.uint64 and st.shape == tuple(cs)
            out[f'{name}_unique_{n}'] = u
            out[f'{name}_stitched_canon_{n}'] = canon[st.astype(np.int64)]
        print(name, 'chunks', len(chunk_list), 'components', max_label, 'pairs', len(pairs), 'objects after merging',
              len(np.unique(canon[1:])) if max_label else 0)
    np.savez_compressed(os.path.join(HERE, 'g11_stitch.npz'), **out)


if __name__ == '__main__':
    main()
