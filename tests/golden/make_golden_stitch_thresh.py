"""Golden vectors for the ``overlap_thresh > 0`` branch of ``_make_stitch_list_thread`` (object_extraction_steps.py:597-615: a pair of
touching ids is kept only if more than 10 % of the two objects' voxels coincide, counted with a cKDTree over their global voxel
coordinates), produced by the reference's own function on the inputs of tests/golden/g11_stitch.npz (same cases, same seeds: the
generator of g11 is imported).  Only the resulting pair lists are stored.

    python tests/golden/make_golden_stitch_thresh.py      ->  tests/golden/g14_stitch_thresh.npz
"""
import os
import pickle as pkl
import sys
import tempfile

import networkx as nx
import numpy as np
import scipy.ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_stitch as m      # noqa: E402


def main():
    ns = {'np': np, 'nx': nx, 'pkl': pkl}
    store = m.Store()
    ns['compression'] = store
    holder = {}

    class Chunky:
        @staticmethod
        def load_dataset(path):
            return holder['cset']
    ns['chunky'] = Chunky
    m.lift(f'{m.REF}/proc/general.py', ['cut_array_in_one_dim'], ns)
    uniq_t, stitch_t = m.lift(f'{m.REF}/extraction/object_extraction_steps.py', ['_make_unique_labels_thread', '_make_stitch_list_thread'], ns)
    g11 = np.load(os.path.join(HERE, 'g11_stitch.npz'))
    out = {'names': np.array([c[0] for c in m.CASES])}
    tmp = tempfile.mkdtemp()
    for name, grid, cs, ol, so, seed, sigma, level in m.CASES:
        cs, ol, so = np.array(cs), np.array(ol), np.array(so)
        cset = m.CSet(grid, cs, ol, f'{tmp}/{name}/')
        holder['cset'] = cset
        chunk_list = list(cset.chunk_dict)
        hdf5names, filename, suffix = ['obj'], 'seg', ''
        for n, ch in cset.chunk_dict.items():          # the per-chunk component labels of g11 (inputs)
            store.save_to_h5py([g11[f'{name}_labels_{n}']], ch.folder + filename + "_connected_components%s.h5" % suffix, hdf5names)
        offs = g11[f'{name}_offsets']
        uniq_t([[cset.chunk_dict[n], filename, hdf5names, {'obj': int(offs[n])}, suffix] for n in chunk_list])
        res = stitch_t([cset.path_head_folder, chunk_list, filename, hdf5names, so, ol, suffix, chunk_list, 1])      # overlap_thresh = 1
        pairs = sorted(tuple(int(v) for v in p) for p in res['obj'])
        plain = [tuple(p) for p in g11[f'{name}_pairs'].tolist()]
        assert set(pairs) <= set(plain)
        out[f'{name}_pairs_thresh'] = np.array(pairs, dtype=np.int64).reshape(-1, 2)
        print(name, 'pairs without threshold', len(plain), 'with', len(pairs))
    np.savez_compressed(os.path.join(HERE, 'g14_stitch_thresh.npz'), **out)


if __name__ == '__main__':
    main()
