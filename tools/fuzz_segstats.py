"""GPU box: randomized sweep of the label-statistics pass -- every kernel form (one voxel per lane / four voxels per lane, sequential or with
1-4 volumes prefetched, with or without a cell volume, properties on / off, uint32 / uint64) against the numpy oracle and against each other.
usage: fuzz_segstats.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.objprops_ref import _props_np
from syconn_amd.extraction.find_object_properties import segstats

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda', 0)


def volume(shape, nid, dtype, coherent):
    if coherent:
        small = rng.integers(0, nid, [max(1, s // 4 + 1) for s in shape])
        v = np.kron(small, np.ones((4, 4, 4), dtype=np.int64))[:shape[0], :shape[1], :shape[2]]
    else:
        v = rng.integers(0, nid, shape)
    if rng.random() < 0.3:
        v = v * (rng.random(shape) < 0.5)
    return np.ascontiguousarray(v).astype(dtype)


def pairs_np(cell, sub):
    both = np.flatnonzero((sub.reshape(-1) != 0) & (cell.reshape(-1) != 0))
    if both.size == 0:
        return np.zeros(0, sub.dtype), np.zeros(0, cell.dtype), np.zeros(0, np.int64)
    pr, cnt = np.unique(np.stack((sub.reshape(-1)[both], cell.reshape(-1)[both]), axis=1), axis=0, return_counts=True)
    return pr[:, 0], pr[:, 1], cnt


def same(a, b):
    return all(np.array_equal(np.asarray(x).astype(np.uint64) if np.asarray(x).dtype.kind in 'ui' else x, np.asarray(y).astype(np.uint64) if np.asarray(y).dtype.kind in 'ui' else y)
               for x, y in zip(a, b))


t0, n, forms = time.time(), 0, {}
while time.time() - t0 < budget:
    z = int(rng.choice([4, 8, 12, 64, 100, 256, 260, 512, 5, 33]))
    shape = (int(rng.integers(1, 24)), int(rng.integers(1, 40)), z)
    dtype = np.uint64 if rng.random() < 0.6 else np.uint32
    nid = int(rng.choice([2, 5, 40, 3000]))
    coh = bool(rng.random() < 0.6)
    has_cell = bool(rng.random() < 0.7)
    n_sub = int(rng.integers(0 if has_cell else 1, 6))
    want = bool(rng.random() < 0.8) or not has_cell or n_sub == 0
    cell = volume(shape, nid, dtype, coh) if has_cell else None
    subs = [volume(shape, max(2, nid // 2), dtype, coh and k % 2 == 0) for k in range(n_sub)]
    if dtype == np.uint64 and has_cell:
        cell[cell == 1] = np.uint64(2 ** 63 + 5)
    os.environ.pop('SD_SEGSTATS_V1', None)
    r4 = segstats(cell, subs, want_props=want, device=dev)
    os.environ['SD_SEGSTATS_V1'] = '1'
    r1 = segstats(cell, subs, want_props=want, device=dev)
    os.environ.pop('SD_SEGSTATS_V1', None)
    key = (z % 4 == 0, has_cell, min(n_sub, 5), want, dtype.__name__)
    forms[key] = forms.get(key, 0) + 1
    if want:
        if has_cell:
            assert same(r4.cell, _props_np(cell)) and same(r1.cell, r4.cell), ('cell', shape, key)
        for k in range(n_sub):
            assert same(r4.sub[k], _props_np(subs[k])) and same(r1.sub[k], r4.sub[k]), ('sub', k, shape, key)
    if has_cell:
        for k in range(n_sub):
            assert same(r4.pairs[k], pairs_np(cell, subs[k])) and same(r1.pairs[k], r4.pairs[k]), ('pairs', k, shape, key)
    n += 1
print(f'fuzz_segstats: {n} cases ok in {time.time() - t0:.0f} s; forms hit: {len(forms)}')
for k in sorted(forms):
    print('  rows % 4 == 0:', k[0], ' cell:', k[1], ' subs:', k[2], ' props:', k[3], k[4], ' x', forms[k])
