"""Prototype (CPU, pure Python) of the level-synchronous formulation of the marker flood that csrc/sd_objseg.hip::k_ws_flood runs
per mask component, checked against the sequential restatement oracle/objseg_ref.py::watershed_ref.

Formulation.  The sequential flood pops (value, age)-ordered; with level = squared distance (higher first) its pop sequence is:
take the FIFO of the highest non-empty level W, generation by generation (a generation = the FIFO's content when its first
element is popped; pushes at level W form the next generation).  A popped element e_i ("block" i) labels its unlabelled
neighbours; neighbours ABOVE W start a cascade that floods the whole connected set of unlabelled voxels above W it touches
before e_{i+1} is popped, labelling that set's unlabelled rim as well.  Hence, per generation:
  * every unlabelled voxel at or below W next to a generation element, or next to a cascade region, goes to the block with the
    smallest index among those claimers; a cascade region belongs to the smallest-index element touching it;
  * pushes are ordered by block index; the order INSIDE a block is free (all its elements carry one label and stay contiguous
    in every FIFO, so by induction no comparison between different labels ever depends on it) -- randomised here.
usage: python tools/experiments/ws_levelsync_proto.py [n_cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.objseg_ref import watershed_ref                                                    # noqa: E402


def flood_levelsync(d2, markers, mask, rng):
    sh = mask.shape
    out = np.where(mask != 0, markers, 0).astype(np.int64).ravel()
    m = (mask != 0).ravel()
    lvl = d2.ravel().astype(np.int64)
    sx, sy = sh[1] * sh[2], sh[2]

    def nbrs(idx):
        x, r = divmod(idx, sx)
        y, z = divmod(r, sy)
        for ok, q in ((x > 0, idx - sx), (y > 0, idx - sy), (z > 0, idx - 1), (z + 1 < sh[2], idx + 1),
                      (y + 1 < sh[1], idx + sy), (x + 1 < sh[0], idx + sx)):
            if ok and m[q]:
                yield q

    # bag of queued elements: (level, T, voxel); markers: T = raster index, pushed: T = 2^31 + block counter
    bag = [(int(lvl[i]), int(i), int(i)) for i in np.flatnonzero(out) if any(out[q] == 0 for q in nbrs(int(i)))]
    tbase = 1 << 31
    while bag:
        W = max(b[0] for b in bag)
        A = sorted((b[1], b[2]) for b in bag if b[0] == W)
        bag = [b for b in bag if b[0] != W]
        while A:
            claim = {}
            region = []                                     # cascade voxels (level > W), in discovery order
            for i, (_, v) in enumerate(A):
                for q in nbrs(v):
                    if out[q] == 0:
                        if q not in claim and lvl[q] > W:
                            region.append(q)
                        claim[q] = min(claim.get(q, i), i)
            # cascade: min-owner propagation inside the unlabelled set above W (label-correcting until stable)
            k = 0
            while k < len(region):
                r = region[k]
                k += 1
                for q in nbrs(r):
                    if out[q] == 0 and lvl[q] > W and q not in claim:
                        claim[q] = claim[r]
                        region.append(q)
            changed = True
            while changed:
                changed = False
                for r in region:
                    for q in nbrs(r):
                        if out[q] == 0 and lvl[q] > W and claim[q] > claim[r]:
                            claim[q] = claim[r]
                            changed = True
            for r in region:                                # rim of the cascade regions
                for q in nbrs(r):
                    if out[q] == 0 and lvl[q] <= W:
                        claim[q] = min(claim.get(q, claim[r]), claim[r])
            new = []
            items = list(claim.items())
            rng.shuffle(items)                              # order inside a block is free
            for q, w in items:
                out[q] = out[A[w][1]]
                if lvl[q] > W:
                    continue                                # cascade voxel: popped inside the block, never queued
                if lvl[q] == W:
                    new.append((tbase + w, q))
                else:
                    bag.append((int(lvl[q]), tbase + w, q))
            tbase += len(A)
            # sort by block; ties (one block) in the shuffled order
            new.sort(key=lambda t: t[0])
            A = new
    return out.reshape(sh).astype(np.int32)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(0)
    bad = 0
    for case in range(n_cases):
        sh = tuple(int(v) for v in rng.integers(3, 13, 3))
        mask = rng.random(sh) < rng.uniform(0.5, 1.0)
        kind = case % 3
        if kind == 0:                                       # few distinct levels: many ties and plateaus
            d2 = rng.integers(1, 4, sh)
        elif kind == 1:                                     # rough landscape: many cascades
            d2 = rng.integers(1, 50, sh)
        else:                                               # distance-like
            from scipy import ndimage
            d2 = np.rint(ndimage.distance_transform_edt(mask, sampling=(1, 1, 2)) ** 2).astype(np.int64)
            d2[~mask] = 0
        d2 = np.where(mask, d2, 0).astype(np.int64)
        markers = np.zeros(sh, np.int32)
        n_mk = int(rng.integers(2, 7))
        pts = np.flatnonzero(mask.ravel())
        if pts.size < n_mk:
            continue
        for lab, p in enumerate(rng.choice(pts, n_mk, replace=False), start=1):
            markers.ravel()[p] = lab
            if rng.random() < 0.5:                          # a marker of several voxels
                for q in (p + 1, p + sh[2]):
                    if q < mask.size and mask.ravel()[q] and markers.ravel()[q] == 0:
                        markers.ravel()[q] = lab
        want = watershed_ref(d2, markers, mask.astype(np.uint8))
        got = flood_levelsync(d2, markers, mask.astype(np.uint8), rng)
        if not np.array_equal(want, got):
            bad += 1
            print('MISMATCH case', case, sh, 'kind', kind, int((want != got).sum()), 'voxels')
    print(f'{n_cases} cases, {bad} mismatches')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
