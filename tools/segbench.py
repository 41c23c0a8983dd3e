"""Throughput of the post-inference stages on the device next to their CPU restatements
(GPU box): label-volume statistics (find_object_properties / map_subcell_extract_props) and the first stage of the object
segmentation (threshold -> morphology -> connected components).  Both are HBM-bound streams: reports achieved GB/s of
ALGORITHMIC bytes (each input byte read once, each output byte written once) against the 8 TB/s HBM peak.
usage: tools/segbench.py [edge=512]"""
import json
import os
import sys
import time

import numpy as np
import torch
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.objprops_ref import find_object_properties_np, map_subcell_extract_props_np      # noqa: E402
from oracle.objseg_ref import object_segmentation_ref                                        # noqa: E402
from syconn_amd.extraction import find_object_properties as fop                              # noqa: E402
from syconn_amd.extraction.object_extraction_steps import object_segmentation_first_stage    # noqa: E402


def sync_time(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    # supervoxel-like cell segmentation (blocky ids) and sparse organelle labels
    small = rng.integers(1, 4000, (n // 16 + 1,) * 3)
    cell = np.kron(small, np.ones((16, 16, 16), np.int64))[:n, :n, :n].astype(np.uint64)
    prob = ndimage.gaussian_filter(rng.random((n // 4, n // 4, n // 4)).astype(np.float32), 1.5)
    prob = np.kron(prob, np.ones((4, 4, 4), np.float32))[:n, :n, :n]
    prob = ((prob - prob.min()) / (prob.max() - prob.min()) * 255).astype(np.uint8)
    thr = float(np.quantile(prob[::4, ::4, ::4], 0.9))
    out = {'edge': n, 'voxels': n ** 3}

    # ---- object segmentation first stage --------------------------------------------------------------------------
    p_dev = torch.from_numpy(prob).to(dev)
    for ops in ([], ['binary_closing', 'binary_opening']):
        t = sync_time(lambda: object_segmentation_first_stage(p_dev, thr, ops), reps=2)
        t_dev = sync_time(lambda: object_segmentation_first_stage(p_dev, thr, ops, return_device=True), reps=5)
        lab, mx = object_segmentation_first_stage(p_dev, thr, ops)
        t0 = time.perf_counter()
        if n <= 256:
            want, wmx = object_segmentation_ref(prob, thr, ops, (10, 10, 20))
            tc = time.perf_counter() - t0
            assert wmx == mx and np.array_equal(want, lab)
        else:                                    # CPU restatement on a 256^3 corner, scaled per voxel
            sub = np.ascontiguousarray(prob[:256, :256, :256])
            object_segmentation_ref(sub, thr, ops, (10, 10, 20))
            tc = (time.perf_counter() - t0) * (n / 256) ** 3
        alg = n ** 3 * (1 + 4)                   # uint8 probability in, int32 labels out
        out['objseg_' + ('morph' if ops else 'plain')] = {
            'ops': ops, 'components': mx, 'gpu_ms_incl_d2h_of_labels': t * 1e3, 'gpu_ms_device_resident': t_dev * 1e3,
            'gpu_Mvox_s': n ** 3 / t_dev / 1e6, 'algorithmic_GB_s': alg / t_dev / 1e9, 'cpu_s_scipy_oracle' + ('' if n <= 256 else '_extrapolated'): tc,
            'cpu_Mvox_s': n ** 3 / tc / 1e6}

    # ---- the default-config watershed branch (config.yml:130-136 'sj' / 'vc' list; object_extraction_steps.py:319-352) --------------
    ws_ops = ['binary_opening', 'binary_closing', 'binary_erosion']
    t_ws = sync_time(lambda: object_segmentation_first_stage(p_dev, thr, ws_ops, return_device=True, min_seed_vx=10), reps=3)
    lab_ws, mx_ws = object_segmentation_first_stage(p_dev, thr, ws_ops, min_seed_vx=10)
    out['objseg_watershed'] = {'ops': ws_ops, 'min_seed_vx': 10, 'labels': mx_ws, 'gpu_ms_device_resident': t_ws * 1e3,
                               'gpu_Mvox_s': n ** 3 / t_ws / 1e6, 'labelled_voxels': int((lab_ws > 0).sum())}
    os.environ['SD_WS_SEQUENTIAL'] = '1'                         # the sequential device restatement: cross-check + its time
    t_seq = sync_time(lambda: object_segmentation_first_stage(p_dev, thr, ws_ops, return_device=True, min_seed_vx=10), reps=1)
    lab_seq, mx_seq = object_segmentation_first_stage(p_dev, thr, ws_ops, min_seed_vx=10)
    del os.environ['SD_WS_SEQUENTIAL']
    assert mx_seq == mx_ws and np.array_equal(lab_seq, lab_ws)
    out['objseg_watershed']['gpu_ms_sequential_flood_kernel'] = t_seq * 1e3

    # the same branch on an organelle-like volume: ~3000 separate ellipsoids (radius 5-14 voxels), about a third of them touching
    # a neighbour -- many small mask components, few markers each (what thresholded mi / vc / sj maps look like)
    sph = np.zeros((n, n, n), np.uint8)
    r2 = np.random.default_rng(5)
    for _ in range(int(3000 * (n / 512) ** 3)):
        c = r2.integers(16, n - 16, 3)
        r = r2.integers(5, 15)
        if r2.random() < 0.35:
            c2 = np.clip(c + r2.integers(-r, r + 1, 3) * 1.4, 16, n - 17).astype(int)
            centres = (c, c2)
        else:
            centres = (c,)
        for cc in centres:
            lo, hi = np.maximum(cc - r, 0), np.minimum(cc + r + 1, n)
            g = np.ogrid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
            m = ((g[0] - cc[0]) ** 2 + (g[1] - cc[1]) ** 2 + ((g[2] - cc[2]) * 1.6) ** 2) <= r * r
            sph[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]][m] = 255
    s_dev2 = torch.from_numpy(sph).to(dev)
    t_ws2 = sync_time(lambda: object_segmentation_first_stage(s_dev2, 127.5, ws_ops, return_device=True, min_seed_vx=10), reps=3)
    lab2, mx2 = object_segmentation_first_stage(s_dev2, 127.5, ws_ops, min_seed_vx=10)
    plain2, pmx2 = object_segmentation_first_stage(s_dev2, 127.5, ws_ops[:2])
    os.environ['SD_WS_SEQUENTIAL'] = '1'
    t_seq2 = sync_time(lambda: object_segmentation_first_stage(s_dev2, 127.5, ws_ops, return_device=True, min_seed_vx=10), reps=1)
    lab2s, mx2s = object_segmentation_first_stage(s_dev2, 127.5, ws_ops, min_seed_vx=10)
    del os.environ['SD_WS_SEQUENTIAL']
    assert mx2s == mx2 and np.array_equal(lab2s, lab2)
    out['objseg_watershed_organelle_like'] = {'ops': ws_ops, 'min_seed_vx': 10, 'mask_components_after_opening_closing': pmx2,
                                              'labels': mx2, 'gpu_ms_device_resident': t_ws2 * 1e3,
                                              'gpu_ms_sequential_flood_kernel': t_seq2 * 1e3,
                                              'gpu_Mvox_s': n ** 3 / t_ws2 / 1e6, 'labelled_voxels': int((lab2 > 0).sum())}

    # ---- label-volume statistics ---------------------------------------------------------------------------------------
    lab64 = lab.astype(np.uint64)
    c_dev = torch.from_numpy(cell.view(np.int64)).to(dev)
    s_dev = torch.from_numpy(lab64.view(np.int64)).to(dev)
    t1 = sync_time(lambda: fop.segstats(c_dev), reps=3)
    t2 = sync_time(lambda: fop.segstats(c_dev, [s_dev]), reps=3)
    t0 = time.perf_counter()
    k = min(n, 256)
    ref = find_object_properties_np(cell[:k, :k, :k])
    tc1 = (time.perf_counter() - t0) * (n / k) ** 3
    t0 = time.perf_counter()
    map_subcell_extract_props_np(cell[:k, :k, :k], lab64[None, :k, :k, :k])
    tc2 = (time.perf_counter() - t0) * (n / k) ** 3
    got = fop.find_object_properties(cell[:k, :k, :k])
    assert got == ref
    out['find_object_properties'] = {'objects': len(fop.segstats(c_dev).cell[0]), 'gpu_ms': t1 * 1e3,
                                     'algorithmic_GB_s': n ** 3 * 8 / t1 / 1e9, 'gpu_Mvox_s': n ** 3 / t1 / 1e6,
                                     'cpu_s_numpy_oracle_extrapolated': tc1, 'cpu_Mvox_s': n ** 3 / tc1 / 1e6}
    out['map_subcell_extract_props_1sub'] = {'gpu_ms': t2 * 1e3, 'algorithmic_GB_s': n ** 3 * 16 / t2 / 1e9,
                                             'gpu_Mvox_s': n ** 3 / t2 / 1e6, 'cpu_s_numpy_oracle_extrapolated': tc2,
                                             'cpu_Mvox_s': n ** 3 / tc2 / 1e6}
    print(json.dumps(out, indent=1))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/segbench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
