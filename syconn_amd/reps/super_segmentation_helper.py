"""Drop-in for the one function of ``syconn.reps.super_segmentation_helper`` that consumes the dense path's output
directly: ``map_myelin2coords`` (/root/reference/syconn/reps/super_segmentation_helper.py:550-615; SURVEY.md
section 8f row 3).  Same name, arguments, return value and error behaviour; the per-node ``kd.load_raw`` + numpy
reduction of the reference (one 11x11x5 box read per skeleton node) becomes: read the region the nodes cover once,
keep it on the GPU, one wave per node (``sd_box_majority``).  No CPU fallback.
"""
import os

import numpy as np
import torch

from .. import _lib as L
from .. import global_params
from ..handler.basics import kd_factory

_REGION_VOX = 768          # nodes are processed in spatial buckets of this many mag-voxels per axis (<= ~0.5 GB each)


def box_majority_device(vol: torch.Tensor, origins_zyx: torch.Tensor, edge_zyx, thresh_proba: float,
                        thresh_majority: float) -> torch.Tensor:
    """out[i] = (count(vol[box_i] > thresh_proba) / prod(edge) > thresh_majority) on the device; boxes may leave the
    volume (zeros outside).  vol: (D,H,W) uint8, origins_zyx: (n,3) int32, both on the same ROCm device."""
    lib = L.load()
    if not (vol.is_cuda and origins_zyx.is_cuda):
        raise RuntimeError('box_majority_device needs device tensors (there is no CPU fallback)')
    assert vol.dtype == torch.uint8 and vol.dim() == 3 and vol.is_contiguous()
    origins_zyx = origins_zyx.to(torch.int32).contiguous()
    n = int(origins_zyx.shape[0])
    out = torch.empty((n,), dtype=torch.uint8, device=vol.device)
    D, H, W = (int(v) for v in vol.shape)
    L.check(lib.sd_box_majority(vol.data_ptr(), D, H, W, origins_zyx.data_ptr(), n, int(edge_zyx[0]), int(edge_zyx[1]),
                                int(edge_zyx[2]), float(thresh_proba), float(thresh_majority), out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), 'sd_box_majority')
    return out


def map_myelin2coords(coords: np.ndarray, cube_edge_avg: np.ndarray = np.array([11, 11, 5]),
                      thresh_proba: float = 255 // 2, thresh_majority: float = 0.5, mag: int = 4) -> np.ndarray:
    """Myelin prediction (0 / 1, uint8) at every coordinate (mag-1 voxels, x,y,z): majority of ``myelin > thresh_proba``
    inside a box of `cube_edge_avg` mag-`mag` voxels around it, read from
    ``<working_dir>/knossosdatasets/myelin/`` (super_segmentation_helper.py:550-615)."""
    myelin_kd_p = global_params.config.working_dir + "/knossosdatasets/myelin/"
    if not os.path.isdir(myelin_kd_p):
        raise ValueError(f'Could not find myelin KnossosDataset at {myelin_kd_p}.')
    if not torch.cuda.is_available():
        raise RuntimeError('syconn_amd.map_myelin2coords needs an MI355X (there is no CPU fallback)')
    kd = kd_factory(myelin_kd_p)
    coords = np.asarray(coords)
    preds = np.zeros((len(coords)), dtype=np.uint8)
    if len(coords) == 0:
        return preds
    edge = np.asarray(cube_edge_avg, dtype=np.int64)
    # reference: offset = c - (edge*mag)//2 (mag-1 voxels); kd.load_raw(size=edge*mag, offset, mag) reads `edge`
    # voxels from floor(offset / mag) in the mag-`mag` volume
    off = np.floor_divide(coords.astype(np.int64) - (edge * mag) // 2, mag)          # (n,3) x,y,z at `mag`
    dev = torch.device('cuda', torch.cuda.current_device())
    _, inv = np.unique(np.floor_divide(off, _REGION_VOX), axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    for k in range(int(inv.max()) + 1):
        ix = np.nonzero(inv == k)[0]
        lo = off[ix].min(axis=0)
        hi = off[ix].max(axis=0) + edge
        vol = kd.load_raw(size=(hi - lo) * mag, offset=lo * mag, mag=mag)              # (z,y,x) uint8, zeros outside
        vol_dev = torch.from_numpy(np.ascontiguousarray(vol)).to(dev)
        org = torch.from_numpy(np.ascontiguousarray((off[ix] - lo)[:, ::-1]).astype(np.int32)).to(dev)
        res = box_majority_device(vol_dev, org, edge[::-1], thresh_proba, thresh_majority)
        preds[ix] = res.cpu().numpy()
    return preds
