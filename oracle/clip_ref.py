"""TEST INFRASTRUCTURE (oracle): Python restatement of the window clipping the product does in C
(`sd_plan_clip_window`, syconn_amd/csrc/sd_host.cpp).  Only tests import this.

What it restates is not a reference function: the reference predicts every tile on its full window (elektronn3 `tiled_apply`,
SURVEY.md row P3; chunk grid /root/reference/syconn/handler/prediction.py:679-683, halo crop :812).  The claim checked with it
is that the oracle U-Net's wanted outputs are the same on the clipped window (tests/test_host_logic.py), and that the C
implementation computes exactly these windows.
"""
import math

from syconn_amd import _lib as L


def clipped_extent(ops, need: int, full: int, axis: int, multiple: int = 1) -> int:
    """Smallest input extent E <= `full` along `axis` (0 = z, 1 = y, 2 = x) for which the network's outputs with index
    < `need` are what they are at extent `full`, given that the input is unchanged below E.  Used for model tiles that reach
    beyond the chunk / dataset (tiled_apply pads them with zeros and the crop throws the results away): everything an output
    voxel depends on lies inside a cone, and 'same' padding, ceil-mode pooling windows and the up-convolution crop at the far
    border only matter to voxels whose cone touches that border.  Backward pass: how many leading indices of every buffer the
    wanted outputs read (conv k: + k//2; pooling f: * f; transposed conv f: ceil(/ f)); forward pass: the extent every buffer
    has at input extent E; E is valid when no buffer is read beyond its extent, which keeps the far border of every layer
    outside every cone.  GroupNorm reads the whole tile: `full`."""
    ksz = (lambda d: (d.kz, d.ky, d.kx)[axis])
    if need >= full or any(d.kind == L.SD_OP_GROUPNORM for d in ops):
        return full
    reads = {}
    for d in reversed(ops):
        k = int(ksz(d))
        if d.kind == L.SD_OP_FINAL:
            n = need
        else:
            if int(d.dst) not in reads:
                continue
            n = reads[int(d.dst)]
        if d.kind == L.SD_OP_CONV:
            n += k // 2
        elif d.kind == L.SD_OP_POOL:
            n *= k
        elif d.kind == L.SD_OP_UPCONV:
            n = -(-n // k)
        for s in (int(d.src0), int(d.src1)) if d.kind == L.SD_OP_CONV else (int(d.src0),):
            if s >= 0:
                reads[s] = max(reads.get(s, 0), n)

    def fits(e: int) -> bool:
        ext = {0: e}
        for d in ops:
            k = int(ksz(d))
            if d.kind == L.SD_OP_FINAL:
                continue
            a = ext[int(d.src0)]
            if d.kind == L.SD_OP_CONV and d.src1 >= 0:
                a = min(a, ext[int(d.src1)])           # autocrop (row U4): the larger operand loses its far end
            elif d.kind == L.SD_OP_POOL:
                a = -(-a // k)                         # ceil_mode
            elif d.kind == L.SD_OP_UPCONV:
                a *= k
            ext[int(d.dst)] = a
        return all(ext[b] >= n for b, n in reads.items())

    e = -(-max(reads.get(0, need), need) // multiple) * multiple
    while e < full and not fits(e):
        e += multiple
    return min(e, full)


def clipped_window(ops, lo: int, hi: int, full: int, axis: int, multiple: int = 1):
    """(start, extent) of the part of an input window of `full` voxels along `axis` that the outputs lo <= index < hi depend
    on: `clipped_extent` for the far side plus the same argument for the near side.  The near border may only move by a multiple
    of the network's total pooling stride along the axis (pooling windows and up-convolution parities keep their places), and no
    wanted output may read any buffer below its new first index (conv k: - k//2; pooling f: * f; transposed conv f: floor(/ f))."""
    ksz = (lambda d: (d.kz, d.ky, d.kx)[axis])
    if any(d.kind == L.SD_OP_GROUPNORM for d in ops) or lo <= 0:
        return 0, clipped_extent(ops, hi, full, axis, multiple)
    scale, reads = {0: 1}, {}
    for d in ops:
        if d.kind != L.SD_OP_FINAL:
            k, sc = int(ksz(d)), scale[int(d.src0)]
            scale[int(d.dst)] = sc * k if d.kind == L.SD_OP_POOL else sc // k if d.kind == L.SD_OP_UPCONV else sc
    for d in reversed(ops):
        k = int(ksz(d))
        if d.kind == L.SD_OP_FINAL:
            n = lo
        elif int(d.dst) in reads:
            n = reads[int(d.dst)]
        else:
            continue
        if d.kind == L.SD_OP_CONV:
            n -= k // 2
        elif d.kind == L.SD_OP_POOL:
            n *= k
        elif d.kind == L.SD_OP_UPCONV:
            n //= k
        for s in (int(d.src0), int(d.src1)) if d.kind == L.SD_OP_CONV else (int(d.src0),):
            if s >= 0:
                reads[s] = min(reads.get(s, n), n)
    stride = max(scale.values())
    stride = stride * multiple // math.gcd(stride, multiple)
    start = max(0, min(n * scale[b] for b, n in reads.items()) // stride * stride)
    return start, clipped_extent(ops, hi - start, full - start, axis, multiple)
