"""Single-node multi-GPU execution of the dense prediction path: one process per GPU, ``torch.distributed``
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU unit tests of the sharding logic).

The path shards naturally: chunks / tiles are independent given read-only input with halo, so
there is NO collective inside the compute.  The reference's multi-GPU mode is share-nothing processes on a shared
file system (prediction.py:708-719); the MI355X-native variant keeps the same static round-robin ownership
(``chunkify``, basics.py:545-561) and adds exactly the two root-centric exchanges the north star names:

* ``broadcast_weights``: rank 0 -> all, the flat float32 parameter vector (16-150 MB), once per model;
* ``gather_to_root``: every rank's uint8 results -> rank 0 (1 byte per voxel and target; widened to uint64 only by
  the writer), issued per step and overlappable with the next step's compute.
"""
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from ._lib import host_box_copy, host_zero
from ._lib import load as _lib_load
from .handler.basics import chunkify

# strided host box copies issued by predict_volume_distributed since import (tests: the device path must not add any)
HOST_BOX_COPIES = 0

# host threads of the strided box copies (pack / stitch on rank 0, HOST path only -- CPU tensors, the gloo unit tests): measured on the 128-thread GPU box for the config-4 volume in
# 128^3-tile chunks, pack + stitch sustain 4.4 Gvox/s with 16 threads and 6.8 Gvox/s with 64 (tools/host_pack_rate.py)
HOST_THREADS = min(64, max(4, (os.cpu_count() or 8) // 2))

# rows per upload piece of the device volume path (predict_volume_distributed): pieces = row block x the z-planes a round needs next
UPLOAD_ROWS = 128


def _single_rank_group() -> bool:
    """``SD_DIST_SINGLE_RANK_GROUP=1``: create a process group even for ONE rank and send every payload through it.  A one-rank
    job needs no collective; the switch exists because a one-GPU box is the only hardware the test suite gets, RCCL refuses
    two ranks on one device, and a group of one is then the only way to run the "nccl" branches (device tensors handed to
    ``dist.scatter`` / ``dist.gather`` / ``dist.broadcast``, asynchronous work handles, the communication stream) on RCCL
    itself rather than on the host-staged gloo stand-in."""
    return os.environ.get('SD_DIST_SINGLE_RANK_GROUP') == '1'


def collectives_active() -> bool:
    """True when payloads travel through the process group: more than one rank, or a one-rank group made on purpose."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single_rank_group())


_collectives = collectives_active


def init_distributed(backend: Optional[str] = None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group (but see `_single_rank_group`)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or _single_rank_group()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_units(units: Sequence, rank: Optional[int] = None, world: Optional[int] = None) -> list:
    """Units (chunk ids / tile ids) owned by `rank`: ``chunkify(units, world)[rank]`` -- identical to the
    reference's chunk -> worker assignment (prediction.py:708-709).  Ranks beyond ``len(units)`` own nothing."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    parts = chunkify(list(units), world)
    return list(parts[rank]) if rank < len(parts) else []


def _staged(t: torch.Tensor) -> bool:
    """True when a payload collective on `t` has to be staged through host memory: device tensors on the gloo backend (gloo
    moves host memory only).  That is the functional stand-in for RCCL where RCCL cannot run -- the CPU unit tests, and
    ``SD_BENCH_ONE_GPU_DEBUG`` runs in which all ranks share ONE GPU (RCCL refuses two ranks on one device)."""
    return t.is_cuda and dist.get_backend() == 'gloo'


class _StagedWork:
    """Work handle of a host-staged collective: the transfer is complete in host order when the call returns; what is left is
    the device copy of the received payload on the issuing stream, and ``wait()`` orders the CURRENT stream behind it -- the
    stream semantics of an RCCL work handle."""

    def __init__(self, event):
        self._ev = event

    def wait(self):
        if self._ev is not None:
            torch.cuda.current_stream().wait_event(self._ev)
        return True


def _state(model) -> dict:
    return model if isinstance(model, dict) else model.state_dict()


def flatten_state(model) -> torch.Tensor:
    """All parameters and buffers that define the network (an ``nn.Module`` or a plain ``state_dict``) as one float32
    vector (deterministic key order)."""
    return torch.cat([v.detach().reshape(-1).to(torch.float32) for k, v in _state(model).items()
                      if not k.endswith('num_batches_tracked')])


def unflatten_state(model, flat: torch.Tensor) -> None:
    sd = _state(model)
    off = 0
    new = {}
    for k, v in sd.items():
        if k.endswith('num_batches_tracked'):
            new[k] = v
            continue
        n = v.numel()
        new[k] = flat[off:off + n].reshape(v.shape).to(v.dtype).cpu()
        off += n
    if off != flat.numel():
        raise ValueError('flat parameter vector does not match the model')
    if isinstance(model, dict):
        model.update(new)
    else:
        model.load_state_dict(new)


def broadcast_weights(model, src: int = 0, device: Optional[torch.device] = None) -> None:
    """Make every rank's `model` (``nn.Module`` or ``state_dict``) equal to rank `src`'s (Coll-1 of SURVEY.md 2.2)."""
    if not _collectives():
        return
    flat = flatten_state(model)
    if device is not None:
        flat = flat.to(device)
    dist.broadcast(flat, src=src)
    unflatten_state(model, flat.cpu())


def gather_to_root(local: torch.Tensor, dst: int = 0, async_op: bool = False, out: Optional[torch.Tensor] = None):
    """Gather equally-shaped uint8 result tensors on `dst` (Coll-3): root-centric, every rank sends its payload once.

    Returns ``(buffers, work)``: `buffers` is the list of per-rank tensors on `dst` (None elsewhere), `work` the async
    handle (None when synchronous or world size 1).  `out` optionally provides the (world, *local.shape) receive
    buffer so that steady-state steps allocate nothing.  A failing collective propagates: there is no fallback to
    another collective (a rank that switched alone would mismatch the others and hang the job)."""
    rank, world = world_info()
    if not _collectives():
        return [local], None
    bufs = None
    if rank == dst:
        if out is None:
            out = torch.empty((world, *local.shape), dtype=local.dtype, device=local.device)
        bufs = list(out.unbind(0))
    if _staged(local):
        host = local.contiguous().cpu()                          # (ordered behind the current stream's work)
        hbufs = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
        dist.gather(host, gather_list=hbufs, dst=dst)
        if rank == dst:
            for b, h in zip(bufs, hbufs):
                b.copy_(h)
        ev = torch.cuda.current_stream().record_event()
        return bufs, (_StagedWork(ev) if async_op else None)
    work = dist.gather(local.contiguous(), gather_list=bufs, dst=dst, async_op=async_op)
    return bufs, (work if async_op else None)


def scatter_from_root(payloads: Optional[Sequence[torch.Tensor]], like: torch.Tensor, src: int = 0, async_op: bool = False,
                      out: Optional[torch.Tensor] = None):
    """Coll-2: rank `src` hands payloads[r] (all shaped like `like`) to rank r; returns this rank's payload (and the
    work handle when `async_op`)."""
    rank, world = world_info()
    if not _collectives():
        return (payloads[0], None) if async_op else payloads[0]
    if out is None:
        out = torch.empty_like(like)
    if _staged(out):
        hout = torch.empty(out.shape, dtype=out.dtype)
        dist.scatter(hout, scatter_list=[p.contiguous().cpu() for p in payloads] if rank == src else None, src=src)
        out.copy_(hout)
        ev = torch.cuda.current_stream().record_event()
        return (out, _StagedWork(ev)) if async_op else out
    work = dist.scatter(out, scatter_list=[p.contiguous() for p in payloads] if rank == src else None, src=src,
                        async_op=async_op)
    return (out, work) if async_op else out


# pool of page-locked staging buffers (expensive to create: kept across calls).  A call TAKES its buffers out of the pool and
# gives them back when it is done, so two concurrent calls never share one; at most _PINNED_CAP idle buffers are kept.
_PINNED_POOL: dict = {}
_PINNED_CAP = 16
_PINNED_LOCK = __import__('threading').Lock()


def _pinned_take(shape, pin: bool) -> torch.Tensor:
    key = (tuple(int(v) for v in shape), bool(pin))
    with _PINNED_LOCK:
        free = _PINNED_POOL.get(key)
        if free:
            return free.pop()
    t = torch.empty(key[0], dtype=torch.uint8)
    return t.pin_memory() if pin else t


def _pinned_give(t: torch.Tensor, pin: bool) -> None:
    key = (tuple(t.shape), bool(pin))
    with _PINNED_LOCK:
        if sum(len(v) for v in _PINNED_POOL.values()) < _PINNED_CAP:
            _PINNED_POOL.setdefault(key, []).append(t)


def predict_volume_distributed(volume_u8: Optional[torch.Tensor], vol_shape: Sequence[int], chunk_shape: Sequence[int],
                               halo: Sequence[int], predict_fn, n_out: int, device=None,
                               pipelined: bool = True, root_computes: bool = True,
                               trace: Optional[list] = None, out: Optional[torch.Tensor] = None,
                               chunk_cost=None) -> Optional[torch.Tensor]:
    """Chunk-parallel dense prediction of one (z,y,x) uint8 volume over all ranks of the process group: the RCCL
    variant of the reference's "one worker per GPU, chunk ids dealt round-robin" (prediction.py:708-719), with the
    file system replaced by collectives.

    Rank 0 holds `volume_u8` in HOST memory (other ranks pass None) and receives the (n_out, *vol_shape) uint8 result in host
    memory as well (None elsewhere; `out`: a caller-owned -- ideally page-locked -- tensor to fill instead of a fresh pinned one).
    On a ROCm device rank 0's CPU does no per-chunk work at all: the volume is uploaded ONCE into rank 0's HBM, in pieces of
    `UPLOAD_ROWS` rows x the z-planes a round needs next (2D copies, each just before the first round that reads it), chunk + halo boxes are cut there by
    `sd_tile_gather` straight into the scatter staging buffer (zeros outside the volume), gathered results are placed by
    `sd_tile_scatter` into a device-resident result volume, and every (z-row, y-row) strip of chunks is downloaded with one 2D copy
    per output channel as soon as its last chunk has arrived -- 2 x 2 GiB over rank 0's PCIe link per 2048 x 2048 x 512 volume,
    under the kernels (rounds 1-4 packed and stitched every chunk with host threads: 6.3-6.9 Gvox/s on the GPU box, below what
    8 GPUs predict).  CPU tensors (the gloo unit tests of the sharding logic) keep the host pack / stitch.
    Chunks of `chunk_shape` are enumerated z-major and dealt round-robin over the WORKER ranks (== ``chunkify``): all
    ranks, or ranks 1 .. world-1 with ``root_computes=False`` (rank 0 then only packs, uploads, downloads and stitches --
    for volumes where its host threads and PCIe link are the bottleneck).  Per round rank 0 cuts one chunk per worker
    incl. `halo` (zeros outside the volume, like ``kd.load_raw``) into pinned staging buffers, uploads and scatters
    them; every worker runs ``predict_fn(chunk_with_halo_u8) -> uint8 (n_out, *chunk_shape)`` (halo already cropped) and
    rank 0 gathers the results, downloads them and stitches them into the output.

    With `pipelined` the stages overlap (two buffer sets; copy streams and a COMMUNICATION stream beside the compute
    stream): the scatter of round r+1 is issued BEFORE the prediction of round r is launched and both collectives are
    issued from the communication stream -- an RCCL collective is ordered behind the work of the stream that is current
    when it is issued, so issued from the compute stream it could not start before the kernels queued there had
    finished.  While the GPUs predict round r, xGMI carries round r+1 in and round r-1 out, and rank 0's host packs
    round r+2 and stitches round r-1 (the host part needs an asynchronous `predict_fn`).  Without `pipelined` every
    round runs scatter -> predict -> gather -> stitch strictly in sequence (A/B and debugging).
    A `predict_fn` that takes a keyword `valid_box` receives ((z0, y0, x0), (z1, y1, x1)), the part of the chunk PROPER that
    lies inside the volume in chunk + halo coordinates (the chunk grid overhangs the volume; ``Predictor`` skips model tiles
    whose result lies entirely beyond it).
    `chunk_cost(valid_box) -> number` (e.g. ``Predictor.chunk_cost_model(...).chunk_cost`` bound to the chunk shape): the chunk list is
    then dealt in order of DESCENDING cost (stable: equal costs stay z-major), so that the chunks of one lock-step round cost the
    same and the round barrier waits for nobody.  The chunk grid of the reference overhangs the dataset: in z-major order a round
    mixes 12-tile interior chunks with 2-tile corner chunks and 8 ranks top out at 6.1-6.4x by geometry alone; cost-sorted rounds
    reach 7.8-7.9x (`round_schedule_speedup`, tests/test_distributed_cpu.py).  Ownership stays ``chunkify`` over that list.
    `trace`: optional list that receives ('scatter' | 'predict' | 'gather' | 'stitch', round) in ISSUE order (tests)."""
    import itertools
    import numpy as np
    rank, world = world_info()
    coll = _collectives()          # payloads go through the process group (world > 1, or a one-rank group made on purpose)
    vs, cs, ol = (np.asarray(v, dtype=np.int64) for v in (vol_shape, chunk_shape, halo))
    grid = [int(-(-vs[i] // cs[i])) for i in range(3)]
    ids = list(itertools.product(*[range(g) for g in grid]))
    workers = list(range(world)) if (root_computes or world == 1) else list(range(1, world))
    nw = len(workers)
    slot_of = {w: k for k, w in enumerate(workers)}              # rank -> position of its chunk in a round
    import inspect
    try:
        wants_box = 'valid_box' in inspect.signature(predict_fn).parameters
    except (TypeError, ValueError):
        wants_box = False

    def valid_box_of(cid):
        lo = np.asarray(cid, dtype=np.int64) * cs - ol            # origin of the chunk + halo box in the volume
        a = np.maximum(ol, -lo)
        b = np.minimum(ol + cs, vs - lo)
        return tuple(int(v) for v in a), tuple(int(v) for v in b)

    if chunk_cost is not None:
        ids = cost_sorted(ids, [chunk_cost(valid_box_of(c)) for c in ids])       # (every rank computes the same list from geometry)
    rounds = [ids[r0:r0 + nw] for r0 in range(0, len(ids), nw)]
    nr = len(rounds)

    in_shape = tuple(int(v) for v in cs + 2 * ol)
    out_shape = (n_out, *[int(c) for c in cs])
    cuda = device is not None and torch.device(device).type == 'cuda'
    root = rank == 0
    my_slot = slot_of.get(rank, -1)

    def note(what, r):
        if trace is not None:
            trace.append((what, r))

    in_buf = [torch.empty(in_shape, dtype=torch.uint8, device=device) for _ in range(2)]
    res_buf = [torch.empty(out_shape, dtype=torch.uint8, device=device) for _ in range(2)]
    vol = None
    taken = []
    full_shape = (n_out, *[int(v) for v in vs])
    if root and out is not None and (tuple(out.shape) != full_shape or out.dtype != torch.uint8 or out.is_cuda or not out.is_contiguous()):
        raise ValueError(f'out: need a contiguous uint8 host tensor of shape {full_shape}')
    fits = True
    if root and cuda:
        # volume + result live in rank 0's HBM for the whole call: (1 + n_out) bytes per voxel beside the model's workspace.  Refuse
        # before the first payload collective what cannot fit (RuntimeError = the reference's out-of-memory convention), on ALL ranks:
        # rank 0 failing alone in the middle of a round would leave the others waiting in a collective
        need_bytes = (0 if volume_u8.is_cuda else int(np.prod(vs))) + n_out * int(np.prod(vs))
        free_bytes = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        fits = need_bytes <= 0.9 * free_bytes
    if coll and cuda:
        flag = torch.tensor([1 if fits else 0], dtype=torch.int32, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.broadcast(flag, src=0)
        fits = bool(flag.item())
    if not fits:
        raise RuntimeError(f"predict_volume_distributed: volume + {n_out} result channel(s) = {(1 + n_out) * int(np.prod(vs)) / 2**30:.1f} GiB do not fit "
                           f"rank 0's free HBM: predict the volume in z-slabs")
    if root and cuda:
        from .engine import tile_gather, tile_scatter
        vol = volume_u8.contiguous()
        vol_dev = vol if vol.is_cuda else torch.empty(tuple(int(v) for v in vs), dtype=torch.uint8, device=device)
        out_dev = torch.empty(full_shape, dtype=torch.uint8, device=device)
        if out is None:
            out = torch.empty(full_shape, dtype=torch.uint8, pin_memory=True)
        stage = [torch.empty((world, *in_shape), dtype=torch.uint8, device=device) for _ in range(2)] if coll else None
        recv = [torch.empty((world, *out_shape), dtype=torch.uint8, device=device) for _ in range(2)] if coll else None
        # the volume goes up in pieces of UPLOAD_ROWS rows x the z-planes a round needs next (one 2D copy each, in the order the rounds
        # need them): up_z[b] = first z-plane of row block b that is not in HBM yet.  (Whole z-slabs, rounds 5: the first round waited for
        # 276 planes x all rows = 1.16 GB of the 2048 x 2048 x 512 volume, 23 ms of the ~185 ms a volume takes on 8 ranks; its 8 chunks
        # read 1024 of the 2048 rows: 0.54 GB.)
        n_row_blocks = -(-int(vs[1]) // UPLOAD_ROWS)
        up_z = [0 if not vol.is_cuda else int(vs[0])] * n_row_blocks
        # results leave HBM strip by strip: a strip = the chunks (zi, yi, all x) = z-planes x a contiguous run of rows, one 2D copy per
        # output channel as soon as its last chunk has arrived (cost-sorted rounds finish whole z-rows only at the very end: row-wise
        # downloads would leave the entire result for a tail after the last kernel)
        strips_left = {(zi, yi): grid[2] for zi in range(grid[0]) for yi in range(grid[1])}
        lib = _lib_load()
    elif root:
        vol = volume_u8.contiguous()
        if out is None:
            out = torch.empty(full_shape, dtype=torch.uint8)
        pin_in = [_pinned_take((world, *in_shape), cuda) for _ in range(2)]       # packed payloads of a round, by RANK
        pin_out = [_pinned_take((world, *out_shape), cuda) for _ in range(2)]     # gathered results of a round, by RANK
        taken = pin_in + pin_out
        stage = recv = None
    else:
        recv = None
    if cuda:
        cur = torch.cuda.current_stream(device)
        s_in, s_out, s_comm = (torch.cuda.Stream(device=device) for _ in range(3))
        ev_h2d = [torch.cuda.Event() for _ in range(2)]       # upload of a round has left pin_in / reached stage
        ev_scat = [torch.cuda.Event() for _ in range(2)]      # scatter of a round has consumed stage and filled in_buf
        ev_pred = [torch.cuda.Event() for _ in range(2)]      # prediction of a round has consumed in_buf and filled res_buf
        ev_gath = [torch.cuda.Event() for _ in range(2)]      # gather of a round has consumed res_buf and filled recv
        ev_d2h = [torch.cuda.Event() for _ in range(2)]       # download of a round has left recv / reached pin_out
    packed = [-1, -1]          # round whose payloads sit in pin_in[s] (uploaded)

    def comm_ctx():
        return torch.cuda.stream(s_comm) if cuda else __import__('contextlib').nullcontext()

    def pack_upload(r):
        """host: cut the chunk + halo boxes of round r out of the volume into pin_in[r % 2] (zeros outside the volume), then
        upload them (copy-in stream)"""
        if not root or r >= nr or packed[r & 1] == r:
            return
        s = r & 1
        if cuda:
            # device path: no host work.  On the copy-in stream: the z-slabs of the volume this round reads that are not in HBM yet
            # (uploads stay in z order, one contiguous copy each), then one sd_tile_gather per chunk of the round
            with torch.cuda.stream(s_in):
                if r >= 2:       # round r-2's scatter has consumed stage[s] (world > 1) / its kernels have consumed in_buf[s]
                    s_in.wait_event(ev_scat[s] if coll else ev_pred[s])
                plane, row = int(vs[1]) * int(vs[2]), int(vs[2])
                for c in rounds[r]:
                    z_need = min(int(vs[0]), (int(c[0]) + 1) * int(cs[0]) + int(ol[0]))
                    y_lo = max(0, int(c[1]) * int(cs[1]) - int(ol[1]))
                    y_hi = min(int(vs[1]), (int(c[1]) + 1) * int(cs[1]) + int(ol[1]))
                    for b in range(y_lo // UPLOAD_ROWS, (y_hi - 1) // UPLOAD_ROWS + 1 if y_hi > y_lo else 0):
                        if up_z[b] >= z_need:
                            continue
                        y0, y1 = b * UPLOAD_ROWS, min(int(vs[1]), (b + 1) * UPLOAD_ROWS)
                        off = up_z[b] * plane + y0 * row
                        rc = lib.sd_memcpy2d_async(vol_dev.data_ptr() + off, plane, vol.data_ptr() + off, plane, (y1 - y0) * row,
                                                   z_need - up_z[b], 0, s_in.cuda_stream)
                        if rc != 0:
                            raise RuntimeError('sd_memcpy2d_async failed: ' + lib.sd_last_error().decode(errors='replace'))
                        up_z[b] = z_need
                for k, c in enumerate(rounds[r]):
                    lo = np.asarray(c, dtype=np.int64) * cs - ol
                    tile_gather(vol_dev, lo, in_shape, stage[s][workers[k]] if coll else in_buf[s])
                ev_h2d[s].record(s_in)
            packed[s] = r
            return
        global HOST_BOX_COPIES
        for k, w in enumerate(workers):
            dst = pin_in[s][w]
            if k >= len(rounds[r]):
                host_zero(dst, HOST_THREADS)
                continue
            lo = np.asarray(rounds[r][k], dtype=np.int64) * cs - ol
            hi = lo + np.asarray(in_shape, dtype=np.int64)
            a, b = np.maximum(lo, 0), np.minimum(hi, vs)
            if np.any(a > lo) or np.any(b < hi):
                host_zero(dst, HOST_THREADS)
            if np.all(b > a):       # strided box copy on host threads (C helper; numpy / torch slicing runs on one core)
                HOST_BOX_COPIES += 1
                host_box_copy(dst[a[0] - lo[0]:b[0] - lo[0], a[1] - lo[1]:b[1] - lo[1], a[2] - lo[2]:b[2] - lo[2]],
                              vol[a[0]:b[0], a[1]:b[1], a[2]:b[2]], HOST_THREADS)
        if not coll:
            in_buf[s].copy_(pin_in[s][0])
        packed[s] = r

    def issue_scatter(r):
        """scatter round r into in_buf[r % 2], issued from the communication stream; returns the work handle (None for
        world size 1)"""
        s = r & 1
        pack_upload(r)
        note('scatter', r)
        if not coll:
            if cuda:
                cur.wait_event(ev_h2d[s])
            return None
        with comm_ctx():
            if cuda:
                if root:
                    s_comm.wait_event(ev_h2d[s])
                if r >= 2:
                    s_comm.wait_event(ev_pred[s])                # the kernels of round r-2 have consumed in_buf[s]
            _, work = scatter_from_root(list((stage[s] if cuda else pin_in[s]).unbind(0)) if root else None, in_buf[s], src=0,
                                        async_op=True, out=in_buf[s])
        return work

    def issue_gather(r):
        """gather round r's results to rank 0 (communication stream) and download them into pin_out[r % 2] (copy-out stream)"""
        s = r & 1
        note('gather', r)
        if coll:
            with comm_ctx():
                if cuda:
                    s_comm.wait_event(ev_pred[s])
                    if root and r >= 2:
                        s_comm.wait_event(ev_d2h[s])             # the download of round r-2 has left recv[s]
                _, work = gather_to_root(res_buf[s], dst=0, async_op=True, out=(recv[s] if cuda else pin_out[s]) if root else None)
                work.wait()                                      # NCCL: orders s_comm behind the collective; gloo: host wait
                if cuda:
                    ev_gath[s].record(s_comm)
        if root and cuda:
            # device path: place the round's results in the device-resident result volume (copy-out stream), and download every
            # row of chunks that is complete now -- one contiguous slab per output channel
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_gath[s] if coll else ev_pred[s])
                for k, c in enumerate(rounds[r]):
                    lo = np.asarray(c, dtype=np.int64) * cs
                    tile_scatter(recv[s][workers[k]] if coll else res_buf[s], (0, 0, 0), np.minimum(cs, vs - lo), out_dev, lo)
                    key = (int(c[0]), int(c[1]))
                    strips_left[key] -= 1
                    if strips_left[key] == 0:
                        z0, z1 = key[0] * int(cs[0]), min(int(vs[0]), (key[0] + 1) * int(cs[0]))
                        y0, y1 = key[1] * int(cs[1]), min(int(vs[1]), (key[1] + 1) * int(cs[1]))
                        plane, row = int(vs[1]) * int(vs[2]), int(vs[2])
                        for ch in range(n_out):
                            off = (ch * int(vs[0]) + z0) * plane + y0 * row
                            rc = lib.sd_memcpy2d_async(out.data_ptr() + off, plane, out_dev.data_ptr() + off, plane, (y1 - y0) * row, z1 - z0,
                                                       1, s_out.cuda_stream)
                            if rc != 0:
                                raise RuntimeError('sd_memcpy2d_async failed: ' + lib.sd_last_error().decode(errors='replace'))
                ev_d2h[s].record(s_out)                          # (recv[s] / res_buf[s] are free again)
        elif root and not coll:
            pin_out[s][0].copy_(res_buf[s])

    def stitch(r):
        """host: results of round r (pin_out[r % 2]) -> output volume"""
        note('stitch', r)
        if not root or cuda:                                     # (device path: done by the copy-out stream, see issue_gather)
            return
        s = r & 1
        global HOST_BOX_COPIES
        HOST_BOX_COPIES += len(rounds[r]) * n_out
        for k in range(len(rounds[r])):
            w = workers[k]
            lo = np.asarray(rounds[r][k], dtype=np.int64) * cs
            n = np.minimum(cs, vs - lo)
            for c in range(n_out):
                host_box_copy(out[c, lo[0]:lo[0] + n[0], lo[1]:lo[1] + n[1], lo[2]:lo[2] + n[2]],
                              pin_out[s][w][c, :n[0], :n[1], :n[2]], HOST_THREADS)

    try:
        pend = issue_scatter(0) if nr else None
        for r in range(nr):
            s = r & 1
            if pend is not None:
                pend.wait()                                      # NCCL: the compute stream waits for scatter(r); gloo: host wait
                if cuda:
                    ev_scat[s].record(cur)
            # the NEXT round's scatter goes out before this round's kernels are launched (its payloads were packed and
            # uploaded while the previous round was predicted): xGMI transfer of r+1 under the kernels of r
            pend = issue_scatter(r + 1) if (pipelined and r + 1 < nr) else None
            note('predict', r)
            if cuda and r >= 2:          # res_buf[s] is free again: round r-2's gather (download for world size 1) has read it
                cur.wait_event(ev_gath[s] if coll else ev_d2h[s])
            if my_slot >= 0 and my_slot < len(rounds[r]):
                if wants_box:
                    res_buf[s].copy_(predict_fn(in_buf[s], valid_box=valid_box_of(rounds[r][my_slot])))
                else:
                    res_buf[s].copy_(predict_fn(in_buf[s]))
            else:
                res_buf[s].zero_()
            if cuda:
                ev_pred[s].record(cur)
            issue_gather(r)
            if not pipelined:
                stitch(r)
                pend = issue_scatter(r + 1) if r + 1 < nr else None
            else:
                pack_upload(r + 2)                               # host packs round r+2 while the GPUs compute round r
                if r >= 1:
                    stitch(r - 1)
        if pipelined and nr:
            stitch(nr - 1)
        if cuda:
            cur.wait_stream(s_comm)                              # nothing of this call is still in flight when it returns
            cur.wait_stream(s_in)
            cur.wait_stream(s_out)
            if root:
                s_out.synchronize()                              # host: every slab of the result has arrived
    finally:
        for t in taken:
            _pinned_give(t, cuda)
    return out if root else None


def chunk_grid(vol_shape: Sequence[int], chunk_shape: Sequence[int], halo: Sequence[int]):
    """-> (ids, valid boxes): chunk ids (iz, iy, ix) in z-major order and, per chunk, the part of the chunk proper that lies inside the
    volume in chunk + halo coordinates -- what ``predict_volume_distributed`` hands to ``predict_fn(valid_box=...)`` / ``chunk_cost``."""
    import itertools
    import numpy as np
    vs, cs, ol = (np.asarray(v, dtype=np.int64) for v in (vol_shape, chunk_shape, halo))
    ids = list(itertools.product(*[range(int(-(-vs[i] // cs[i]))) for i in range(3)]))
    boxes = []
    for cid in ids:
        lo = np.asarray(cid, dtype=np.int64) * cs - ol
        boxes.append((tuple(int(v) for v in np.maximum(ol, -lo)), tuple(int(v) for v in np.minimum(ol + cs, vs - lo))))
    return ids, boxes


def cost_sorted(ids: Sequence, costs: Sequence[float]) -> list:
    """`ids` in order of descending cost; equal costs keep their order (z-major ids: the z-slab uploads stay in order inside a class)."""
    order = sorted(range(len(ids)), key=lambda i: -float(costs[i]))      # sorted() is stable
    return [ids[i] for i in order]


def round_schedule_speedup(costs_in_deal_order: Sequence[float], world: int) -> float:
    """Modelled speed-up of lock-step rounds over one rank: every round takes as long as its most expensive chunk
    (sum of all costs / sum over rounds of the round's maximum) -- geometry only, no communication."""
    c = [float(v) for v in costs_in_deal_order]
    rounds = [c[i:i + world] for i in range(0, len(c), world)]
    return sum(c) / sum(max(r) for r in rounds) if c else 1.0


def barrier():
    if _collectives():
        dist.barrier()


def values_of_all_ranks(value: float, device: Optional[torch.device] = None) -> List[float]:
    """`value` of every rank, in rank order, on every rank (one small all-gather; [value] without a process group)."""
    if not _collectives():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if (device is not None and dist.get_backend() == 'nccl') else 'cpu')
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not _collectives():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
