"""CPU, world_size 2, gloo: the sharding + collective logic of the multi-GPU path (syconn_amd/parallel.py).
The HIP forward cannot run here, so the per-unit 'compute' is a stand-in; what is tested is exactly what differs
between N = 1 and N > 1: weight broadcast, round-robin ownership identical to the reference's chunkify, gather of
uint8 results on rank 0, barrier / max-over-ranks timing helpers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle.unet_ref import build_unet
    from syconn_amd import parallel as par
    r, w, _ = par.init_distributed('gloo')
    assert (r, w) == (rank, world)
    # Coll-1: ranks start from different weights, end with rank 0's
    model = build_unet('myelin', seed=rank, n_blocks=2, start_filts=4)
    par.broadcast_weights(model, src=0)
    ref = build_unet('myelin', seed=0, n_blocks=2, start_filts=4)
    same = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ref.state_dict().values()))
    # ownership: identical to the reference's chunkify(chunk_ids, ngpu_total)[rank]
    mine = par.shard_units(range(n_units))
    # stand-in compute: unit id -> constant uint8 block; pad to the max share so that gather shapes agree
    share = -(-n_units // world)
    local = torch.zeros((share, 4, 4, 4), dtype=torch.uint8)
    for k, u in enumerate(mine):
        local[k] = u + 1
    got, _ = par.gather_to_root(local, dst=0)
    got_async, work = par.gather_to_root(local, dst=0, async_op=True)
    if work is not None:
        work.wait()
    if rank == 0:
        assert all(torch.equal(a, b) for a, b in zip(got, got_async))
    t = par.max_over_ranks(float(rank + 1))
    par.barrier()
    if rank == 0:
        vol = np.zeros(n_units, np.int64)
        for rr, buf in enumerate(got):
            for k, u in enumerate(par.shard_units(range(n_units), rr, world)):
                vol[u] = int(buf[k, 0, 0, 0])
        q.put(('root', same, mine, vol.tolist(), t))
    else:
        q.put(('rank', same, mine, got is None, t))
    dist.destroy_process_group()


@pytest.mark.parametrize('n_units', [75, 5, 1])
def test_two_rank_shard_broadcast_gather(n_units):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    root = [r for r in res if r[0] == 'root'][0]
    other = [r for r in res if r[0] == 'rank'][0]
    assert root[1] and other[1], 'weights differ after broadcast'
    assert root[2] == list(range(n_units))[0::2] and other[2] == (list(range(n_units))[1::2] if n_units > 1 else [])
    assert root[3] == [u + 1 for u in range(n_units)], 'gathered results are not the union of all shards'
    assert other[3] is True and root[4] == 2.0 and other[4] == 2.0


def test_shard_units_matches_reference_partition():
    sys.path.insert(0, ROOT)
    from syconn_amd import parallel as par
    parts = [par.shard_units(range(75), r, 8) for r in range(8)]
    assert [len(p) for p in parts] == [10, 10, 10, 9, 9, 9, 9, 9]
    assert sorted(v for p in parts for v in p) == list(range(75))
    assert par.shard_units(range(3), 5, 8) == []


def _worker_volume(rank, world, port, q, single=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    if single:
        os.environ['SD_DIST_SINGLE_RANK_GROUP'] = '1'
    import torch.distributed as dist
    from syconn_amd import parallel as par
    par.init_distributed('gloo')
    assert par._collectives() and dist.get_world_size() == world
    vol_shape, chunk, halo = (20, 30, 26), (8, 16, 12), (2, 3, 1)
    vol = torch.from_numpy(np.random.default_rng(0).integers(0, 200, vol_shape, dtype=np.uint8)) if rank == 0 else None

    def predict_fn(ch):      # stand-in network: 3x3x3 box MAX with zero padding == needs the halo, exposes every offset bug
        x = ch[None, None].float()
        m = torch.nn.functional.max_pool3d(x, 3, stride=1, padding=1)[0, 0]
        core = m[halo[0]:-halo[0], halo[1]:-halo[1], halo[2]:-halo[2]]
        return torch.stack([core.to(torch.uint8), (255 - core).to(torch.uint8)])
    ok = True
    # overlapped scatter / predict / gather, the lock-step variant, and "rank 0 does not compute": same result
    for pipelined, root_computes in ((True, True), (False, True), (True, False)):
        trace = []
        out = par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=2, pipelined=pipelined,
                                             root_computes=root_computes, trace=trace)
        if rank == 0:
            ref = torch.nn.functional.max_pool3d(vol[None, None].float(), 3, stride=1, padding=1)[0, 0].to(torch.uint8)
            ok = ok and bool(torch.equal(out[0], ref) and torch.equal(out[1], 255 - ref))
        else:
            ok = ok and out is None
        # ISSUE ORDER (every rank; collectives must be issued in the same order everywhere): pipelined -> the scatter of
        # round r+1 goes out BEFORE the prediction of round r is launched and the gather of round r behind it; lock-step ->
        # scatter(r+1) only after gather(r) and stitch(r)
        pos = {e: i for i, e in enumerate(trace)}
        nr = 1 + max(r for _, r in trace)
        ok = ok and nr >= 3 and len(pos) == len(trace)
        for r in range(nr):
            ok = ok and pos[('scatter', r)] < pos[('predict', r)] < pos[('gather', r)] < pos[('stitch', r)]
            if r + 1 < nr:
                if pipelined:
                    ok = ok and pos[('scatter', r + 1)] < pos[('predict', r)]
                else:
                    ok = ok and pos[('stitch', r)] < pos[('scatter', r + 1)]
    if rank == 0:
        q.put(ok)
    dist.destroy_process_group()


def test_two_rank_chunk_scatter_predict_gather():
    """Coll-2 + Coll-3 end to end on 2 ranks: scatter chunk+halo payloads, per-rank prediction, gather + stitch."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_volume, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok, 'distributed chunk prediction does not reproduce the single-process result'


def test_group_of_one_sends_every_payload_through_the_process_group():
    """``SD_DIST_SINGLE_RANK_GROUP=1``: a process group of ONE rank through which every payload still travels (how the GPU suite
    runs the collectives on RCCL on a one-GPU box, tests/test_gpu_multirank.py); here on gloo: same volume, same issue order."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_volume, args=(0, 1, _free_port(), q, True))
    p.start()
    ok = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0 and ok


def test_single_process_volume_prediction_is_the_same_code_path():
    """world size 1 (no process group): the pipelined chunk loop must reproduce the whole-volume result as well."""
    from syconn_amd import parallel as par
    vol_shape, chunk, halo = (17, 20, 33), (8, 8, 16), (1, 2, 3)
    vol = torch.from_numpy(np.random.default_rng(1).integers(0, 255, vol_shape, dtype=np.uint8))

    def predict_fn(ch):
        m = torch.nn.functional.max_pool3d(ch[None, None].float(), 3, stride=1, padding=1)[0, 0]
        return m[halo[0]:-halo[0], halo[1]:-halo[1], halo[2]:-halo[2]].to(torch.uint8)[None]
    ref = torch.nn.functional.max_pool3d(vol[None, None].float(), 3, stride=1, padding=1)[0, 0].to(torch.uint8)
    for pipelined in (True, False):
        out = par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=1, pipelined=pipelined)
        assert torch.equal(out[0], ref)
