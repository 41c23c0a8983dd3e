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
    assert par.values_of_all_ranks(10.0 * (rank + 1)) == [10.0 * (k + 1) for k in range(world)]      # (what bench.py reports per rank)
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
    # (the last combination deals the chunks from a cost-sorted list: cost = voxels of the chunk inside the volume, so the ragged
    # edge chunks come last -- the result must not depend on the deal order)
    def cost(valid_box):
        return int(np.prod(np.subtract(valid_box[1], valid_box[0])))
    for pipelined, root_computes, chunk_cost in ((True, True, None), (False, True, None), (True, False, None), (True, True, cost)):
        trace = []
        out = par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=2, pipelined=pipelined,
                                             root_computes=root_computes, trace=trace, chunk_cost=chunk_cost)
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


@pytest.mark.parametrize('arch,geometry', [('myelin', 'reference'), ('myelin', 'tile128'), ('mivcsj', 'reference'), ('mivcsj', 'tile128')])
def test_cost_sorted_rounds_reach_7_5x_on_8_ranks_by_geometry(arch, geometry):
    """BASELINE configs[3] (myelin) and configs[4] (mivcsj) on the 2048 x 2048 x 512 volume: lock-step rounds dealt in z-major order
    mix 12-tile interior chunks with 2-tile corner chunks of the overhanging chunk grid (6.1-6.4x of 8 by window voxels alone in the
    reference geometry); dealt from the cost-sorted list every round holds chunks of one cost class.  Model: a round takes as long as
    its most expensive chunk; cost = voxels of all windows the chunk's predicted tiles run on (syconn_amd.tiling, the arithmetic
    Predictor._tiled runs).  No communication in the model (DESIGN.md section 6 has that budget)."""
    from syconn_amd import parallel as par
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.tiling import ChunkCostModel, PlanClipper
    vol_shape = (512, 2048, 2048)
    if geometry == 'reference':      # prediction.py:672-677 (x,y,z) -> (z,y,x)
        chunk, halo, tile, inc = (236, 481, 482), (20, 31, 30), (138, 181, 271), False
    else:
        chunk, halo, tile, inc = (224, 192, 192), (8, 16, 16), (112, 96, 96), True
    cm = ChunkCostModel(PlanClipper.for_model(random_state_dict(arch, seed=0)), tile, halo, halo, inc)
    ids, boxes = par.chunk_grid(vol_shape, chunk, halo)
    costs = [cm.chunk_cost(chunk, b) for b in boxes]
    assert len(ids) == (75 if geometry == 'reference' else 363) and min(costs) > 0
    order = par.cost_sorted(list(range(len(ids))), costs)
    assert sorted(order) == list(range(len(ids)))
    sorted_costs = [costs[i] for i in order]
    assert all(a >= b for a, b in zip(sorted_costs, sorted_costs[1:]))
    zmajor, dealt = par.round_schedule_speedup(costs, 8), par.round_schedule_speedup(sorted_costs, 8)
    assert dealt >= 7.5 and dealt >= zmajor - 1e-9, (zmajor, dealt)
    if geometry == 'reference':
        assert zmajor < 6.5          # what the old order could reach at best
    for world in (2, 4):
        assert par.round_schedule_speedup(sorted_costs, world) >= 0.97 * world
    # ownership is still chunkify over the dealt list: rank r predicts dealt[r::8]
    assert [len(par.shard_units(order, r, 8)) for r in range(8)] == [len(order[r::8]) for r in range(8)]


def test_tile_plan_skips_and_clips_like_the_predictor_contract():
    """plan_tile_windows: tiles entirely beyond the valid box are skipped, the others keep full windows without clipping and never
    larger ones with it; GroupNorm plans keep full windows."""
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.tiling import PlanClipper, convolved_voxels, plan_tile_windows, tile_grid
    cl = PlanClipper.for_model(random_state_dict('myelin', seed=0))
    spatial = np.array([276, 543, 542])
    tile, ol, nt = tile_grid(spatial, (138, 181, 271), (20, 31, 30))
    assert tuple(nt) == (2, 3, 2)
    full, z0 = plan_tile_windows(spatial, tile, ol, nt, None, False, None, False)
    assert not z0 and convolved_voxels(full) == 12 * 178 * 243 * 331
    vb = ((20, 31, 30), (60, 155, 150))                       # a corner chunk of the overhanging grid: 40 x 124 x 120 voxels inside
    skipped, z1 = plan_tile_windows(spatial, tile, ol, nt, vb, False, None, False)
    assert z1 and sum(len(v) for v in skipped.values()) == 1 and convolved_voxels(skipped) == 178 * 243 * 331
    clipped, _ = plan_tile_windows(spatial, tile, ol, nt, vb, True, cl, False)
    assert convolved_voxels(clipped) < convolved_voxels(skipped)
    (win, roi), = clipped.keys()
    assert all(w % 8 == 0 or w == f for w, f in zip(win, (178, 243, 331)))
    gn = PlanClipper.for_model(random_state_dict('mivcsj', seed=0))
    assert gn.has_groupnorm and gn(20, 60, 178, 0) == (0, 178)
