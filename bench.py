#!/usr/bin/env python3
"""Headline benchmark of the dense 3D-CNN prediction path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched through torch.distributed.run)

metric   : segmented Mvoxels/s (whole node) on synthetic 128^3 uint8 EM tiles  (BASELINE.json `metric`)
workload : BASELINE.json configs[1] -- `semseg_spine` 3D U-Net (myelin trunk, 5 classes),
           bf16 activations / fp32 accumulate, whole 128^3 tile per forward.
step     : one pass of the hot path over one batch of `--tiles` tiles per GPU, inputs resident in HBM as uint8:
           uint8 tile -> [normalise, U-Net, softmax, floor(255 p)] -> uint8 probabilities -> label rule
           (prediction.py:813-833) -> uint8 label volume; with N > 1 the label volumes are gathered on rank 0 (RCCL).
value    = tiles * 128^3 * N * K / (max over ranks of the wall time of K steps), in Mvox/s.  Weak scaling.

Extra objects on the JSON line: `roofline` for the dominant kernel (the 3x3x3 MFMA convolution; live HIP-event
timings recorded on the launch stream inside the timed region) and `cpu_baseline` (the torch-CPU oracle on ONE tile
on this box's host cores, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_MFMA_TFLOPS = 2500.0    # dense bf16/f16 MFMA


def synthetic_em_tiles(n, size, seed):
    """Structured synthetic EM: Gaussian-filtered (sigma 2) uniform noise rescaled to 0..255."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    out = np.empty((n, size, size, size), np.uint8)
    for i in range(n):
        v = gaussian_filter(rng.random((size, size, size), dtype=np.float32), 2.0)
        v = (v - v.min()) / (v.max() - v.min())
        out[i] = (v * 255).astype(np.uint8)
    return out


def layer_accounting(ops, L, D, H, W, act_bytes=2, out_bytes_per_class=1, batch=1):
    """Algorithmic FLOPs and bytes of every plan op for a (D,H,W) tile.  Bytes follow SURVEY.md section 8(d):
    activation read + write per fused conv(+norm+act) layer, a pooled tensor counts its write only (it belongs in
    the producing conv's epilogue), concat = two read pointers, weights ignored (L2/MALL resident)."""
    dims = {0: (D, H, W)}
    chans = {0: 1}
    rows = []
    fused_first = set()
    for io, o in enumerate(ops):
        if o.kind == L.SD_OP_CONV:
            d = dims[o.src1] if o.src1 >= 0 else dims[o.src0]
            vox = d[0] * d[1] * d[2]
            cin = o.cin0 + max(o.cin1, 0)
            taps = o.kz * 9
            flops = 2.0 * vox * cin * o.cout * taps
            inb = vox * (1 if o.src0 == 0 else cin * act_bytes)
            # kernel instantiation the library dispatches (mirrors launch_conv_knt in sd_kernels.hip)
            if o.src0 == 0:
                name = 'k_conv_first<%dx3x3>' % o.kz
            else:
                ntile = (o.cout + 31) // 32
                nt = 3 if ntile % 3 == 0 else (2 if ntile >= 2 else 1)
                nbk = (ntile + nt - 1) // nt
                waves = 8 if ((vox * batch) // 512) * nbk >= 512 else 4
                # LDS-resident weights (NSLOT=2) when the layer's weight groups fit beside a 2-slot halo ring
                bz, by = ((waves // 4) * 2, 8) if o.kz == 3 else (1, waves * 4)
                a_bytes = -(-((bz + o.kz - 1) * (by + 2) * 18 * 2) // 64) * 1024
                nstages = (-(-o.cin0 // 16) + (-(-o.cin1 // 16) if o.cin1 > 0 else 0)) * o.kz
                fused_final = io + 1 < len(ops) and ops[io + 1].kind == L.SD_OP_FINAL and nbk == 1
                lds = 2 * a_bytes + nstages * 9 * nt * 1024 + 512 + (nt * 4096 if fused_final else 0) + 1024
                resident = lds <= (96 if waves == 8 else 80) * 1024
                name = 'k_conv_mfma<%dx3x3,NT=%d,%d waves,%s>' % (o.kz, nt, waves, 'NSLOT=2' if resident else 'NSLOT=0')
                # first conv computed inside this conv (conv_can_fuse_first in sd_kernels.hip)
                prev = ops[io - 1] if io > 0 else None
                if (prev is not None and prev.kind == L.SD_OP_CONV and prev.src0 == 0 and prev.kz == 1 and prev.cout == 32
                        and o.kz == 1 and o.src0 == prev.dst and o.src1 < 0 and nt <= 2 and nstages == 2 and waves == 8
                        and lds + 36 * 20 * 4 <= 96 * 1024):
                    name = name[:-1] + ',FF>'
                    fused_first.add(io - 1)
            rows.append((name, flops, inb + vox * o.cout * act_bytes))
            dims[o.dst], chans[o.dst] = d, o.cout
        elif o.kind == L.SD_OP_POOL:
            d = dims[o.src0]
            do = ((d[0] + 1) // 2 if o.kz == 2 else d[0], (d[1] + 1) // 2, (d[2] + 1) // 2)
            rows.append(('pool', 0.0, do[0] * do[1] * do[2] * chans[o.src0] * act_bytes))
            dims[o.dst], chans[o.dst] = do, chans[o.src0]
        elif o.kind == L.SD_OP_UPCONV:
            d = dims[o.src0]
            vox = d[0] * d[1] * d[2]
            taps = o.kz * 4
            do = (d[0] * o.kz, d[1] * 2, d[2] * 2)
            rows.append(('upconv', 2.0 * vox * o.cin0 * o.cout * taps,
                         (vox * o.cin0 + vox * taps * o.cout) * act_bytes))
            dims[o.dst], chans[o.dst] = do, o.cout
        elif o.kind == L.SD_OP_GROUPNORM:
            d = dims[o.src1] if o.src1 >= 0 else dims[o.src0]
            vox = d[0] * d[1] * d[2]
            rows.append(('groupnorm', 0.0, 2.0 * vox * chans[o.src0] * act_bytes))
        elif o.kind == L.SD_OP_FINAL:
            d = dims[o.src0]
            vox = d[0] * d[1] * d[2]
            rows.append(('final', 2.0 * vox * o.cin0 * o.cout, vox * (o.cin0 * act_bytes + o.cout * out_bytes_per_class)))
    for i in fused_first:             # the first conv's work is done inside its consumer: no launch of its own
        rows[i] = ('k_conv_first(fused into next)',) + tuple(rows[i][1:])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--tiles', type=int, default=8, help='128^3 tiles per GPU per step')
    ap.add_argument('--tile', type=int, default=128)
    ap.add_argument('--arch', default='semseg_spine')
    ap.add_argument('--act', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--batch', type=int, default=0, help='tiles per sd_forward_batch launch set (0 = all tiles of a step)')
    ap.add_argument('--streams', type=int, default=1, help='HIP streams the batches of a step alternate over')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from syconn_amd import _lib as L
    from syconn_amd import parallel as par
    from syconn_amd.engine import DenseModel, StreamRing, postproc_labels
    from oracle.unet_ref import build_unet   # architecture definition + seeded random init (no trained weights exist)

    # SD_BENCH_ONE_GPU_DEBUG=1: exercise the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo) -- a
    # functional check of the sharding / gather / timing logic only, never a measurement
    one_gpu_debug = bool(os.environ.get('SD_BENCH_ONE_GPU_DEBUG'))
    rank, world, local_rank = par.init_distributed('gloo' if one_gpu_debug else None)
    if one_gpu_debug:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    # random-init weights: rank 0 is authoritative, everybody else receives them over RCCL (Coll-1)
    model = build_unet(args.arch, seed=0 if rank == 0 else 1000 + rank, final_scale=8.0)
    par.broadcast_weights(model, src=0, device=dev)
    dm = DenseModel(model, act_dtype=args.act, device=dev)
    ncls = dm.out_channels
    ids = list(range(1, ncls))
    thr = [127.5] * len(ids)                     # channel_thresholds None -> 255/2 (prediction.py:824-825)

    S, T = args.tile, args.tiles
    tiles = torch.from_numpy(synthetic_em_tiles(T, S, seed=1 + rank)).to(dev)
    B = T if args.batch <= 0 else min(args.batch, T)
    ring = StreamRing(dev, args.streams)         # batch i runs on stream i % n with its own workspace / probability buffer
    probs_k = [torch.empty((B, ncls, S, S, S), dtype=torch.uint8, device=dev) for _ in range(ring.n)]
    probs = probs_k[0][0]
    # label volumes are double-buffered so that the gather of step k (RCCL, asynchronous) overlaps step k+1's compute
    labels = [torch.empty((T, S, S, S), dtype=torch.uint8, device=dev) for _ in range(2)]
    recv = [torch.empty((world, T, S, S, S), dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
            for _ in range(2)]
    pending = [None, None]
    step_no = [0]

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        with ring:
            for i, t0 in enumerate(range(0, T, B)):
                n = min(B, T - t0)
                with ring.stream(i):
                    # U-Net + softmax + uint8 cast + label rule (prediction.py:813-833) in one launch set: the rule is
                    # evaluated in the final layer's epilogue (== postproc_labels(forward_batch(PROBS_U8)), tested)
                    dm.forward_labels_batch(tiles[t0:t0 + n], ids, thr, out=labels[k][t0:t0 + n], slot=ring.slot(i))
        if world > 1:
            _, pending[k] = par.gather_to_root(labels[k], dst=0, async_op=True, out=recv[k])

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(args.warmup):
        step()
    drain()
    nbatch = (T + B - 1) // B                    # launch sets per step (the last one may hold fewer tiles)
    dm.profile(nbatch * args.steps)              # event ring: every launch set of the timed region keeps its own slot
    torch.cuda.synchronize(dev)
    par.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(dev)
    par.barrier()
    torch.cuda.synchronize(dev)
    elapsed = par.max_over_ranks(time.perf_counter() - t0, device=dev)

    vox_total = float(T) * S ** 3 * world * args.steps
    value = vox_total / elapsed / 1e6

    # ---- per-kernel timings from the HIP events recorded inside the timed region (rank 0) ----------------------
    per_op = np.zeros(dm.n_ops)
    n_fw = nbatch * args.steps
    for k in range(n_fw):
        per_op += dm.profile_read(k)
    per_op /= n_fw                                # ms per launch, averaged over the timed region
    tiles_per_launch = T / nbatch                 # average tiles one launch processes
    dm.profile(0)
    rows = [(n, f * tiles_per_launch, b * tiles_per_launch)
            for n, f, b in layer_accounting(dm.ops, L, S, S, S, batch=B)]
    groups = {}
    for (name, fl, by), ms in zip(rows, per_op):
        g = groups.setdefault(name, [0.0, 0.0, 0.0, 0])
        g[0] += fl; g[1] += by; g[2] += ms; g[3] += 1
    dom = max(groups, key=lambda k: groups[k][2])
    fl, by, ms, nlaunch = groups[dom]
    kern_ms = float(per_op.sum())
    b_alg = sum(r[2] for r in rows)
    if fl / max(by, 1) > PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
        roof = {'kernel': dom, 'bound': 'mfma', 'achieved': fl / (ms * 1e-3) / 1e12, 'peak': PEAK_MFMA_TFLOPS,
                'unit': 'TFLOP/s'}
    else:
        roof = {'kernel': dom, 'bound': 'hbm', 'achieved': by / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
    roof['frac'] = roof['achieved'] / roof['peak']
    roof['launches_per_forward'] = nlaunch
    roof['tiles_per_launch'] = tiles_per_launch
    roof['avg_launch_us'] = ms / nlaunch * 1e3
    roof['algorithmic_per_launch'] = (fl if roof['bound'] == 'mfma' else by) / nlaunch
    roof['traffic'] = None
    tr_file = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.isfile(tr_file):
        tr = json.load(open(tr_file))
        if tr.get('arch') == args.arch and tr.get('tile') == S and tr.get('act') == args.act:
            t1 = tr.get('hbm_bytes_per_launch', {}).get(dom)      # measured with tr['tiles_per_launch'] tiles per launch
            roof['traffic'] = None if t1 is None else t1 * tiles_per_launch / float(tr.get('tiles_per_launch', 1))
            roof['traffic_note'] = tr.get('note')
    # whole-network view the north star asks for: algorithmic bytes of one tile / device time of one tile / 8 TB/s
    b_alg /= tiles_per_launch
    kern_ms /= tiles_per_launch
    rows = [(n, f / tiles_per_launch, b / tiles_per_launch) for n, f, b in rows]
    net = {'b_alg_bytes_per_tile': b_alg, 'gflop_per_tile': sum(r[1] for r in rows) / 1e9,
           'kernel_ms_per_tile': kern_ms, 'hbm_roofline_frac': b_alg / (kern_ms * 1e-3) / (PEAK_HBM_GBS * 1e9),
           'effective_tflops': sum(r[1] for r in rows) / (kern_ms * 1e-3) / 1e12,
           'per_kernel_ms_per_tile': {k: round(v[2] / tiles_per_launch, 4) for k, v in groups.items()}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.predictor_ref import label_rule_ref
        ncpu = min(3, T)                                           # bounded sample: ~15 s of CPU work
        xs = tiles[:ncpu].cpu()
        labs = []
        with torch.no_grad():
            model((xs[0, :16].float() / 255.)[None, None])          # warm the CPU kernels
            t1 = time.perf_counter()
            for i in range(ncpu):
                p = model((xs[i].float() / 255.)[None, None]).softmax(1)[0].numpy()
                u8 = (p * 255).astype(np.uint8)
                labs.append(label_rule_ref(u8, ids, [None] * ncls)[0])
            cpu_s = time.perf_counter() - t1
        dm.forward_batch(tiles[:ncpu], L.SD_OUT_PROBS_U8, probs_k[0][:ncpu])
        agree = 0.0
        for i in range(ncpu):
            postproc_labels(probs_k[0][i], ids, thr, out=labels[0][i])
            agree += float((torch.from_numpy(labs[i].astype(np.uint8)) == labels[0][i].cpu()).float().mean()) / ncpu
        cpu = {'value': ncpu * S ** 3 / cpu_s / 1e6, 'unit': 'Mvox/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': f'{ncpu} {S}^3 tiles of the same workload through the torch-CPU fp32 oracle '
                         f'(U-Net + softmax + uint8 + label rule), {cpu_s:.1f} s',
               'label_agreement_with_gpu': agree}

    if rank == 0:
        line = {'metric': 'segmented Mvoxels/s (whole node), 128^3 EM tiles', 'value': value, 'unit': 'Mvox/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': args.act, 'data': 'synthetic',
                'config': {'workload': f'BASELINE configs[1]: {args.arch} 3D U-Net on {S}^3 uint8 tiles, {T} tiles per GPU '
                                       f'per step in launch sets of {B} (sd_forward_batch), random-init weights',
                           'tiles_per_launch_set': B,
                           'tiles_per_gpu_per_step': T, 'tile': [S, S, S], 'parallelism': f'tile-sharded x{world}',
                           'hip_streams_per_gpu': ring.n,
                           'collective': 'gather of uint8 labels to rank 0' if world > 1 else 'none'},
                'roofline': roof, 'network': net, 'cpu_baseline': cpu}
        print(json.dumps(line))
    if world > 1:
        par.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
