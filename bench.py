#!/usr/bin/env python3
"""Headline benchmark of the dense 3D-CNN prediction path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts torch.distributed.run itself as a
                                                           child process; under a launcher -- WORLD_SIZE set -- it is one of the ranks)

metric   : segmented Mvoxels/s (whole node) on synthetic 128^3 uint8 EM tiles  (BASELINE.json `metric`)
workload : BASELINE.json configs[1] -- `semseg_spine` 3D U-Net (myelin trunk, 5 classes),
           bf16 activations / fp32 accumulate, whole 128^3 tile per forward.
step     : one pass of the hot path over one batch of `--tiles` tiles per GPU, HOST TO HOST (BASELINE.md section 2 /
           SURVEY.md section 8d: "uint8 input in host memory to uint8 output in host memory"; the reference does one PCIe
           round trip per tile, /root/reference/syconn/handler/prediction.py:806-809, 863):
               pinned host uint8 tiles --H2D--> [normalise, U-Net, softmax, floor(255 p), label rule
               (prediction.py:813-833)] --> pinned host uint8 label volume
           on three HIP streams (copy-in / compute / copy-out) with two buffer sets, so the copies of steps k-1 and k+1
           overlap the kernels of step k.  On one rank the last kernel of a launch set stores its labels (1 byte per voxel)
           STRAIGHT INTO the page-locked host buffer: the runtime executes a device-to-host hipMemcpyAsync as a blit KERNEL
           here (rocprofv3: __amd_rocclr_copyBuffer, 572 us per 16 MiB; the host-to-device copies go through SDMA), which took
           CUs from the persistent compute kernels -- 2850-2874 -> 2900-2920 Mvox/s on one box, same sha256 of the labels
           (SD_BENCH_D2H_COPY=1: labels into HBM + an explicit copy, the pipeline of rounds 1-4).  With N > 1 every rank feeds its own tiles over its own PCIe link, the label
           volumes are gathered on rank 0 over RCCL / xGMI and leave through rank 0's link (one writer, as the north
           star's "gather of per-chunk logits").
value    = tiles * 128^3 * N * K / (max over ranks of the wall time of K steps), in Mvox/s.  Weak scaling.
           `config.device_resident_value` is the same K steps with inputs and outputs left in HBM (kernel throughput).

Extra objects on the JSON line: `roofline` for the dominant kernel (the 3x3x3 MFMA convolution; live HIP-event
timings recorded on the launch stream inside the timed region), `network` (whole-net HBM-roofline fraction) and
`cpu_baseline` (the torch-CPU fp32 oracle on a bounded sample on this box's host cores, rank 0, N = 1 only, plus the
margin-safe / margin-unsafe split of every label disagreement between the HIP path and that oracle).
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
BENCH_FINAL_SCALE = 8.0      # scale of the random final 1x1x1 weights (spreads the class logits)


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


def layer_accounting(ops, L, D, H, W, act_bytes=2, out_bytes_per_class=1):
    """Algorithmic FLOPs and bytes of every plan op for a (D,H,W) tile, [(flops, bytes)] in op order.  Bytes follow SURVEY.md section
    8(d): activation read + write per fused conv(+norm+act) layer, a pooled tensor counts its write only (it belongs in the producing
    conv's epilogue), concat = two read pointers, weights ignored (L2/MALL resident).  Which KERNEL computed an op is not decided here:
    the library reports it (sd_debug_op_kernel -> DenseModel.op_kernels())."""
    dims = {0: (D, H, W)}
    chans = {0: 1}
    rows = []
    for o in ops:
        if o.kind == L.SD_OP_CONV:
            d = dims[o.src1] if o.src1 >= 0 else dims[o.src0]
            vox = d[0] * d[1] * d[2]
            cin = o.cin0 + max(o.cin1, 0)
            inb = vox * (1 if o.src0 == 0 else cin * act_bytes)
            rows.append((2.0 * vox * cin * o.cout * o.kz * 9, inb + vox * o.cout * act_bytes))
            dims[o.dst], chans[o.dst] = d, o.cout
        elif o.kind == L.SD_OP_POOL:
            d = dims[o.src0]
            do = ((d[0] + 1) // 2 if o.kz == 2 else d[0], (d[1] + 1) // 2, (d[2] + 1) // 2)
            rows.append((0.0, do[0] * do[1] * do[2] * chans[o.src0] * act_bytes))
            dims[o.dst], chans[o.dst] = do, chans[o.src0]
        elif o.kind == L.SD_OP_UPCONV:
            d = dims[o.src0]
            vox = d[0] * d[1] * d[2]
            taps = o.kz * 4
            rows.append((2.0 * vox * o.cin0 * o.cout * taps, (vox * o.cin0 + vox * taps * o.cout) * act_bytes))
            dims[o.dst], chans[o.dst] = (d[0] * o.kz, d[1] * 2, d[2] * 2), o.cout
        elif o.kind == L.SD_OP_GROUPNORM:
            d = dims[o.src1] if o.src1 >= 0 else dims[o.src0]
            rows.append((0.0, 2.0 * d[0] * d[1] * d[2] * chans[o.src0] * act_bytes))
        elif o.kind == L.SD_OP_FINAL:
            d = dims[o.src0]
            vox = d[0] * d[1] * d[2]
            rows.append((2.0 * vox * o.cin0 * o.cout, vox * (o.cin0 * act_bytes + o.cout * out_bytes_per_class)))
        else:
            rows.append((0.0, 0.0))
    return rows


class HostToHostPipeline:
    """K steps of [pinned host tiles -> H2D -> sd_forward_labels_batch -> (RCCL gather on rank 0) -> D2H -> pinned host
    labels] on three HIP streams with two buffer sets.  Step k uses set k % 2; the only host-side wait is for the copy-out
    of step k-2 (its buffers are about to be reused), so two steps are in flight."""

    def __init__(self, dm, tiles_host, ids, thr, batch, dev, par, rank, world):
        self.dm, self.ids, self.thr, self.B, self.dev, self.par, self.rank, self.world = dm, ids, thr, batch, dev, par, rank, world
        T, S = tiles_host.shape[0], tiles_host.shape[1]
        self.T = T
        self.in_host = tiles_host.pin_memory()
        n_out = world if rank == 0 else 1
        self.out_host = [torch.empty((n_out, T, S, S, S), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.in_dev = [torch.empty((T, S, S, S), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.lab_dev = [torch.empty((T, S, S, S), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.coll = par.collectives_active()          # world > 1 (or SD_DIST_SINGLE_RANK_GROUP=1: a group of one, tests)
        self.recv = [torch.empty((world, T, S, S, S), dtype=torch.uint8, device=dev) if (self.coll and rank == 0) else None
                     for _ in range(2)]
        self.s_in, self.s_comp, self.s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
        self.ev_in = [torch.cuda.Event() for _ in range(2)]       # H2D of the set finished
        self.ev_comp = [torch.cuda.Event() for _ in range(2)]     # kernels of the set finished
        self.ev_out = [torch.cuda.Event() for _ in range(2)]      # D2H (and gather) of the set finished
        # one rank: the last kernel stores the labels straight into the page-locked host buffer (module docstring); with a process group
        # the gather needs them in HBM
        self.zero_copy = not os.environ.get('SD_BENCH_D2H_COPY') and not self.coll
        # SD_BENCH_D2H_REPEAT=n (one rank with SD_BENCH_D2H_COPY=1, A/B only): the labels leave n times per step -- what rank 0's device-to-host
        # leg carries at N = n ranks (measured, n = 8: 2986 -> 2845 Mvox/s = -4.7 %; a lean 8-workgroup copy kernel in the place of the
        # runtime's blit kernel was WORSE, -12 %: tools/experiments/round6_notes.md)
        self.d2h_repeat = max(1, int(os.environ.get('SD_BENCH_D2H_REPEAT', '1'))) if not self.coll else 1
        self.extra_host = [[torch.empty((T, S, S, S), dtype=torch.uint8).pin_memory() for _ in range(self.d2h_repeat - 1)] for _ in range(2)]
        self.k = 0

    def step(self):
        s = self.k & 1
        first_use = self.k < 2
        self.k += 1
        if not first_use:
            self.ev_out[s].synchronize()              # host: the pinned output of step k-2 is complete (and reusable)
        # (the host has just waited for the copy-out of step k-2, which is ordered behind that step's kernels: in_dev[s] and
        # lab_dev[s] are free -- no device-side waits for step k-2 are needed, and every cross-stream wait queued between two
        # kernels costs the compute stream a bubble)
        with torch.cuda.stream(self.s_in):
            self.in_dev[s].copy_(self.in_host, non_blocking=True)
            self.ev_in[s].record(self.s_in)
        with torch.cuda.stream(self.s_comp):
            self.s_comp.wait_event(self.ev_in[s])
            for t0 in range(0, self.T, self.B):
                n = min(self.B, self.T - t0)
                dst = self.out_host[s][0] if self.zero_copy else self.lab_dev[s]
                self.dm.forward_labels_batch(self.in_dev[s][t0:t0 + n], self.ids, self.thr, out=dst[t0:t0 + n])
            self.ev_comp[s].record(self.s_comp)
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(self.ev_comp[s])
            if self.coll:
                _, work = self.par.gather_to_root(self.lab_dev[s], dst=0, async_op=True, out=self.recv[s])
                work.wait()                             # stream-level dependency of s_out on the collective
                if self.rank == 0:
                    self.out_host[s].copy_(self.recv[s], non_blocking=True)
            elif not self.zero_copy:
                self.out_host[s][0].copy_(self.lab_dev[s], non_blocking=True)
                for extra in self.extra_host[s]:
                    extra.copy_(self.lab_dev[s], non_blocking=True)
            self.ev_out[s].record(self.s_out)

    def drain(self):
        for e in self.ev_out:
            e.synchronize()
        torch.cuda.synchronize(self.dev)


def self_launch(n: int) -> int:
    """``python bench.py --gpus N`` (N > 1) without a launcher: start ``torch.distributed.run`` with one rank per GPU as a CHILD process --
    this process has not touched the GPU (``import torch`` alone does not), it never execs, it only relays the child's output and
    return code.  The reference's counterpart is one worker process per GPU (/root/reference/syconn/handler/prediction.py:708-719)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    sys.stdout.flush()
    raise SystemExit(proc.returncode)


def dist_info(par):
    """What the process group looks like from inside (so that a reader of the JSON line sees that RCCL saw N ranks)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {'world_size': 1, 'backend': None, 'rccl_version': None}
    backend = dist.get_backend()
    ver = None
    if backend == 'nccl':
        try:
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:      # noqa: BLE001  (a version string is decoration, never a reason to fail the run)
            ver = 'unknown'
    return {'world_size': dist.get_world_size(), 'backend': backend, 'rccl_version': ver}


def timed(fn_step, fn_drain, steps, par, dev):
    torch.cuda.synchronize(dev)
    par.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn_step()
    fn_drain()
    torch.cuda.synchronize(dev)
    mine = time.perf_counter() - t0          # this rank's own K steps (before it waits for the others): who the straggler is
    par.barrier()
    torch.cuda.synchronize(dev)
    total = par.max_over_ranks(time.perf_counter() - t0, device=dev)
    timed.rank_seconds = par.values_of_all_ranks(mine, device=dev)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--tiles', type=int, default=96, help='128^3 tiles per GPU per step (96 tiles in launch sets of 8: 20 steps are a\n'
                    '                    timed region of >= 1.3 s -- long enough for the clock to settle and for an SMI sampler to see the load)')
    ap.add_argument('--tile', type=int, default=128)
    ap.add_argument('--arch', default='semseg_spine')
    ap.add_argument('--act', default=None, choices=['bf16', 'f16', 'f16x2', 'f32'],
                    help="activation storage type (default: bf16 = what BASELINE configs[1] names; volume workloads: their config's)")
    ap.add_argument('--batch', type=int, default=8, help='tiles per sd_forward_batch launch set (0 = all tiles of a step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--predict-outside', action='store_true',
                    help='volume workloads: also predict model tiles that lie entirely beyond the volume, as the reference does')
    ap.add_argument('--full-windows', action='store_true',
                    help='volume workloads: predict tiles that reach beyond the volume on their full window (no clipping)')
    ap.add_argument('--labels-sha', action='store_true', help='report a sha256 of the result volume(s) on the JSON line (tests)')
    ap.add_argument('--workload', default='config2', choices=['config2', 'config3', 'config4', 'config5'],
                    help='BASELINE.json configs[1..4]; the default (config2 = configs[1]) is the headline metric')
    ap.add_argument('--geometry', default='tile128', choices=['tile128', 'reference'],
                    help='volume workloads: 128^3 model tiles (SURVEY 8d) or the reference chunk / tile geometry')
    ap.add_argument('--volume', type=int, nargs=3, default=None, help='volume workloads: z y x of the synthetic volume')
    ap.add_argument('--zmajor-rounds', action='store_true', help='volume workloads: deal the chunks in z-major order (A/B against cost-sorted rounds)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)
    if args.workload != 'config2':
        return volume_main(args)
    args.act = args.act or 'bf16'

    from syconn_amd import _lib as L
    from syconn_amd import parallel as par
    from syconn_amd.cnn import random_state_dict    # architecture + seeded random init (no trained weights exist)
    from syconn_amd.engine import DenseModel

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
    sd = random_state_dict(args.arch, seed=0 if rank == 0 else 1000 + rank, final_scale=BENCH_FINAL_SCALE)
    par.broadcast_weights(sd, src=0, device=dev)
    dm = DenseModel(sd, act_dtype=args.act, device=dev)
    ncls = dm.out_channels
    ids = list(range(1, ncls))
    thr = [127.5] * len(ids)                     # channel_thresholds None -> 255/2 (prediction.py:824-825)

    S, T = args.tile, args.tiles
    tiles_host = torch.from_numpy(synthetic_em_tiles(T, S, seed=1 + rank))
    B = T if args.batch <= 0 else min(args.batch, T)
    nbatch = (T + B - 1) // B                    # launch sets per step (the last one may hold fewer tiles)
    pipe = HostToHostPipeline(dm, tiles_host, ids, thr, B, dev, par, rank, world)

    # ---- timed region: host -> host ------------------------------------------------------------------------------
    for _ in range(args.warmup):
        pipe.step()
    pipe.drain()
    dm.profile(nbatch * args.steps)              # event ring: every launch set of the timed region keeps its own slot
    elapsed = timed(pipe.step, pipe.drain, args.steps, par, dev)
    rank_seconds = list(timed.rank_seconds)
    vox_total = float(T) * S ** 3 * world * args.steps
    value = vox_total / elapsed / 1e6

    # ---- per-kernel timings from the HIP events recorded inside the timed region (rank 0) ----------------------
    per_op = np.zeros(dm.n_ops)
    n_fw = nbatch * args.steps
    for k in range(n_fw):
        per_op += dm.profile_read(k)
    per_op /= n_fw                                # ms per launch, averaged over the timed region
    # shader clock of every convolution launch of the timed region (stamped by the launch's first workgroup): (n_fw, n_ops) GHz, 0 = no stamp
    # GHz of every stamped launch = cycles / (100 MHz ticks * 10 ns) between entry and exit of its first workgroup.  A launch whose first
    # workgroup lived < 10 us says nothing, and about 3 % of the stamps carry a cycle counter that jumped by ~2^23.6 between the two
    # reads (seen on every box; the tick counter never does): everything outside 0.3 ... 3 GHz is dropped and the statistic is a median
    clk = np.full((n_fw, dm.n_ops), np.nan)
    for k in range(n_fw):
        st = dm.profile_read_clocks(k).astype(np.float64)
        dt = st[:, 3] - st[:, 1]
        ok = dt >= 1000
        g = np.full(dm.n_ops, np.nan)
        g[ok] = (st[ok, 2] - st[ok, 0]) / (dt[ok] * 10.0)
        g[(g < 0.3) | (g > 3.0)] = np.nan
        clk[k] = g
    if os.environ.get('SD_BENCH_DUMP_CLOCKS'):      # (debugging aid: raw stamps and launch times of every launch set)
        np.savez(os.environ['SD_BENCH_DUMP_CLOCKS'], stamps=np.stack([dm.profile_read_clocks(k) for k in range(n_fw)]),
                 ms=np.stack([dm.profile_read(k) for k in range(n_fw)]))
    dm.profile(0)

    # ---- the same K steps with inputs / outputs resident in HBM (kernel throughput; NOT `value`) ------------------
    tiles_dev = tiles_host.to(dev)
    lab_res = torch.empty((T, S, S, S), dtype=torch.uint8, device=dev)

    def resident_step():
        for t0 in range(0, T, B):
            n = min(B, T - t0)
            dm.forward_labels_batch(tiles_dev[t0:t0 + n], ids, thr, out=lab_res[t0:t0 + n])

    resident_step()
    elapsed_res = timed(resident_step, lambda: None, args.steps, par, dev)
    value_res = vox_total / elapsed_res / 1e6
    # both legs compute the same labels
    if world == 1:
        assert torch.equal(lab_res.cpu(), pipe.out_host[(pipe.k - 1) & 1][0]), 'host-to-host labels differ from resident run'

    # ---- what BASELINE configs[1] literally names -- ONE 128^3 tile per launch set -- and the reference-precision plan, device-resident
    def resident_rate(model, n_tiles, min_reps=5, budget_s=0.4):
        x, o = tiles_dev[:n_tiles], torch.empty((n_tiles, S, S, S), dtype=torch.uint8, device=dev)
        model.forward_labels_batch(x, ids, thr, out=o)
        torch.cuda.synchronize(dev)
        reps, t1 = 0, time.perf_counter()
        while reps < min_reps or (time.perf_counter() - t1 < budget_s and reps < 200):
            model.forward_labels_batch(x, ids, thr, out=o)
            reps += 1
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t1) / reps / n_tiles * 1e3      # ms per tile
    # (not in the rocprofv3 passes, which run with --no-cpu-baseline: their per-kernel averages must be those of the timed launch sets)
    extras = not args.no_cpu_baseline
    single_tile_ms = resident_rate(dm, 1) if extras else None
    ref_prec = None
    if extras and world == 1 and args.act != 'f16x2':
        m2 = DenseModel(sd, act_dtype='f16x2', device=dev)
        ref_prec = resident_rate(m2, T)
        del m2
    dm.forward_labels_batch(tiles_dev[:B], ids, thr, out=lab_res[:B])      # (the launch set op_kernels() below reports on)

    tiles_per_launch = T / nbatch                 # average tiles one launch processes
    rows = [(f * tiles_per_launch, b * tiles_per_launch) for f, b in layer_accounting(dm.ops, L, S, S, S)]
    # kernel of every op as the library ran it in the last forward (fused ops: the launch they ran inside); the algorithmic work of an op
    # is booked on that launch, its HIP-event time is the launch's
    executed = dm.op_kernels()
    groups = {}
    for i, ((fl, by), ms) in enumerate(zip(rows, per_op)):
        e, name = executed[i]
        g = groups.setdefault(name or 'not run', [0.0, 0.0, 0.0, 0])
        g[0] += fl; g[1] += by; g[2] += ms; g[3] += 1 if e == i else 0
    dom = max(groups, key=lambda k: groups[k][2])
    fl, by, ms, nlaunch = groups[dom]
    kern_ms = float(per_op.sum())
    b_alg = sum(r[1] for r in rows)
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
    if args.act == 'f16x2':
        # the split-fp16 plan executes THREE fp16 MFMA passes per algorithmic product (Wlo.Xhi + Whi.Xhi + Whi.Xlo): `achieved` stays
        # algorithmic (what the bench contract asks for), the matrix pipe does three times that
        roof['mfma_passes_per_product'] = 3
        roof['executed_frac_of_peak'] = 3 * roof['frac'] if roof['bound'] == 'mfma' else None
    # the same run's view of THIS box: shader clock of the dominant kernel's launches inside the timed region, and what a chip-wide pure
    # MFMA loop sustains here right after it (the part is power-limited under matrix load and boxes differ: DESIGN.md section 5) --
    # `frac` is against the nominal 2.5 PFLOP/s, `frac_of_sustained` against what this box's matrix pipes deliver at their own clock
    dom_ops = [i for i in range(dm.n_ops) if executed[i][1] == dom and executed[i][0] == i]
    def med(a):
        a = a[np.isfinite(a)]
        return (float(np.median(a)), [float(np.percentile(a, 10)), float(np.percentile(a, 90))], int(a.size)) if a.size else (None, None, 0)
    roof['clock_ghz_timed_region'], roof['clock_ghz_timed_region_p10_p90'], roof['clock_stamps_used'] = med(clk[:, dom_ops]) if dom_ops else (None, None, 0)
    roof['clock_ghz_all_conv_launches'] = med(clk)[0]
    if roof['bound'] == 'mfma' and roof['clock_ghz_timed_region']:
        # the matrix peak is a 2.4 GHz figure: what fraction of the MFMA issue slots of the clock the launches really ran at was used
        roof['frac_at_measured_clock'] = roof['achieved'] / (PEAK_MFMA_TFLOPS * roof['clock_ghz_timed_region'] / 2.4)
    if roof['bound'] == 'mfma' and rank == 0:
        from syconn_amd.engine import probe_mfma_rate
        sus_tf, sus_ghz = probe_mfma_rate(dev, random_operands=True)
        cst_tf, cst_ghz = probe_mfma_rate(dev, random_operands=False)
        roof['sustained_mfma_tflops_this_box'] = sus_tf            # pure MFMA loop on pseudo-random operands (what data looks like)
        roof['sustained_clock_ghz_this_box'] = sus_ghz
        roof['sustained_mfma_tflops_constant_operands'] = cst_tf   # the same loop on constants: nothing toggles, less power, higher clock
        roof['sustained_clock_ghz_constant_operands'] = cst_ghz
        roof['frac_of_sustained'] = roof['achieved'] / sus_tf if sus_tf > 0 else None
    roof['traffic'] = None
    tr_file = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.isfile(tr_file):
        tr = json.load(open(tr_file))
        if tr.get('arch') == args.arch and tr.get('tile') == S and tr.get('act') == args.act:
            per = tr.get('hbm_bytes_per_launch', {})                # measured with tr['tiles_per_launch'] tiles per launch
            # (rocprofv3 prints the planar 16-bit forms without their storage type: those entries carry a '*' there)
            t1 = per.get(dom, per.get(dom.replace('<bf16,', '<*,').replace('<f16,', '<*,')))
            roof['traffic'] = None if t1 is None else t1 * tiles_per_launch / float(tr.get('tiles_per_launch', 1))
            roof['traffic_note'] = tr.get('note')
    # whole-network view the north star asks for: algorithmic bytes of one tile / device time of one tile / 8 TB/s
    b_alg /= tiles_per_launch
    kern_ms /= tiles_per_launch
    rows = [(f / tiles_per_launch, b / tiles_per_launch) for f, b in rows]
    net = {'b_alg_bytes_per_tile': b_alg, 'gflop_per_tile': sum(r[0] for r in rows) / 1e9,
           'kernel_ms_per_tile': kern_ms, 'hbm_roofline_frac': b_alg / (kern_ms * 1e-3) / (PEAK_HBM_GBS * 1e9),
           'effective_tflops': sum(r[0] for r in rows) / (kern_ms * 1e-3) / 1e12,
           'per_kernel_ms_per_tile': {k: round(v[2] / tiles_per_launch, 4) for k, v in groups.items()}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sd, dm, tiles_host[:B], ids, L)

    labels_sha = None
    if rank == 0 and args.labels_sha:
        # checksum of the label volumes of the LAST step as they arrived in rank 0's host memory, rank by rank (tests compare
        # an N-rank run with the same tiles predicted by one rank)
        import hashlib
        pipe.drain()
        labels_sha = hashlib.sha256(pipe.out_host[(pipe.k - 1) & 1].numpy().tobytes()).hexdigest()
    if rank == 0:
        line = {'metric': 'segmented Mvoxels/s (whole node), 128^3 EM tiles', 'value': value, 'unit': 'Mvox/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': args.act, 'data': 'synthetic',
                'config': {'workload': f'BASELINE configs[1]: {args.arch} 3D U-Net on {S}^3 uint8 tiles, {T} tiles per GPU '
                                       f'per step in launch sets of {B} (sd_forward_labels_batch), random-init weights, '
                                       f'HOST TO HOST: pinned host uint8 tiles -> H2D -> kernels -> '
                                       + ('pinned host uint8 labels (stored by the last kernel, no copy)' if pipe.zero_copy
                                          else 'D2H -> pinned host uint8 labels')
                                       + (f' x{pipe.d2h_repeat}' if pipe.d2h_repeat > 1 else '') + ', 3 HIP streams, 2 buffer sets',
                           'tiles_per_launch_set': B,
                           'tiles_per_gpu_per_step': T, 'tile': [S, S, S], 'parallelism': f'tile-sharded x{world}',
                           'hip_streams_per_gpu': 3,
                           'collective': 'RCCL gather of uint8 labels to rank 0, D2H there' if par.collectives_active() else 'none',
                           'single_tile_ms': single_tile_ms,
                           'single_tile_mvox_per_s': None if single_tile_ms is None else S ** 3 / single_tile_ms / 1e3,
                           'reference_precision_f16x2_mvox_per_s': None if ref_prec is None else S ** 3 / ref_prec / 1e3,
                           'reference_precision_f16x2_ms_per_tile': ref_prec,
                           'device_resident_value': value_res,
                           'device_resident_ms_per_step': elapsed_res / args.steps * 1e3,
                           'pcie_bytes_per_step_each_way': T * S ** 3, 'labels_sha256': labels_sha},
                'roofline': roof, 'network': net, 'cpu_baseline': cpu,
                'distributed': dict(dist_info(par), rank_ms_per_step=[round(v / args.steps * 1e3, 3) for v in rank_seconds])}
        print(json.dumps(line))
    if par.collectives_active():
        par.barrier()
        torch.distributed.destroy_process_group()


# ---- BASELINE configs[2..4]: whole volumes, chunk-parallel ------------------------------------------------------------
VOLUME_WORKLOADS = {
    # name: (arch, act, default volume z,y,x, what BASELINE.json calls it)
    'config3': ('semseg_axon', 'bf16', (512, 512, 512), 'configs[2]: semseg_axon 3D U-Net, 512^3 volume in overlapping 128^3 tiles'),
    'config4': ('myelin', 'bf16', (512, 2048, 2048), 'configs[3]: 2048x2048x512 synthetic KnossosDataset volume, chunk-parallel, '
                                                     'RCCL scatter / gather'),
    'config5': ('mivcsj', 'f16', (512, 2048, 2048), 'configs[4]: 3-head mito/vesicle/synapse dense prediction, fp16, '
                                                    'overlap-and-crop stitching'),
}


def synthetic_volume(shape, seed):
    """(z,y,x) uint8 volume: a 128^3 block of structured synthetic EM repeated with flips (cheap for 2 GiB volumes)."""
    blk = synthetic_em_tiles(1, 128, seed)[0]
    reps = [-(-s // 128) for s in shape]
    vol = np.tile(blk, reps)[:shape[0], :shape[1], :shape[2]]
    return np.ascontiguousarray(vol)


def volume_main(args):
    """One step = the whole volume, host to host: rank 0 holds the uint8 volume in host memory, chunks (+ halo) are dealt
    round-robin to the ranks (== chunkify, /root/reference/syconn/handler/prediction.py:708-709) over RCCL, every rank
    predicts its chunks tile by tile, the uint8 results are gathered and stitched in rank 0's host memory
    (syconn_amd.parallel.predict_volume_distributed, scatter / predict / gather overlapped).  Strong scaling."""
    from syconn_amd import parallel as par
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.handler.prediction import Predictor
    arch, act, vol_default, what = VOLUME_WORKLOADS[args.workload]
    act = args.act or act
    vol_shape = tuple(args.volume) if args.volume else vol_default
    one_gpu_debug = bool(os.environ.get('SD_BENCH_ONE_GPU_DEBUG'))
    rank, world, local_rank = par.init_distributed('gloo' if one_gpu_debug else None)
    if one_gpu_debug:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    # (random myelin weights of seed 0 answer every voxel with class 0 at p = 254 / 255: a constant result volume would make the
    # equality checks of this workload vacuous; seed 3 spreads channel 1 over 47 ... 138)
    base_seed = 3 if arch == 'myelin' else 0
    sd = random_state_dict(arch, seed=base_seed if rank == 0 else 1000 + rank, final_scale=BENCH_FINAL_SCALE)
    par.broadcast_weights(sd, src=0, device=dev)
    if args.geometry == 'reference':      # prediction.py:672-677 (x,y,z) -> (z,y,x)
        chunk, halo, tile = (236, 481, 482), (20, 31, 30), (138, 181, 271)
    else:                                 # model tile 128^3 = useful (112,96,96) + halo (8,16,16); 2x2x2 tiles per chunk
        chunk, halo, tile = (224, 192, 192), (8, 16, 16), (112, 96, 96)
    pred = Predictor(sd, device=dev, tile_shape=tile, overlap_shape=halo, apply_softmax=True, act_dtype=act,
                     defer_guard=True, clip_tiles=not args.full_windows)      # (fp16 range guard asked once per volume: no sync per chunk inside the pipeline)
    ncls = pred.out_channels
    ids, thr = list(range(1, ncls)), [127.5] * (ncls - 1)      # channel_thresholds None -> 255/2 (prediction.py:824-825)
    in_halo = args.geometry != 'reference'

    skip_outside = not args.predict_outside

    def predict_fn(ch, valid_box=None):
        """chunk + halo (uint8, device) -> (1, *chunk) uint8: a multi-id target gives the label volume (mivcsj: ids 1,2,3),
        a two-class model the probability map of channel 1 (myelin), as exec_dense_prediction binds them.
        tile128: the halo is real neighbouring data and the tile grid continues across chunks; reference: the chunk + halo
        is zero-padded and tiled like /root/reference/syconn/handler/prediction.py:775-781, then the halo is cropped (:812)."""
        # model tiles whose result lies entirely beyond the volume are not predicted (Predictor._tiled, `valid_box` in the
        # coordinates of the Predictor's output: chunk + halo for the reference geometry, chunk proper for tile128)
        vb = None
        if valid_box is not None and skip_outside:
            vb = valid_box if not in_halo else (tuple(a - h for a, h in zip(valid_box[0], halo)),
                                                tuple(b - h for b, h in zip(valid_box[1], halo)))
        if ncls > 2:
            r = pred.predict_labels_u8_device(ch, ids, thr, halo_included=in_halo, valid_box=vb)[None]
        else:
            r = pred.predict_proba_u8_device(ch, halo_included=in_halo, valid_box=vb)[1:2]
        return r if in_halo else _crop(r, halo)
    vol = torch.from_numpy(synthetic_volume(vol_shape, seed=3)).pin_memory() if rank == 0 else None
    # (the result arrives in ONE page-locked host tensor, reused by every step: rank 0's part of a step is two contiguous PCIe streams,
    # volume in and result out -- chunk + halo boxes are cut and results placed on its GPU, parallel.predict_volume_distributed)
    res_host = torch.empty((1, *vol_shape), dtype=torch.uint8).pin_memory() if rank == 0 else None
    steps, warm = (args.steps if args.steps != 20 else 2), min(args.warmup, 1)
    # rounds dealt from a cost-sorted chunk list (cost = voxels of all windows the chunk's predicted tiles run on, from geometry):
    # the chunks of a lock-step round then cost the same (parallel.predict_volume_distributed; --zmajor-rounds: the old order)
    cm = pred.chunk_cost_model(halo, in_halo, skip_outside)
    chunk_cost = None if args.zmajor_rounds else (lambda vb: cm.chunk_cost(chunk, vb))
    for _ in range(warm):
        par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=1, device=dev, out=res_host, chunk_cost=chunk_cost)
    out = [None]

    def step():
        out[0] = par.predict_volume_distributed(vol, vol_shape, chunk, halo, predict_fn, n_out=1, device=dev, out=res_host,
                                                chunk_cost=chunk_cost)
    elapsed = timed(step, lambda: None, steps, par, dev)
    rank_seconds = list(timed.rank_seconds)
    if pred.overflowed():
        raise SystemExit('fp16 activation overflow during the volume workload: rerun with bf16')
    nvox = float(np.prod(vol_shape))
    if rank == 0:
        nchunks = int(np.prod([-(-v // c) for v, c in zip(vol_shape, chunk)]))
        line = {'metric': 'segmented Mvoxels/s (whole node), whole volume host to host', 'value': nvox * steps / elapsed / 1e6,
                'unit': 'Mvox/s', 'n_gpus': world, 'steps': steps, 'warmup': warm, 'ms_per_step': elapsed / steps * 1e3,
                'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': act, 'data': 'synthetic',
                'config': {'workload': f'BASELINE {what}; {arch}, volume z,y,x = {vol_shape}, geometry {args.geometry}: chunks '
                                       f'{chunk} + halo {halo}, model tiles {tuple(t + 2 * h for t, h in zip(tile, halo))}, '
                                       f'{nchunks} chunks dealt round-robin over {world} rank(s)',
                           'parallelism': f'chunk-sharded x{world}',
                           'output_distinct_values': int(torch.unique(out[0][0, ::4, ::8, ::8]).numel()),      # (a strided sample)
                           'labels_sha256': (__import__('hashlib').sha256(out[0].numpy().tobytes()).hexdigest()
                                             if args.labels_sha else None),
                           'host_box_copies': par.HOST_BOX_COPIES,      # (0: rank 0's CPU cut and stitched nothing)
                           'round_order': 'z-major' if args.zmajor_rounds else 'cost-sorted (descending window voxels, stable)',
                           'modelled_speedup_8_ranks': dict(zip(('z_major', 'cost_sorted'), _modelled(par, cm, vol_shape, chunk, halo))),
                           'collective': 'RCCL scatter of uint8 chunks / gather of uint8 results; volume and result resident in rank 0\'s HBM, '
                                         'one contiguous PCIe stream each way' if par.collectives_active() else 'none'},
                'distributed': dict(dist_info(par), rank_ms_per_step=[round(v / steps * 1e3, 3) for v in rank_seconds])}
        print(json.dumps(line))
    if par.collectives_active():
        par.barrier()
        torch.distributed.destroy_process_group()


def _modelled(par, cm, vol_shape, chunk, halo, world=8):
    """What lock-step rounds can reach on `world` ranks by chunk cost alone (no communication): z-major vs cost-sorted deal order."""
    ids, boxes = par.chunk_grid(vol_shape, chunk, halo)
    costs = [cm.chunk_cost(chunk, b) for b in boxes]
    order = par.cost_sorted(list(range(len(ids))), costs)
    return (round(par.round_schedule_speedup(costs, world), 3), round(par.round_schedule_speedup([costs[i] for i in order], world), 3))


def _crop(t, halo):
    return t[:, halo[0]:t.shape[1] - halo[0], halo[1]:t.shape[2] - halo[1], halo[2]:t.shape[3] - halo[2]].contiguous()


def cpu_baseline(args, sd, dm, tiles_host, ids, L):
    """The torch-CPU fp32 oracle (U-Net + softmax + uint8 + label rule) on a bounded sample of the same workload, timed
    on this box's host cores, and the margin-safe / margin-unsafe split of every label disagreement with the HIP path."""
    from oracle.label_margin import label_split, merge_splits, stated_tolerance
    from oracle.predictor_ref import label_rule_ref
    from oracle.unet_ref import ARCHS, UNet
    from syconn_amd.engine import DenseModel
    ncls = dm.out_channels
    model = UNet(in_channels=1, **ARCHS[args.arch]).eval()
    model.load_state_dict(sd)
    ncpu = min(3, tiles_host.shape[0])                         # bounded sample: ~15 s of CPU work
    S = tiles_host.shape[1]
    ref = []
    with torch.no_grad():
        model((tiles_host[0, :16].float() / 255.)[None, None])     # warm the CPU kernels
        cpu_s = 0.0
        for i in range(ncpu):
            t1 = time.perf_counter()
            lg = model((tiles_host[i].float() / 255.)[None, None])[0]
            u8 = (lg.softmax(0).numpy() * 255).astype(np.uint8)
            label_rule_ref(u8, ids, [None] * ncls)
            cpu_s += time.perf_counter() - t1
            ref.append(lg)

    def split_for(m, act):
        parts = []
        for i in range(ncpu):
            x = tiles_host[i:i + 1].to(m.device)
            parts.append(label_split(ref[i], m.forward_batch(x, L.SD_OUT_LOGITS_F32)[0].cpu(),
                                     m.forward_batch(x, L.SD_OUT_PROBS_F32)[0].cpu(),
                                     m.forward_labels_batch(x, ids, [127.5] * len(ids))[0].cpu(), ids, [None] * ncls,
                                     stated_tolerance(args.arch, act)))
        return merge_splits(parts)

    keys = ('label_agreement', 'label_mismatch_safe', 'label_mismatch_unsafe', 'label_unsafe_frac', 'argmax_agreement',
            'argmax_mismatch_safe', 'argmax_mismatch_unsafe', 'argmax_unsafe_frac', 'tol_logit_rel_stated', 'logit_err_max_rel',
            'logit_err_rms', 'median_top2_margin_over_tol', 'label_unsafe_frac_2x_measured_err',
            'argmax_unsafe_frac_2x_measured_err')
    def time_for(m):
        # device-resident throughput of this storage type on the bench's own launch set (all tiles of a step in one set)
        x = tiles_host.to(m.device)
        out = torch.empty(tuple(x.shape), dtype=torch.uint8, device=m.device)
        m.forward_labels_batch(x, ids, [127.5] * len(ids), out=out)
        torch.cuda.synchronize(m.device)
        reps, t1 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t1 < 0.5 and reps < 50):
            m.forward_labels_batch(x, ids, [127.5] * len(ids), out=out)
            torch.cuda.synchronize(m.device)
            reps += 1
        ms = (time.perf_counter() - t1) / reps / x.shape[0] * 1e3
        return {'ms_per_tile': ms, 'mvox_per_s': S ** 3 / ms / 1e3, 'tiles_per_launch_set': int(x.shape[0])}

    sp = split_for(dm, args.act)
    cpu = {'value': ncpu * S ** 3 / cpu_s / 1e6, 'unit': 'Mvox/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': f'{ncpu} {S}^3 tiles of the same workload through the torch-CPU fp32 oracle '
                     f'(U-Net + softmax + uint8 + label rule), {cpu_s:.1f} s'}
    cpu.update({k: sp[k] for k in keys})
    # the same tiles in every storage type of the library, each with its agreement AND its device-resident throughput:
    # 'f16x2' = the reference-precision plan on the matrix cores (what Predictor(float16=False) selects, csrc/sd_split.hip),
    # 'f32' = fp32 storage + FMA arithmetic (csrc/sd_f32.hip), 'f16' / 'bf16' = the fast plans
    modes = {args.act: dict({k: sp[k] for k in keys}, **time_for(dm))}
    for act in ('bf16', 'f16', 'f16x2', 'f32'):
        if act in modes:
            continue
        m2 = DenseModel(sd, act_dtype=act, device=dm.device)
        r = split_for(m2, act)
        modes[act] = dict({k: r[k] for k in keys}, **time_for(m2))
        del m2
    cpu['precision_modes'] = modes
    return cpu

if __name__ == '__main__':
    main()
