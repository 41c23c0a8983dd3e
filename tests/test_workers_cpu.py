"""Multi-worker host logic of the dense path, no GPU needed:

* two worker PROCESSES whose chunks share target cubes must produce the same KnossosDataset as one worker
  (chunks of 482x481x236 are not aligned to the 256^3 target cubes, /root/reference/syconn/handler/prediction.py:672-677,
  700-702: cube files on chunk borders are read-modify-written by different workers);
* ``batchjob_script`` never runs two jobs on one GPU at the same time, also with more jobs than GPUs
  (/root/reference/syconn/mp/batchjob_utils.py:390-516 does not pin devices at all).
"""
import multiprocessing as mp
import os
import pickle
import sys
import time

import numpy as np
import pytest

from syconn_amd.knossos import KnossosDataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vol(shape_zyx, seed):
    return np.random.default_rng(seed).integers(1, 255, shape_zyx, dtype=np.uint8)


def _writer(path, boxes, seed, ext, barrier, combine=False):
    kd = KnossosDataset().initialize_from_knossos_path(path)
    if combine:
        kd.enable_write_combining(max_cubes=3)          # (tiny cache: evictions = read-modify-write merges in mid-run)
    barrier.wait()
    for rep in range(1 if combine else 3):                       # repeated passes widen the race window; the content is idempotent
        for off_xyz, size_xyz in boxes:
            z, y, x = off_xyz[2], off_xyz[1], off_xyz[0]
            data = _vol((96, 96, 96), seed)[z:z + size_xyz[2], y:y + size_xyz[1], x:x + size_xyz[0]]
            if ext == 'raw':
                kd.save_raw(offset=np.array(off_xyz), mags=[1, 2], data=data, data_mag=1, fast_resampling=True,
                            upsample=False)
            else:
                kd.save_seg(offset=np.array(off_xyz), mags=[1], data=data.astype(np.uint64), data_mag=1,
                            fast_resampling=True, upsample=False)
    kd.flush()


def _boxes():
    # 24^3 chunks of a 96^3 volume stored in 64^3 cubes: every cube is shared by up to 27 chunks
    out = []
    for x in range(0, 96, 24):
        for y in range(0, 96, 24):
            for z in range(0, 96, 24):
                out.append(((x, y, z), (24, 24, 24)))
    return out


def _make_kd(path):
    kd = KnossosDataset()
    kd._cube_shape = (64, 64, 64)
    kd.initialize_without_conf(str(path), np.array([96, 96, 96]), np.array([10., 10., 20.]), 'race', mags=[1, 2])
    return kd


@pytest.mark.parametrize('combine', [False, True])
def test_two_processes_sharing_cubes_lose_no_update(tmp_path, combine):
    """`combine`: every writer assembles the cubes it touches in memory (KnossosDataset.enable_write_combining: complete cubes
    are written once, partial ones merged under the cube lock at eviction / flush) -- same dataset, and no lock file is
    left behind either way."""
    boxes = _boxes()
    for ext in ('raw', 'seg'):
        p = tmp_path / f'kd_{ext}'
        _make_kd(p)
        ctx = mp.get_context('fork')
        barrier = ctx.Barrier(2)
        procs = [ctx.Process(target=_writer, args=(str(p), boxes[i::2], 7, ext, barrier, combine)) for i in range(2)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(120)
            assert pr.exitcode == 0
        kd = KnossosDataset().initialize_from_knossos_path(str(p))
        want = _vol((96, 96, 96), 7)
        if ext == 'raw':
            got = kd.load_raw(size=np.array([96, 96, 96]), offset=np.zeros(3, int), mag=1)
            assert np.array_equal(got, want)
            got2 = kd.load_raw(size=np.array([96, 96, 96]), offset=np.zeros(3, int), mag=2)
            # every 24^3 chunk contributes its own order-0 pyramid level (chunk origins are even -> same sample grid)
            assert np.array_equal(got2, want[::2, ::2, ::2])
        else:
            got = kd.load_seg(size=np.array([96, 96, 96]), offset=np.zeros(3, int), mag=1)
            assert np.array_equal(got, want.astype(np.uint64))
        assert not [f for _, _, fs in os.walk(p) for f in fs if f.endswith('.lock')]


def test_write_combining_writes_complete_cubes_once(tmp_path, monkeypatch):
    """one writer covering the whole dataset: with write combining every cube file is written exactly once and never read"""
    p = tmp_path / 'kd'
    _make_kd(p)
    kd = KnossosDataset().initialize_from_knossos_path(str(p))
    kd.enable_write_combining(max_cubes=64)
    n_write, n_read = [0], [0]
    w0, r0 = kd._write_cube, kd._read_cube
    monkeypatch.setattr(kd, '_write_cube', lambda *a, **k: (n_write.__setitem__(0, n_write[0] + 1), w0(*a, **k))[1])
    monkeypatch.setattr(kd, '_read_cube', lambda *a, **k: (n_read.__setitem__(0, n_read[0] + 1), r0(*a, **k))[1])
    want = _vol((96, 96, 96), 3)
    for off_xyz, size_xyz in _boxes():
        z, y, x = off_xyz[2], off_xyz[1], off_xyz[0]
        kd.save_raw(offset=np.array(off_xyz), mags=[1], data=want[z:z + 24, y:y + 24, x:x + 24], data_mag=1)
    kd.flush()
    assert n_write[0] == 8 and n_read[0] == 0          # 96^3 in 64^3 cubes: 2 x 2 x 2 files
    monkeypatch.undo()
    got = KnossosDataset().initialize_from_knossos_path(str(p)).load_raw(size=np.array([96, 96, 96]), offset=np.zeros(3, int), mag=1)
    assert np.array_equal(got, want)


_JOB_SCRIPT = '''
import os, pickle, sys, time
args = []
with open(sys.argv[1], 'rb') as f:
    while True:
        try:
            args.append(pickle.load(f))
        except EOFError:
            break
t0 = time.time()
time.sleep(args[1])
with open(os.path.join(args[0], 'span_%d.txt' % args[2]), 'w') as f:
    f.write('%s %.6f %.6f' % (os.environ.get('HIP_VISIBLE_DEVICES'), t0, time.time()))
with open(sys.argv[2], 'wb') as f:
    pickle.dump(None, f)
'''


def test_batchjob_script_one_worker_per_gpu(tmp_path, monkeypatch):
    from syconn_amd import global_params
    from syconn_amd.handler.config import generate_default_conf
    from syconn_amd.mp import batchjob_utils as qu
    wd = tmp_path / 'wd'
    generate_default_conf(str(wd), scaling=(10, 10, 20), key_value_pairs=[('ngpus_per_node', 2)])
    global_params.wd = str(wd)
    scripts = tmp_path / 'scripts'
    scripts.mkdir()
    (scripts / 'batchjob_sleepy.py').write_text(_JOB_SCRIPT)
    spans = tmp_path / 'spans'
    spans.mkdir()
    monkeypatch.setattr(qu, '_visible_gpus', lambda: 2)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '3,5')
    # uneven durations: with dynamic hand-out the thread that finishes job 1 would start job 2 on GPU "3" while job 0
    # still runs there
    durs = [0.6, 0.1, 0.1, 0.1, 0.1]
    params = [(str(spans), d, i) for i, d in enumerate(durs)]
    qu.batchjob_script(params, 'sleepy', script_folder=str(scripts), additional_flags='--gres=gpu:1',
                       remove_jobfolder=True)
    rec = {}
    for i in range(len(durs)):
        dev, a, b = (spans / f'span_{i}.txt').read_text().split()
        rec[i] = (dev, float(a), float(b))
    assert [rec[i][0] for i in range(5)] == ['3', '5', '3', '5', '3']          # jobs[g::ngpu] on device g
    for dev in ('3', '5'):
        iv = sorted((a, b) for d, a, b in rec.values() if d == dev)
        for (a0, b0), (a1, b1) in zip(iv, iv[1:]):
            assert a1 >= b0 - 1e-3, f'two workers overlapped on GPU {dev}'


def test_write_combining_survives_a_region_written_twice(tmp_path):
    """a region saved twice is counted twice by the completeness counter: the cube must then be MERGED into the file, not written
    as 'complete' with zeros where nothing arrived yet"""
    p = tmp_path / 'kd'
    _make_kd(p)
    want = _vol((96, 96, 96), 5)
    first = KnossosDataset().initialize_from_knossos_path(str(p))
    first.save_raw(offset=np.array((0, 0, 32)), mags=[1], data=want[32:64, :64, :64], data_mag=1)      # on disk before
    kd = KnossosDataset().initialize_from_knossos_path(str(p))
    kd.enable_write_combining()
    for _ in range(2):                                                                                  # 2 x half a cube = "complete"
        kd.save_raw(offset=np.array((0, 0, 0)), mags=[1], data=want[:32, :64, :64], data_mag=1)
    kd.flush()
    got = KnossosDataset().initialize_from_knossos_path(str(p)).load_raw(size=np.array([64, 64, 64]), offset=np.zeros(3, int), mag=1)
    assert np.array_equal(got, want[:64, :64, :64])
