# dev probe: end-to-end wall clock of the file-system mode (KnossosDataset in -> KnossosDataset out) for one worker:
# predict_dense_to_kd-equivalent call of dense_predictor on a synthetic volume, with the time spent in host I/O.
import os, sys, tempfile, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.unet_ref import build_unet
from syconn_amd import global_params
from syconn_amd.handler.config import generate_default_conf
from syconn_amd.handler import prediction as P
from syconn_amd.knossos import KnossosDataset

shape_xyz = tuple(int(v) for v in os.environ.get('E2E_SHAPE', '1024,1024,256').split(','))      # x, y, z
geo = {'overlap_shape_tiles': [16, 16, 8], 'chunk_size': [480, 480, 240], 'tile_shape': [256, 256, 128], 'act_dtype': 'bf16'}
if os.environ.get('E2E_DEFAULT'):      # the reference's own chunk / tile geometry and the default storage type (f16x2)
    geo = {}
if os.environ.get('E2E_ACT'):
    geo = dict(geo, act_dtype=os.environ['E2E_ACT'])
with tempfile.TemporaryDirectory(dir='/dev/shm' if os.path.isdir('/dev/shm') else None) as tmp:
    wd, kd_path = tmp + '/wd', tmp + '/kd_raw'
    generate_default_conf(wd, scaling=(10, 10, 25), kd_seg=kd_path,
                          key_value_pairs=[('ngpus_per_node', 1), ('nnodes_total', 1), ('dense_prediction', geo)])
    os.makedirs(f'{wd}/models/myelin', exist_ok=True)
    model = build_unet('myelin', seed=0, final_scale=8.0)
    torch.jit.trace(model, torch.randn(1, 1, 8, 16, 16)).save(f'{wd}/models/myelin/model.pts')
    kd = KnossosDataset()
    kd.initialize_without_conf(kd_path, boundary=shape_xyz, scale=(10, 10, 25), experiment_name='synth', mags=[1])
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 256, shape_xyz[::-1], dtype=np.uint8)
    kd.save_raw(offset=(0, 0, 0), mags=[1], data=vol, data_mag=1, fast_resampling=True, upsample=False)
    global_params.wd = wd
    import syconn_amd.mp.batchjob_utils as bu
    # run the worker in-process so that the clock sees only the worker loop (no interpreter start / model trace load)
    calls = []
    orig = bu.batchjob_script
    def inproc(params, name, **kw):
        for prm in params:
            import cProfile, pstats
            pr = cProfile.Profile(); t = time.perf_counter(); pr.enable(); P.dense_predictor(prm); pr.disable(); calls.append(time.perf_counter() - t)
            if os.environ.get('E2E_PROFILE'):
                pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
                pstats.Stats(pr).print_callers('is_available')
        return tmp + '/out'
    bu.batchjob_script = inproc
    t0 = time.perf_counter()
    try:
        P.predict_dense_to_kd(kd_path + '/', f'{wd}/knossosdatasets/', f'{wd}/models/myelin/model.pts', n_channel=2,
                              target_names=['myelin'], target_channels=[(1,)], mag=1, overwrite=True,
                              cube_of_interest=None)
    finally:
        bu.batchjob_script = orig
    dt = time.perf_counter() - t0
    nvox = float(np.prod(shape_xyz))
    print(f'volume {shape_xyz}: total {dt:.2f} s, worker loop {sum(calls):.2f} s -> {nvox / max(sum(calls), 1e-9) / 1e6:.0f} Mvox/s end to end '
          f'(file system = {"tmpfs" if "/dev/shm" in tmp else "disk"})')
