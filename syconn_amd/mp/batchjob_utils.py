"""Job fan-out of the dense prediction path: one worker process per GPU of ONE node.

Keeps the call contract of /root/reference/syconn/mp/batchjob_utils.py (``batchjob_script`` :69-361 and its
local fallback ``batchjob_fallback`` :390-516) that ``predict_dense_to_kd`` relies on:

* every parameter tuple is serialised element-wise with ``pickle`` into ``<job_folder>/storage/job_<i>.pkl``
  (:478-480); the worker is ``python batchjob_<name>.py <in.pkl> <out.pkl>`` with env ``syconn_wd`` set (:473-477);
* a job succeeded iff ``<job_folder>/out/job_<i>.pkl`` exists; missing outputs raise ``ValueError`` (:487-495).

The SLURM machinery (sbatch / sacct polling / requeue) is out of scope.  What replaces it is MI355X-specific: the
reference's fallback does not pin devices (every local worker would land on GPU 0), here
every visible GPU gets ONE dispatcher thread that runs its share of the jobs (``jobs[g::ngpu]``) one after the
other with ``HIP_VISIBLE_DEVICES`` set to that GPU, so at most one worker process uses a GPU at any time -- also when
there are more jobs than GPUs (``nnodes_total > 1`` on one node, or fewer visible GPUs than ``ngpus_per_node``).
"""
import glob
import os
import pickle as pkl
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

from .. import global_params
from ..handler.config import initialize_logging

path_to_scripts_default = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                       'batchjob_scripts')
python_path_global = sys.executable


def batchjob_enabled() -> bool:
    """The reference probes ``squeue`` (batchjob_utils.py:37-61); this build always uses the single-node launcher."""
    return False


def _visible_gpus() -> int:
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return n


def batchjob_script(params: list, name: str, n_cores: int = 1, suffix: str = "", script_folder=None,
                    python_path=None, remove_jobfolder: bool = False, log=None, overwrite: bool = True,
                    job_folder=None, additional_flags: str = '', show_progress: bool = False, **_):
    """Run ``batchjob_<name>.py`` once per entry of `params`, one process per GPU, and block until all are done.
    Returns the path of the output folder."""
    if python_path is None:
        python_path = python_path_global
    wd = global_params.config.working_dir
    if job_folder is None:
        job_folder = f'{wd}/tmp/{name}_folder{suffix}/'
    if os.path.exists(job_folder):
        if not overwrite:
            raise FileExistsError(f'Batchjob folder already exists at "{job_folder}".')
        shutil.rmtree(job_folder, ignore_errors=True)
    job_folder = job_folder.rstrip('/')
    log_batchjob = log if log is not None else initialize_logging(name + suffix, log_dir=job_folder)
    path_to_scripts = script_folder if script_folder is not None else path_to_scripts_default
    path_to_script = f'{path_to_scripts}/batchjob_{name}.py'
    if not os.path.exists(path_to_script):
        raise FileNotFoundError(f'Specified script does not exist: {path_to_script}')
    dirs = {k: f'{job_folder}/{k}/' for k in ('storage', 'sh', 'log', 'err', 'out')}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)

    use_gpu = 'gpu' in additional_flags
    ngpu = max(1, min(global_params.config['ngpus_per_node'], _visible_gpus() or 1)) if use_gpu else 0
    n_workers = ngpu if use_gpu else max(1, min((os.cpu_count() or 1) // max(n_cores, 1), len(params)))
    n_workers = max(1, min(n_workers, len(params)))
    log_batchjob.info(f'Started batch job "{name}" with {len(params)} task(s) on {n_workers} worker(s)'
                      + (f', one at a time on each of {ngpu} GPU(s).' if use_gpu else '.'))
    start = time.time()
    repo_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    jobs = []
    for i_job in range(len(params)):
        storage = f'{dirs["storage"]}job_{i_job}.pkl'
        out = f'{dirs["out"]}job_{i_job}.pkl'
        with open(storage, 'wb') as f:
            for param in params[i_job]:
                pkl.dump(param, f)
        with open(f'{dirs["sh"]}job_{i_job}.sh', 'w') as f:   # kept for parity with the reference's job folder
            f.write(f'#!/bin/bash -l\nexport syconn_wd="{wd}"\n{python_path} {path_to_script} {storage} {out}')
        jobs.append((i_job, storage, out))

    devs = []
    if use_gpu:
        base = os.environ.get('HIP_VISIBLE_DEVICES')
        devs = [d for d in base.split(',') if d] if base else [str(k) for k in range(ngpu)]
        devs = devs[:ngpu]
        # SYCONN_AMD_WORKERS_PER_GPU=k (default 1): k dispatchers -- k concurrent worker processes -- per GPU.  One worker
        # already overlaps its own file I/O with the GPU (dense_predictor); more than one is for I/O-bound file systems and
        # for exercising the concurrent-writer path (neighbouring chunks of different workers share target cubes).
        devs = devs * max(1, int(os.environ.get('SYCONN_AMD_WORKERS_PER_GPU', '1')))

    def run(job, device=None):
        i_job, storage, out = job
        env = dict(os.environ)
        env['syconn_wd'] = wd
        env['PYTHONPATH'] = repo_root + os.pathsep + env.get('PYTHONPATH', '')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if device is not None:
            env['HIP_VISIBLE_DEVICES'] = device
        with open(f'{dirs["log"]}job_{i_job}.log', 'w') as lo, open(f'{dirs["err"]}job_{i_job}.log', 'w') as le:
            rc = subprocess.call([python_path, path_to_script, storage, out], env=env, stdout=lo, stderr=le)
        err = open(f'{dirs["err"]}job_{i_job}.log').read()
        return rc, err

    results = [None] * len(jobs)
    if use_gpu:
        # one dispatcher per GPU, each walking its own job list sequentially: a GPU never hosts two workers at once
        def dispatcher(g):
            for job in jobs[g::len(devs)]:
                results[job[0]] = run(job, devs[g])

        with ThreadPoolExecutor(max_workers=len(devs)) as ex:
            for f in [ex.submit(dispatcher, g) for g in range(len(devs))]:
                f.result()
    else:
        with ThreadPoolExecutor(max_workers=n_workers) as ex:
            results = list(ex.map(run, jobs))
    out_files = glob.glob(dirs['out'] + '*.pkl')
    if len(out_files) < len(params):
        errs = '\n'.join(f'job {i}: rc={rc}\n{err[-2000:]}' for i, (rc, err) in enumerate(results) if rc != 0)
        msg = (f'Critical errors occurred during "{name}". {len(params) - len(out_files)}/{len(params)} '
               f'worker(s) failed.\n{errs}')
        log_batchjob.error(msg)
        raise ValueError(msg)
    path_to_out = dirs['out']
    if remove_jobfolder:
        shutil.rmtree(job_folder, ignore_errors=True)
    log_batchjob.debug('Finished "{}" after {:.2f}s.'.format(name, time.time() - start))
    return path_to_out
