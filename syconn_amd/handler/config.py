"""Minimal ``DynConfig``: exactly the keys the dense prediction path reads ("Config / flags").

Mirrors /root/reference/syconn/handler/config.py:
* YAML ``config.yml`` in the working directory overlaid on package defaults (``__getitem__`` fallback, :201-215);
* working directory taken from env ``syconn_wd`` (set for worker processes, batchjob_utils.py:223) or from
  ``global_params.wd``, re-checked on every access (``_check_actuality``, :238-267);
* model paths ``mpath_*`` (:613-656), ``kd_seg_path`` (:294-299), ``ngpu_total`` (:799-801).
Geometry (chunk / tile / halo) is additionally exposed under ``dense_prediction`` -- the reference hard-codes it
with a "TODO: these should be config parameters" (prediction.py:671-677).
"""
import logging
import os
from typing import Dict, Any, Optional

import yaml

DEFAULTS = {
    'scaling': [1, 1, 1],
    'batch_proc_system': None,        # the SLURM machinery is out of scope: local one-process-per-GPU launcher
    'ncores_per_node': os.cpu_count() or 8,
    'ngpus_per_node': 8,              # one MI355X node
    'nnodes_total': 1,
    'log_level': 20,
    'default_log_dir': None,
    'disable_file_logging': True,
    'paths': {'kd_seg': None},
    'process_cell_organelles': ['mi', 'vc'],      # config.yml:15
    'dense_prediction': {
        'overlap_shape_tiles': [30, 31, 20],   # xyz, prediction.py:672
        'chunk_size': [482, 481, 236],         # prediction.py:674
        'tile_shape': [271, 181, 138],         # prediction.py:677
        # storage type of the activations on the device.  'f16x2' = the reference's precision (it computes in fp32,
        # prediction.py:777-779) on the matrix cores; the fast plans 'f16' / 'bf16' (~3.4x the throughput, 0.06 % / 0.45 % of
        # the threshold-rule labels differ from fp32) are an explicit choice; 'f32' = fp32 FMA arithmetic, ~30x slower
        'act_dtype': 'f16x2',
        # model tiles whose whole result lies beyond the dataset boundary are not predicted (the reference's chunk grid covers
        # up to 1.9x the dataset and it predicts all of it); False reproduces the reference's values in that overhang too
        'skip_tiles_outside_dataset': True,
        # model tiles that reach beyond the dataset boundary are predicted on the part of their window the voxels inside the
        # dataset depend on (identical values there; GroupNorm networks always use full windows)
        'clip_boundary_tiles': True,
        # fp16 range guard: a chunk whose activations overflow fp16 storage is predicted again in the plan with fp32's exponent range
        # ('f16x2' -> 'f32', 'f16' -> 'bf16'); False = only that chunk, True = the worker stays in the fallback plan for its remaining chunks
        'sticky_overflow_fallback': False,
    },
    # first consumer of the probability maps (object extraction): the reference's defaults,
    # /root/reference/syconn/handler/config.yml:108-136
    'cell_objects': {
        'probathresholds': {'mi': 0.428571429, 'sj': 0.19047619, 'vc': 0.285714286, 'er': 0.5, 'golgi': 0.5},
        'min_seed_vx': {'mi': 50, 'sj': 10, 'vc': 10, 'er': 30, 'golgi': 30},
        # size threshold applied during object extraction (config.yml:80-89); sv: all cell supervoxels are extracted
        'min_obj_vx': {'mi': 500, 'sj': 100, 'vc': 100, 'er': 100, 'golgi': 100, 'sv': 1, 'cs': 10, 'syn': 10, 'syn_ssv': 100},
        'extract_morph_op': {
            'mi': ['binary_opening', 'binary_closing', 'binary_erosion', 'binary_erosion', 'binary_erosion', 'binary_erosion'],
            'sj': ['binary_opening', 'binary_closing', 'binary_erosion'],
            'vc': ['binary_opening', 'binary_closing', 'binary_erosion'],
            'er': ['binary_dilation'] * 3 + ['binary_erosion'] * 3,
            'golgi': ['binary_dilation'] * 3 + ['binary_erosion'] * 3,
        },
    },
}


class DynConfig:
    def __init__(self, wd: Optional[str] = None):
        self._wd = None
        self._entries = {}
        self.initialized = False
        if wd is not None:
            self._load(wd)

    def _load(self, wd: str):
        self._wd = os.path.abspath(os.path.expanduser(wd))
        self._entries = {}
        p = self.path_config
        self.initialized = os.path.isfile(p)
        if self.initialized:
            with open(p) as f:
                self._entries = yaml.safe_load(f) or {}

    def _check_actuality(self):
        from .. import global_params
        new_wd = None
        env = os.environ.get('syconn_wd')
        if env and env != 'None':
            if self._wd != os.path.abspath(os.path.expanduser(env)):
                new_wd = env
        elif global_params.wd and global_params.wd != 'None' and \
                self._wd != os.path.abspath(os.path.expanduser(global_params.wd)):
            new_wd = global_params.wd
        if new_wd is not None:
            self._load(new_wd)

    @property
    def path_config(self) -> str:
        return f'{self._wd}/config.yml'

    @property
    def working_dir(self) -> str:
        self._check_actuality()
        if self._wd is None:
            raise ValueError('working directory not set (global_params.wd or env syconn_wd)')
        return self._wd

    def __getitem__(self, item: str) -> Any:
        self._check_actuality()
        if item in self._entries and self._entries[item] is not None:
            v = self._entries[item]
            if isinstance(v, dict) and isinstance(DEFAULTS.get(item), dict):
                merged = dict(DEFAULTS[item])
                merged.update({k: x for k, x in v.items() if x is not None})
                return merged
            return v
        if item in DEFAULTS:
            return DEFAULTS[item]
        raise KeyError(item)

    @property
    def kd_seg_path(self) -> str:
        return self['paths']['kd_seg']

    def __getattr__(self, name: str):
        # kd_<name>_path (config.py:300-360: kd_sym / kd_asym / kd_sj / kd_vc / kd_mi / kd_er / kd_golgi): the probability-map datasets
        if name.startswith('kd_') and name.endswith('_path') and name != 'kd_seg_path':
            key = name[:-len('_path')]
            try:
                return self['paths'][key]
            except KeyError:
                raise AttributeError(f"config['paths'] has no '{key}'") from None
        raise AttributeError(name)

    @property
    def kd_organelles_paths(self) -> Dict[str, str]:
        """config.py:362-373: probability-map KnossosDatasets of ``config['process_cell_organelles']``."""
        return {k: self['paths']['kd_{}'.format(k)] for k in self['process_cell_organelles']}

    @property
    def kd_organelle_seg_paths(self) -> Dict[str, str]:
        """config.py:375-386: where the organelle SEGMENTATION KnossosDatasets live."""
        return {k: "{}/knossosdatasets/{}_seg/".format(self.working_dir, k) for k in self['process_cell_organelles']}

    @property
    def temp_path(self) -> str:
        return "{}/tmp/".format(self.working_dir)

    @property
    def model_dir(self) -> str:
        return self.working_dir + '/models/'

    @property
    def mpath_myelin(self) -> str:
        return self.model_dir + '/myelin/model.pts'

    @property
    def mpath_syntype(self) -> str:
        return self.model_dir + '/syntype/model.pts'

    @property
    def mpath_er(self) -> str:
        return self.model_dir + '/er/model.pts'

    @property
    def mpath_golgi(self) -> str:
        return self.model_dir + '/golgi/model.pts'

    @property
    def mpath_mivcsj(self) -> str:
        return self.model_dir + '/mivcsj/model.pt'

    @property
    def ngpu_total(self) -> int:
        return self['nnodes_total'] * self['ngpus_per_node']


def generate_default_conf(working_dir: str, scaling=(1, 1, 1), kd_seg: Optional[str] = None,
                          key_value_pairs=None):
    """Write ``<wd>/config.yml`` (reference: config.py:812-931, reduced to this path's keys)."""
    os.makedirs(working_dir, exist_ok=True)
    entries = {'scaling': list(scaling), 'paths': {'kd_seg': kd_seg}}
    for k, v in (key_value_pairs or []):
        entries[k] = v
    with open(f'{working_dir}/config.yml', 'w') as f:
        yaml.safe_dump(entries, f)


def initialize_logging(log_name: str, log_dir: Optional[str] = None, overwrite: bool = True) -> logging.Logger:
    """Logger factory (reference: config.py:934-995); std logging instead of coloredlogs."""
    from .. import global_params
    logger = logging.getLogger(log_name)
    try:
        level = global_params.config['log_level']
    except (KeyError, ValueError):
        level = logging.INFO
    logger.setLevel(level)
    if not logger.handlers:
        sh = logging.StreamHandler()
        sh.setFormatter(logging.Formatter('%(asctime)s [%(name)s] %(levelname)s: %(message)s'))
        logger.addHandler(sh)
    if log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
        fp = os.path.join(log_dir, log_name + '.log')
        if overwrite and os.path.isfile(fp):
            os.remove(fp)
        if not any(isinstance(h, logging.FileHandler) and h.baseFilename == os.path.abspath(fp)
                   for h in logger.handlers):
            fh = logging.FileHandler(fp)
            fh.setFormatter(logging.Formatter('%(asctime)s [%(name)s] %(levelname)s: %(message)s'))
            logger.addHandler(fh)
    return logger
