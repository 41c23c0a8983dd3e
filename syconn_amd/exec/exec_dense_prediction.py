"""The five per-target wrappers of /root/reference/syconn/exec/exec_dense_prediction.py:12-150: parameter binding
only (model path, `n_channel`, `mag`, `target_channels`, `target_names`) onto ``predict_dense_to_kd``."""
from typing import Optional, Tuple

import numpy as np

from .. import global_params
from ..handler.prediction import predict_dense_to_kd


def predict_myelin(kd_raw_path: str = None, cube_of_interest: Optional[Tuple[np.ndarray]] = None):
    """Myelin probability map (uint8 raw channel) at ``<wd>/knossosdatasets/myelin/``, predicted at mag 4 with the
    2-class model (exec_dense_prediction.py:12-54)."""
    if kd_raw_path is None:
        kd_raw_path = global_params.config.kd_seg_path
    predict_dense_to_kd(kd_raw_path, global_params.config.working_dir + '/knossosdatasets/',
                        global_params.config.mpath_myelin, n_channel=2, mag=4, target_channels=[(1,)],
                        target_names=['myelin'], cube_of_interest=cube_of_interest)


def predict_synapsetype(cube_of_interest: Optional[Tuple[np.ndarray]] = None):
    """Synapse type labels (1: asymmetric, 2: symmetric) as overlay at ``<wd>/knossosdatasets/syntype_v2/``
    (exec_dense_prediction.py:57-76)."""
    predict_dense_to_kd(global_params.config.kd_seg_path, global_params.config.working_dir + '/knossosdatasets/',
                        global_params.config.mpath_syntype, mag=1, n_channel=4, target_names=['syntype_v2'],
                        target_channels=[(1, 2)], cube_of_interest=cube_of_interest)


def predict_cellorganelles(cube_of_interest: Optional[Tuple[np.ndarray]] = None):
    """Labels 1: mitochondria, 2: vesicle clouds, 3: synaptic junctions as overlay at
    ``<wd>/knossosdatasets/mivcsj/`` (exec_dense_prediction.py:79-102)."""
    predict_dense_to_kd(global_params.config.kd_seg_path, global_params.config.working_dir + '/knossosdatasets/',
                        global_params.config.mpath_mivcsj, mag=1, n_channel=4, target_names=['mivcsj'],
                        target_channels=[(1, 2, 3)], cube_of_interest=cube_of_interest)


def predict_er(cube_of_interest: Optional[Tuple[np.ndarray]] = None):
    """ER probability map at ``<wd>/knossosdatasets/er/`` (exec_dense_prediction.py:105-126)."""
    predict_dense_to_kd(global_params.config.kd_seg_path, global_params.config.working_dir + '/knossosdatasets/',
                        global_params.config.mpath_er, mag=1, n_channel=2, target_names=['er'],
                        target_channels=[(1,)], cube_of_interest=cube_of_interest)


def predict_golgi(cube_of_interest: Optional[Tuple[np.ndarray]] = None):
    """Golgi probability map at ``<wd>/knossosdatasets/golgi/`` (exec_dense_prediction.py:129-150)."""
    predict_dense_to_kd(global_params.config.kd_seg_path, global_params.config.working_dir + '/knossosdatasets/',
                        global_params.config.mpath_golgi, mag=1, n_channel=2, target_names=['golgi'],
                        target_channels=[(1,)], cube_of_interest=cube_of_interest)
