"""``chunkify`` and ``kd_factory`` of /root/reference/syconn/handler/basics.py (the two functions the dense
path uses)."""
import glob
import os
from typing import List, Union

import numpy as np

from ..knossos import KnossosDataset


def chunkify(lst: Union[list, np.ndarray], n: int) -> List[list]:
    """Split `lst` into ``min(n, len(lst))`` round-robin sub-lists ``lst[i::n]`` (basics.py:545-561).
    This is the static chunk -> GPU partition of the dense path (prediction.py:708-709)."""
    if len(lst) < n:
        n = len(lst)
    return [lst[i::n] for i in range(n)]


def kd_factory(kd_path: str, channel: str = 'jpg') -> KnossosDataset:
    """Open a KnossosDataset from a conf file, a ``*.pyk.conf`` or a ``mag1/knossos.conf`` (basics.py:33-68)."""
    kd = KnossosDataset()
    if os.path.isfile(kd_path):
        kd.initialize_from_conf(kd_path)
    elif len(glob.glob(f'{kd_path}/*.pyk.conf')) == 1:
        kd.initialize_from_pyknossos_path(glob.glob(f'{kd_path}/*.pyk.conf')[0])
    elif os.path.isfile(kd_path + '/mag1/knossos.conf'):
        kd.initialize_from_knossos_path(kd_path + '/mag1/knossos.conf')
    else:
        raise ValueError(f'Could not find KnossosDataset config at {kd_path}.')
    return kd
