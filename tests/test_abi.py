"""CPU: the C-ABI library is built, loads, and exports every symbol include/syconn_dense.h declares.
No compute is called here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, 'include', 'syconn_dense.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sd_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_list_agree():
    from syconn_amd import _lib
    assert _declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from syconn_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), 'libsyconn_dense_hip.so missing: run __graft_entry__.build()'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f'{name} declared in include/syconn_dense.h but not exported'
    v = _lib.load().sd_version()
    assert b'gfx950' in v


def test_op_desc_layout_matches_header():
    from syconn_amd import _lib
    # 14 x 4-byte fields then 6 x int64 -> 56 + 48 = 104 bytes, 8-byte aligned
    assert ctypes.sizeof(_lib.OpDesc) == 104
    assert _lib.OpDesc.w_off.offset == 56


def test_fails_loudly_without_gpu():
    """No silent CPU fallback: constructing the product Predictor without a ROCm device must raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    from oracle.unet_ref import build_cnn3
    from syconn_amd.handler.prediction import Predictor
    with pytest.raises(RuntimeError):
        Predictor(build_cnn3(0))
    from syconn_amd import _lib
    lib = _lib.load()
    assert lib.sd_init(0) == _lib.SD_ERR_NODEVICE
    with pytest.raises(RuntimeError):
        _lib.check(lib.sd_init(0), 'sd_init')
    # the post-inference entry points likewise
    import numpy as np
    from syconn_amd.extraction.find_object_properties import find_object_properties
    from syconn_amd.extraction.object_extraction_steps import marker_flood, object_segmentation_first_stage
    z = np.zeros((4, 4, 4), np.uint8)
    for call in (lambda: object_segmentation_first_stage(z, 1.0, ['binary_erosion']), lambda: marker_flood(z, z, z),
                 lambda: find_object_properties(z.astype(np.uint64))):
        with pytest.raises(RuntimeError):
            call()
