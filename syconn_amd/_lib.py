"""ctypes binding of ``libsyconn_dense_hip.so`` (C ABI: ``include/syconn_dense.h``).

There is deliberately no fallback: if the shared library is missing (not built) loading raises, and if no
MI355X is visible ``sd_init`` returns ``SD_ERR_NODEVICE`` which is raised as ``RuntimeError``.
"""
import ctypes as C
import os

SD_OK, SD_ERR_INVALID, SD_ERR_NOMEM, SD_ERR_HIP, SD_ERR_NODEVICE = 0, -1, -2, -3, -4
SD_U8, SD_F32, SD_BF16, SD_F16, SD_U64, SD_U32, SD_F16X2 = 0, 1, 2, 3, 4, 5, 6
SD_OUT_LOGITS_F32, SD_OUT_PROBS_F32, SD_OUT_PROBS_U8 = 0, 1, 2
SD_OP_CONV, SD_OP_POOL, SD_OP_UPCONV, SD_OP_GROUPNORM, SD_OP_FINAL = 1, 2, 3, 4, 5
SD_MOP_OPENING, SD_MOP_CLOSING, SD_MOP_DILATION, SD_MOP_EROSION = 1, 2, 3, 4

LIB_NAME = 'libsyconn_dense_hip.so'
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get('SD_LIB_NAME', LIB_NAME))

# every symbol include/syconn_dense.h declares (checked by tests/test_abi.py)
EXPORTS = ['sd_init', 'sd_device_count', 'sd_model_create', 'sd_model_destroy', 'sd_workspace_bytes', 'sd_model_overflow', 'sd_forward', 'sd_forward_batch', 'sd_forward_labels_batch',
           'sd_tile_gather', 'sd_tile_scatter', 'sd_postproc_labels', 'sd_profile_enable', 'sd_profile_read',
           'sd_debug_read_buffer', 'sd_model_num_ops', 'sd_debug_last_launch_count', 'sd_debug_op_kernel', 'sd_last_error', 'sd_version', 'sd_snappy_max_compressed_length',
           'sd_snappy_compress', 'sd_snappy_uncompressed_length', 'sd_snappy_uncompress', 'sd_downsample2', 'sd_box_majority',
           'sd_objtable_bytes', 'sd_pairtable_bytes', 'sd_segstats_scan', 'sd_segstats_compact_objects',
           'sd_segstats_compact_pairs', 'sd_objseg_workspace_bytes', 'sd_object_segmentation', 'sd_objseg_watershed_workspace_bytes',
           'sd_object_segmentation_watershed', 'sd_marker_flood', 'sd_host_box_copy', 'sd_host_zero', 'sd_plan_clip_window',
           'sd_gauss_workspace_bytes', 'sd_gaussian_threshold', 'sd_model_set_roi', 'sd_labels_make_unique', 'sd_labels_box_lut',
           'sd_chunkprops_append', 'sd_chunkpairs_append', 'sd_propmerge_temp_bytes', 'sd_propmerge_objects', 'sd_propmerge_pairs', 'sd_profile_read_clocks', 'sd_probe_mfma_rate', 'sd_memcpy2d_async']


class OpDesc(C.Structure):
    """Mirror of ``sd_op_desc``."""
    _fields_ = [('kind', C.c_int32), ('src0', C.c_int32), ('src1', C.c_int32), ('dst', C.c_int32),
                ('cin0', C.c_int32), ('cin1', C.c_int32), ('cout', C.c_int32),
                ('kz', C.c_int32), ('ky', C.c_int32), ('kx', C.c_int32),
                ('relu', C.c_int32), ('norm', C.c_int32), ('groups', C.c_int32), ('eps', C.c_float),
                ('w_off', C.c_int64), ('b_off', C.c_int64), ('gamma_off', C.c_int64), ('beta_off', C.c_int64),
                ('mean_off', C.c_int64), ('var_off', C.c_int64)]


_lib = None


def load():
    """Load the library once; raise if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not found: build it with `make -C syconn_amd/csrc` '
                          f'(or __graft_entry__.build()); syconn_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.sd_init.argtypes = [i32]; lib.sd_init.restype = i32
    lib.sd_device_count.argtypes = []; lib.sd_device_count.restype = i32
    lib.sd_model_create.argtypes = [C.POINTER(OpDesc), i32, C.POINTER(C.c_float), sz, i32, C.POINTER(vp)]
    lib.sd_model_create.restype = i32
    lib.sd_model_destroy.argtypes = [vp]; lib.sd_model_destroy.restype = None
    lib.sd_workspace_bytes.argtypes = [vp, i32, i32, i32]; lib.sd_workspace_bytes.restype = sz
    lib.sd_model_overflow.argtypes = [vp, vp, C.POINTER(C.c_int)]; lib.sd_model_overflow.restype = i32
    lib.sd_forward.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]; lib.sd_forward.restype = i32
    lib.sd_forward_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    lib.sd_forward_batch.restype = i32
    lib.sd_forward_labels_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_double), i32,
                                            vp, vp, sz, vp]
    lib.sd_forward_labels_batch.restype = i32
    lib.sd_tile_gather.argtypes = [vp, i32] + [i32] * 6 + [vp] + [i32] * 3 + [vp]; lib.sd_tile_gather.restype = i32
    lib.sd_tile_scatter.argtypes = [vp, i32] + [i32] * 10 + [vp] + [i32] * 6 + [vp]; lib.sd_tile_scatter.restype = i32
    lib.sd_postproc_labels.argtypes = [vp, i32, sz, C.POINTER(C.c_int32), C.POINTER(C.c_double), i32, vp, i32, vp]
    lib.sd_postproc_labels.restype = i32
    lib.sd_profile_enable.argtypes = [vp, i32]; lib.sd_profile_enable.restype = i32
    lib.sd_profile_read.argtypes = [vp, i32, C.POINTER(C.c_float), i32]; lib.sd_profile_read.restype = i32
    lib.sd_profile_read_clocks.argtypes = [vp, i32, C.POINTER(C.c_uint64), i32]; lib.sd_profile_read_clocks.restype = i32
    lib.sd_probe_mfma_rate.argtypes = [i32, i32, i32, C.c_double, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    lib.sd_probe_mfma_rate.restype = i32
    lib.sd_memcpy2d_async.argtypes = [vp, sz, vp, sz, sz, sz, i32, vp]; lib.sd_memcpy2d_async.restype = i32
    lib.sd_debug_read_buffer.argtypes = [vp, i32, vp, vp, C.POINTER(C.c_int32), vp]
    lib.sd_debug_read_buffer.restype = i32
    lib.sd_model_num_ops.argtypes = [vp]; lib.sd_model_num_ops.restype = i32
    lib.sd_debug_last_launch_count.argtypes = [vp]; lib.sd_debug_last_launch_count.restype = i32
    lib.sd_debug_op_kernel.argtypes = [vp, i32, C.c_char_p, i32]; lib.sd_debug_op_kernel.restype = i32
    lib.sd_last_error.argtypes = []; lib.sd_last_error.restype = C.c_char_p
    lib.sd_version.argtypes = []; lib.sd_version.restype = C.c_char_p
    lib.sd_snappy_max_compressed_length.argtypes = [sz]; lib.sd_snappy_max_compressed_length.restype = sz
    lib.sd_snappy_compress.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]; lib.sd_snappy_compress.restype = i32
    lib.sd_snappy_uncompressed_length.argtypes = [vp, sz, C.POINTER(sz)]; lib.sd_snappy_uncompressed_length.restype = i32
    lib.sd_snappy_uncompress.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]; lib.sd_snappy_uncompress.restype = i32
    lib.sd_downsample2.argtypes = [vp, i32, i32, i32, i32, vp, vp]; lib.sd_downsample2.restype = i32
    lib.sd_box_majority.argtypes = [vp, i32, i32, i32, vp, sz, i32, i32, i32, C.c_double, C.c_double, vp, vp]
    lib.sd_box_majority.restype = i32
    lib.sd_objtable_bytes.argtypes = [sz]; lib.sd_objtable_bytes.restype = sz
    lib.sd_pairtable_bytes.argtypes = [sz]; lib.sd_pairtable_bytes.restype = sz
    lib.sd_segstats_scan.argtypes = [vp, C.POINTER(vp), i32, i32, i32, i32, i32, vp, C.POINTER(vp), sz, C.POINTER(vp), sz,
                                     i32, vp, vp]
    lib.sd_segstats_scan.restype = i32
    lib.sd_segstats_compact_objects.argtypes = [vp, sz, vp, vp, vp, vp, sz, vp, vp]
    lib.sd_segstats_compact_objects.restype = i32
    lib.sd_segstats_compact_pairs.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp, sz, vp, vp]
    lib.sd_segstats_compact_pairs.restype = i32
    lib.sd_objseg_workspace_bytes.argtypes = [i32, i32, i32, i32]; lib.sd_objseg_workspace_bytes.restype = sz
    lib.sd_object_segmentation.argtypes = [vp, i32, i32, i32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32,
                                           vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.sd_object_segmentation.restype = i32
    lib.sd_objseg_watershed_workspace_bytes.argtypes = [i32, i32, i32, i32]; lib.sd_objseg_watershed_workspace_bytes.restype = sz
    lib.sd_object_segmentation_watershed.argtypes = [vp, i32, i32, i32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32,
                                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32, vp, i32, i32, i32, i32,
                                                     C.POINTER(C.c_int32), vp, vp, vp, vp, vp, vp, sz, vp]
    lib.sd_object_segmentation_watershed.restype = i32
    lib.sd_marker_flood.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]; lib.sd_marker_flood.restype = i32
    lib.sd_model_set_roi.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]; lib.sd_model_set_roi.restype = i32
    lib.sd_gauss_workspace_bytes.argtypes = [i32, i32, i32]; lib.sd_gauss_workspace_bytes.restype = sz
    lib.sd_gaussian_threshold.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_double), C.c_double, vp, vp, vp, sz, vp]
    lib.sd_gaussian_threshold.restype = i32
    lib.sd_labels_make_unique.argtypes = [vp, sz, C.c_uint64, vp, vp]; lib.sd_labels_make_unique.restype = i32
    lib.sd_labels_box_lut.argtypes = [vp] + [i32] * 9 + [vp, sz, vp, vp, vp]; lib.sd_labels_box_lut.restype = i32
    lib.sd_chunkprops_append.argtypes = [vp, sz] + [i32] * 6 + [C.c_uint64, vp, vp, vp, vp, sz, vp, vp]
    lib.sd_chunkprops_append.restype = i32
    lib.sd_chunkpairs_append.argtypes = [vp, sz, vp, vp, sz] + [i32] * 3 + [C.c_uint64, vp, vp, vp, sz, vp, vp]
    lib.sd_chunkpairs_append.restype = i32
    lib.sd_propmerge_temp_bytes.argtypes = [sz]; lib.sd_propmerge_temp_bytes.restype = sz
    lib.sd_propmerge_objects.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp]; lib.sd_propmerge_objects.restype = i32
    lib.sd_propmerge_pairs.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp, sz, vp]; lib.sd_propmerge_pairs.restype = i32
    i64 = C.c_int64
    lib.sd_host_box_copy.argtypes = [vp, i64, i64, vp, i64, i64, i64, i64, i64, i32]; lib.sd_host_box_copy.restype = i32
    lib.sd_host_zero.argtypes = [vp, i64, i32]; lib.sd_host_zero.restype = i32
    lib.sd_plan_clip_window.argtypes = [C.POINTER(OpDesc), i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.sd_plan_clip_window.restype = i32
    _lib = lib
    return lib


class ActivationOverflowError(RuntimeError, FloatingPointError):
    """fp16 activation storage ('f16', 'f16x2') overflowed (> 65504) during a forward pass: its results are invalid.  Use
    act_dtype='bf16' / 'f32' (fp32's exponent range)."""


def check(rc: int, what: str = ''):
    """Map library error codes to the exceptions the reference's callers expect
    (RuntimeError == 'out of device memory' for the tile-halving loop)."""
    if rc == SD_OK:
        return
    msg = load().sd_last_error().decode(errors='replace')
    text = f'{what}: {msg}' if what else msg
    if rc == SD_ERR_INVALID:
        raise ValueError(text)
    raise RuntimeError(text)


def host_box_copy(dst, src, n_threads: int = 16):
    """dst[...] = src for two 3D uint8 CPU tensors / arrays of equal shape whose last axis is contiguous (views into larger
    volumes are the point), copied row by row on `n_threads` host threads by the C helper."""
    import torch
    d = dst if isinstance(dst, torch.Tensor) else torch.from_numpy(dst)
    s = src if isinstance(src, torch.Tensor) else torch.from_numpy(src)
    assert d.dtype == torch.uint8 and s.dtype == torch.uint8 and d.dim() == 3 and tuple(d.shape) == tuple(s.shape)
    assert (d.stride(2) == 1 or d.shape[2] <= 1) and (s.stride(2) == 1 or s.shape[2] <= 1)
    rc = load().sd_host_box_copy(s.data_ptr(), s.stride(0), s.stride(1), d.data_ptr(), d.stride(0), d.stride(1),
                                 d.shape[0], d.shape[1], d.shape[2], int(n_threads))
    if rc != SD_OK:
        raise ValueError('sd_host_box_copy: bad argument')


def host_zero(t, n_threads: int = 16):
    assert t.is_contiguous() and t.dtype.itemsize == 1
    load().sd_host_zero(t.data_ptr(), t.numel(), int(n_threads))


# -- snappy raw format (host side; the codec of the KNOSSOS ``*.seg.sz.zip`` overlay cubes) ---------------------------
def snappy_compress(data) -> bytes:
    """``snappy.compress(data)`` of python-snappy (raw format), computed by the C codec of the library."""
    lib = load()
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    n = len(buf)
    cap = lib.sd_snappy_max_compressed_length(n)
    out = C.create_string_buffer(cap)
    out_len = C.c_size_t(0)
    src = (C.c_char * n).from_buffer_copy(buf) if n else None
    rc = lib.sd_snappy_compress(src, n, out, cap, C.byref(out_len))
    if rc != SD_OK:
        raise ValueError('snappy_compress: invalid argument')
    return out.raw[:out_len.value]


def snappy_decompress(data) -> bytes:
    """``snappy.decompress(data)``; raises ValueError on a corrupt or truncated stream."""
    lib = load()
    buf = bytes(data)
    n = len(buf)
    src = (C.c_char * n).from_buffer_copy(buf) if n else None
    ulen = C.c_size_t(0)
    if lib.sd_snappy_uncompressed_length(src, n, C.byref(ulen)) != SD_OK:
        raise ValueError('snappy_decompress: corrupt input (length preamble)')
    out = C.create_string_buffer(max(ulen.value, 1))
    out_len = C.c_size_t(0)
    if lib.sd_snappy_uncompress(src, n, out, ulen.value, C.byref(out_len)) != SD_OK:
        raise ValueError('snappy_decompress: corrupt input')
    return out.raw[:out_len.value]
