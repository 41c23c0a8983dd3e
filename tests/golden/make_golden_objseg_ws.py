"""Golden vectors for the WATERSHED branch of probability-map -> object segmentation (SURVEY.md section 8f row 2;
/root/reference/syconn/extraction/object_extraction_steps.py:319-347), up to and including the relabelled MARKER volume -- the
part of that branch that is scipy / numpy in the reference and can therefore be produced by the reference's own code here:
``apply_morphological_operations`` & co. are lifted by AST from /root/reference/syconn/proc/image.py and the statements
:319-347 of ``_object_segmentation_thread`` are lifted as they stand (the ``if 'binary_erosion' in morph_ops[hdf5_name]:`` body
up to ``relabel_vol``), with ``relabel_vol`` (Cython, block_processing_C.pyx:161-169: "for every voxel: if its value is a key
of label_map, replace it") given as the same loop in Python.  The distance transform (vigra) and the flood (skimage) that
follow cannot be executed here (packages absent) and are NOT part of this fixture.  Only inputs and outputs are stored.

    python tests/golden/make_golden_objseg_ws.py      ->  tests/golden/g10_objseg_ws.npz
"""
import ast
import os
import sys
import textwrap

import numpy as np
import scipy.ndimage
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = '/root/reference/syconn'
from make_golden_objseg import blobs, lift          # noqa: E402


def relabel_vol(vol, label_map):
    """block_processing_C.pyx:161-169, the same loop in Python"""
    sh = vol.shape
    for x in range(sh[0]):
        for y in range(sh[1]):
            for z in range(sh[2]):
                prev = vol[x, y, z]
                if prev in label_map:
                    vol[x, y, z] = label_map[prev]


def lift_seed_statements():
    """source lines of object_extraction_steps.py from `first_erosion_ix = ...` to `relabel_vol(markers, label_m)`, dedented"""
    lines = open(f'{REF}/extraction/object_extraction_steps.py').read().split('\n')
    a = next(i for i, l in enumerate(lines) if 'first_erosion_ix = morph_ops[hdf5_name].index' in l)
    b = next(i for i, l in enumerate(lines) if 'relabel_vol(markers, label_m)' in l)
    src = textwrap.dedent('\n'.join(lines[a:b + 1]))
    ast.parse(src)
    return compile(src, 'object_extraction_steps.py:319-347', 'exec')


CASES = [
    # name, shape, seed, sigma, fill, threshold, ops (the default config's lists, config.yml:130-136), scaling, min_seed_vx
    ('sj_default', (48, 44, 24), 21, 1.6, 1.0, 120.0, ['binary_opening', 'binary_closing', 'binary_erosion'], (10, 10, 20), 10),
    ('vc_default', (40, 52, 20), 22, 1.4, 1.0, 128.0, ['binary_opening', 'binary_closing', 'binary_erosion'], (10, 10, 20), 10),
    ('mi_default', (72, 64, 36), 23, 3.2, 1.0, 105.0,
     ['binary_opening', 'binary_closing', 'binary_erosion', 'binary_erosion', 'binary_erosion', 'binary_erosion'], (10, 10, 20), 50),
    ('er_default', (40, 40, 24), 24, 2.0, 1.0, 150.0,
     ['binary_dilation'] * 3 + ['binary_erosion'] * 3, (10, 10, 20), 30),
    ('holes', (56, 48, 28), 25, 1.3, 1.0, 118.0, ['binary_erosion'], (10, 10, 20), 12),       # many small seeds: id hole filling
    ('no_filter', (30, 30, 16), 26, 1.5, 1.0, 125.0, ['binary_closing', 'binary_erosion'], (10, 10, 20), 1),
    ('iso', (36, 33, 31), 27, 1.8, 1.0, 122.0, ['binary_opening', 'binary_erosion', 'binary_erosion'], (10, 10, 10), 5),
    ('all_deleted', (24, 24, 12), 28, 1.2, 1.0, 140.0, ['binary_erosion'], (10, 10, 20), 100000),
    ('empty', (12, 10, 8), 29, 1.0, 0.2, 250.0, ['binary_opening', 'binary_erosion'], (10, 10, 20), 10),
]


def main():
    import typing
    ns = {'np': np, 'ndimage': ndimage, 'scipy': scipy}
    ns.update({k: getattr(typing, k) for k in ('List', 'Union', 'Optional', 'Tuple')})
    apply_mops, _, _, get_struct = lift(f'{REF}/proc/image.py',
                                        ['apply_morphological_operations', '_count_subsequent_mops',
                                         '_multi_mop_findobjects', 'get_aniso_struct'], ns)
    code = lift_seed_statements()
    out = {'names': np.array([c[0] for c in CASES])}
    for name, shape, seed, sigma, fill, thr, ops, scaling, min_seed in CASES:
        prob = blobs(shape, seed, sigma, fill)
        scaling = np.array(scaling)
        struct = get_struct(scaling)                                          # object_extraction_steps.py:243
        tmp_data = np.array(prob > thr, dtype=np.uint8)                       # :316-317
        env = {'np': np, 'scipy': scipy, 'apply_morphological_operations': apply_mops, 'relabel_vol': relabel_vol,
               'morph_ops': {'x': list(ops)}, 'min_seed_vx': {'x': min_seed}, 'hdf5_name': 'x', 'struct': struct,
               'tmp_data': tmp_data}
        exec(code, env)                                                       # :320-347 as written in the reference
        pre, markers = env['tmp_data'], env['markers']
        assert markers.dtype == np.uint32
        n_raw = int(scipy.ndimage.label(apply_mops(pre.copy(), ops[ops.index('binary_erosion'):],
                                                   mop_kwargs=dict(structure=struct)))[1])
        out.update({f'{name}_prob': prob, f'{name}_thr': np.float64(thr), f'{name}_ops': np.array(ops, dtype='U32'),
                    f'{name}_scaling': scaling, f'{name}_min_seed': np.int64(min_seed), f'{name}_pre_mask': pre.astype(np.uint8),
                    f'{name}_markers': markers})
        ids = np.unique(markers)
        print(name, shape, 'fg', int(tmp_data.sum()), '->', int(pre.sum()), 'seeds before filter', n_raw, 'kept', len(ids) - 1,
              'max id', int(ids.max()))
    np.savez_compressed(os.path.join(HERE, 'g10_objseg_ws.npz'), **out)


if __name__ == '__main__':
    main()
