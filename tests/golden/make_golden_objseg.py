"""Golden vectors for the first stage of probability-map -> object segmentation, produced by
the REFERENCE'S OWN code: ``apply_morphological_operations``, ``_count_subsequent_mops``, ``_multi_mop_findobjects`` and
``get_aniso_struct`` are lifted by AST from /root/reference/syconn/proc/image.py (:357-438, :485-539) and executed here
with scipy; the inline threshold / label statements of /root/reference/syconn/extraction/object_extraction_steps.py
(:316-317, :354-358) are applied around them exactly as written there.  Only inputs and outputs are stored.

    python tests/golden/make_golden_objseg.py      ->  tests/golden/g9_objseg.npz
"""
import ast
import os
import sys

import numpy as np
import scipy.ndimage
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = '/root/reference/syconn'


def lift(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return [ns[n] for n in names]


def blobs(shape, seed, sigma, fill):
    """smooth random field -> uint8 'probability' map whose threshold set has blobs, holes and specks"""
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), sigma)
    v = (v - v.min()) / (v.max() - v.min())
    v = v + 0.08 * rng.random(shape)              # salt: single-voxel specks and holes for opening / closing to act on
    return np.clip(v * 255 * fill, 0, 255).astype(np.uint8)


CASES = [
    # name, shape (x,y,z), seed, sigma, fill, threshold (uint8 scale), morph ops, scaling
    ('plain',      (24, 20, 16), 1, 1.2, 1.0, 0.428571429 * 255 + 30, [], (10, 10, 20)),
    ('plain_big',  (64, 48, 40), 11, 1.5, 1.0, 150.0, [], (10, 10, 20)),
    ('open_close', (30, 28, 14), 2, 1.5, 1.0, 135.0, ['binary_opening', 'binary_closing'], (10, 10, 20)),
    ('close_open', (26, 31, 17), 3, 1.3, 1.1, 150.0, ['binary_closing', 'binary_opening'], (10, 10, 20)),
    ('close2',     (22, 22, 12), 4, 1.5, 1.0, 125.0, ['binary_closing', 'binary_closing'], (10, 10, 20)),
    ('open2_iso',  (21, 25, 19), 5, 1.4, 1.0, 135.0, ['binary_opening', 'binary_opening', 'binary_closing'], (10, 10, 10)),
    ('dilate',     (18, 17, 9),  6, 1.5, 0.9, 140.0, ['binary_dilation'], (9, 9, 20)),
    ('touch_edge', (16, 16, 8),  7, 3.0, 1.6, 100.0, ['binary_closing', 'binary_opening'], (10, 10, 20)),
    ('empty',      (9, 8, 7),    8, 1.0, 0.2, 250.0, ['binary_opening', 'binary_closing'], (10, 10, 20)),
    ('aniso3',     (20, 20, 10), 9, 2.0, 1.0, 118.0, ['binary_closing'], (10, 10, 30)),
    ('sj_like',    (48, 40, 24), 12, 1.3, 1.0, 145.0, ['binary_closing', 'binary_opening'], (10, 10, 20)),
]


def main():
    import typing
    ns = {'np': np, 'ndimage': ndimage, 'scipy': scipy}
    ns.update({k: getattr(typing, k) for k in ('List', 'Union', 'Optional', 'Tuple')})
    apply_mops, _, _, get_struct = lift(f'{REF}/proc/image.py',
                                        ['apply_morphological_operations', '_count_subsequent_mops',
                                         '_multi_mop_findobjects', 'get_aniso_struct'], ns)
    out = {'names': np.array([c[0] for c in CASES])}
    for name, shape, seed, sigma, fill, thr, ops, scaling in CASES:
        prob = blobs(shape, seed, sigma, fill)
        scaling = np.array(scaling)
        struct = get_struct(scaling)                                          # object_extraction_steps.py:243
        tmp_data = np.array(prob > thr, dtype=np.uint8)                       # :316-317
        if len(ops):
            mop_data = apply_mops(tmp_data.copy(), ops, mop_kwargs=dict(structure=struct))          # :354-355
            labels, max_label = scipy.ndimage.label(mop_data)                 # :356
        else:
            mop_data = tmp_data
            labels, max_label = scipy.ndimage.label(tmp_data)                 # :358
        out.update({f'{name}_prob': prob, f'{name}_thr': np.float64(thr), f'{name}_ops': np.array(ops, dtype='U32'),
                    f'{name}_scaling': scaling, f'{name}_struct': np.asarray(struct).astype(np.uint8),
                    f'{name}_mask': np.asarray(mop_data).astype(np.uint8), f'{name}_labels': labels.astype(np.int32),
                    f'{name}_max_label': np.int64(max_label)})
        print(name, shape, 'foreground', int(tmp_data.sum()), '->', int(np.asarray(mop_data).sum()), 'components', max_label)
    np.savez_compressed(os.path.join(HERE, 'g9_objseg.npz'), **out)


if __name__ == '__main__':
    main()
