"""Generate the golden vectors under tests/golden/ (run ONCE in the build container, where /root/reference exists;
the committed .npz files are what the tests read -- nothing here runs on the GPU box).

G1-G3 pin the oracle's numpy wrappers against the REFERENCE'S OWN functions: ``dense_predicton_helper``,
``xyz2zyx``, ``zyx2xyz`` (/root/reference/syconn/handler/prediction.py:279-307, 846-868), ``chunkify``
(/root/reference/syconn/handler/basics.py:545-561) and the label rule (prediction.py:813-833) are lifted from the
read-only reference source by AST at generation time and executed here (``import syconn`` itself fails on missing
dependencies).  Only inputs and outputs are stored -- no reference source text.

G4-G6 are produced by the torch-CPU oracle (the reference has no numeric test of the path, and elektronn3 is not
available: PARITY UNPINNED for the U-Net / tiled_apply arithmetic) and pin the oracle against drift and serve as
known answers for the HIP path.
"""
import ast
import os
import sys
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/syconn'


def lift(path, names, extra_globals=None):
    """exec the named top-level functions of a reference source file in an isolated namespace."""
    src = open(path).read()
    tree = ast.parse(src)
    ns = {'np': np}
    import typing
    ns.update({k: getattr(typing, k) for k in ('List', 'Union', 'Optional', 'Tuple', 'Iterable', 'Any')})
    ns.update(extra_globals or {})
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, 'exec'), ns)
    return [ns[n] for n in names]


def lift_label_rule(path):
    """The label rule is inline code of dense_predictor (prediction.py:813-833): lift exactly the statements of the
    ``for j in range(len(target_channels))`` loop body up to (not including) the save calls into a function."""
    src = open(path).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'dense_predictor'][0]
    chunk_loop = [n for n in fn.body if isinstance(n, ast.For) and getattr(n.target, 'id', '') == 'ch_id'][0]
    target_loop = [n for n in chunk_loop.body if isinstance(n, ast.For) and getattr(n.target, 'id', '') == 'j'][0]
    body = [n for n in target_loop.body if not isinstance(n, ast.If)]   # drop the trailing `if save_as_raw: save...`
    ret = ast.parse('return data, save_as_raw').body
    f = ast.FunctionDef(name='label_rule', args=ast.arguments(
        posonlyargs=[], args=[ast.arg('pred'), ast.arg('j'), ast.arg('target_channels'),
                              ast.arg('target_kd_path_list'), ast.arg('channel_thresholds')],
        kwonlyargs=[], kw_defaults=[], defaults=[]), body=body + ret, decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[f], type_ignores=[]))
    ns = {'np': np}
    exec(compile(mod, path, 'exec'), ns)
    return ns['label_rule']


class StubPredictor:
    """predict() -> softmax over (x, 1-x, 0.5): deterministic, C=3."""

    def predict(self, inp):
        x = torch.as_tensor(np.asarray(inp), dtype=torch.float32)
        logits = torch.cat([x, 1 - x, torch.full_like(x, 0.5)], dim=1) * 3
        return logits.softmax(1)


def main():
    rng = np.random.default_rng(0)
    helper, xyz2zyx, zyx2xyz = lift(f'{REF}/handler/prediction.py', ['dense_predicton_helper', 'xyz2zyx', 'zyx2xyz'])
    # dense_predicton_helper looks the axis helpers up as module globals
    helper.__globals__.update({'xyz2zyx': xyz2zyx, 'zyx2xyz': zyx2xyz})
    (chunkify,) = lift(f'{REF}/handler/basics.py', ['chunkify'])
    label_rule = lift_label_rule(f'{REF}/handler/prediction.py')

    # ---- G1: wrapper math -----------------------------------------------------------------------------------
    raw_xyz = rng.random((6, 7, 8)).astype(np.float32)
    g1 = {'raw_xyz': raw_xyz}
    for is_zyx in (False, True):
        for ret_zyx in (False, True):
            g1[f'out_{int(is_zyx)}{int(ret_zyx)}'] = helper(raw_xyz, StubPredictor(), is_zyx=is_zyx, return_zyx=ret_zyx)
    g1['xyz2zyx'] = np.ascontiguousarray(xyz2zyx(raw_xyz))
    g1['zyx2xyz'] = np.ascontiguousarray(zyx2xyz(raw_xyz))
    np.savez_compressed(f'{HERE}/g1_wrapper.npz', **g1)

    # ---- G2: chunkify ---------------------------------------------------------------------------------------
    g2 = {}
    for n_items, n in [(75, 8), (3, 8), (8, 8), (10, 3), (0, 4), (1, 1)]:
        parts = chunkify(list(range(n_items)), n)
        g2[f'n{n_items}_k{n}_len'] = np.array([len(p) for p in parts], dtype=np.int64)
        g2[f'n{n_items}_k{n}_flat'] = np.array([v for p in parts for v in p], dtype=np.int64)
    np.savez_compressed(f'{HERE}/g2_chunkify.npz', **g2)

    # ---- G3: label rule known answers ----------------------------------------------------------------------
    pred = rng.integers(0, 256, (4, 5, 6, 7), dtype=np.uint8)
    pred[1, 0, 0, :4] = [126, 127, 128, 129]        # around the default threshold 127.5
    pred[2, 0, 0, :4] = [200, 10, 200, 10]
    pred[3, 0, 0, :4] = [0, 0, 255, 255]            # later id overwrites earlier
    g3 = {'pred': pred}
    cases = {
        'multi_default': ([(1, 2, 3)], [None, None, None, None]),
        'multi_frac': ([(1, 2)], [None, 0.3, 0.9, None]),
        'multi_abs': ([(1, 3)], [None, 100, None, 254.999]),
        'multi_order': ([(3, 1)], [None, None, None, None]),
        'single_raw': ([(1,)], [None, None, None, None]),
        'single_thresh_ignored': ([(2,)], [None, None, 0.5, None]),
        'two_targets': ([(1,), (2, 3)], [None, None, 0.2, 0.2]),
    }
    for name, (tc, thr) in cases.items():
        for j in range(len(tc)):
            data, raw_flag = label_rule(pred, j, tc, ['p%d' % k for k in range(len(tc))], thr)
            g3[f'{name}_{j}_data'] = np.asarray(data)
            g3[f'{name}_{j}_raw'] = np.array(int(raw_flag))
    np.savez_compressed(f'{HERE}/g3_label_rule.npz', **g3)
    import json
    json.dump({k: {'target_channels': [list(t) for t in v[0]], 'thresholds': v[1]} for k, v in cases.items()},
              open(f'{HERE}/g3_label_rule_cases.json', 'w'), indent=1)

    # ---- G4: U-Net logits (oracle) -----------------------------------------------------------------------------
    from oracle.unet_ref import ARCHS, build_cnn3, build_unet
    g4 = {}
    for arch in ARCHS:
        for tag, shape in (('even', (8, 24, 24)), ('odd', (7, 19, 21))):
            x = torch.from_numpy(rng.integers(0, 256, shape, dtype=np.uint8))
            model = build_unet(arch, seed=100)
            with torch.no_grad():
                y = model((x.float() / 255.)[None, None])[0]
            g4[f'{arch}_{tag}_in'] = x.numpy()
            g4[f'{arch}_{tag}_logits'] = y.numpy()
    np.savez_compressed(f'{HERE}/g4_unet_logits.npz', **g4)

    # ---- G5: tiled_apply -----------------------------------------------------------------------------------
    from oracle.predictor_ref import PredictorRef, dense_predicton_helper_ref
    vol = rng.random((1, 1, 24, 40, 40)).astype(np.float32)
    g5 = {'vol': vol}
    ident = PredictorRef(None, tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6), out_shape=(1, 24, 40, 40),
                         strict_shapes=True, apply_softmax=False, forward=lambda t: t)
    g5['identity'] = ident.predict(vol).numpy()

    def box_sum(t):   # 3x3x3 box filter with zero padding: exposes tile-border zero padding iff overlap < 1
        return torch.nn.functional.conv3d(t, torch.ones(1, 1, 3, 3, 3), padding=1)
    for name, ol in (('box_ol1', (1, 1, 1)), ('box_ol0', (0, 0, 0))):
        p = PredictorRef(None, tile_shape=(12, 20, 20), overlap_shape=ol, out_shape=(1, 24, 40, 40),
                         strict_shapes=True, apply_softmax=False, forward=box_sum)
        g5[name] = p.predict(np.ones_like(vol)).numpy()
    small = build_unet('myelin', seed=101, n_blocks=3, start_filts=8)
    p = PredictorRef(small, tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6), out_shape=(2, 24, 40, 40),
                     strict_shapes=True, apply_softmax=True)
    g5['unet_tiled_probs'] = p.predict(vol).numpy()
    np.savez_compressed(f'{HERE}/g5_tiled_apply.npz', **g5)

    # ---- G6: BASELINE config 1 end to end (64^3, 3-layer CNN) ----------------------------------------------
    vol64 = np.random.default_rng(0).integers(0, 256, (64, 64, 64), dtype=np.uint8)
    cnn = build_cnn3(0)
    p = PredictorRef(cnn, tile_shape=(32, 32, 32), overlap_shape=(8, 8, 8), out_shape=(2, 64, 64, 64),
                     strict_shapes=True, apply_softmax=True)
    out = dense_predicton_helper_ref(vol64.astype(np.float32) / 255., p, is_zyx=True, return_zyx=True)
    import hashlib
    np.savez_compressed(f'{HERE}/g6_config1.npz', out_u8=out,
                        sha256=np.frombuffer(hashlib.sha256(out.tobytes()).digest(), dtype=np.uint8))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
