"""CPU: pin the oracle (oracle/) against the committed golden vectors (tests/golden/, made by make_golden.py).

G1-G3 were produced by the REFERENCE'S OWN functions (AST-lifted from /root/reference at generation time), so
these tests pin the oracle's numpy wrappers to the reference.  G4-G6 were produced by the oracle itself (parity
unpinned for the elektronn3 part) and guard against drift (torch version, seeding, refactors)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle.predictor_ref import (PredictorRef, chunkify_ref, dense_predicton_helper_ref, label_rule_ref, xyz2zyx_ref,
                                  zyx2xyz_ref)
from oracle.unet_ref import ARCHS, build_cnn3, build_unet

G = os.path.join(os.path.dirname(__file__), 'golden')


class StubPredictor:
    def predict(self, inp):
        x = torch.as_tensor(np.asarray(inp), dtype=torch.float32)
        return (torch.cat([x, 1 - x, torch.full_like(x, 0.5)], dim=1) * 3).softmax(1)


def test_g1_wrapper_math():
    g = np.load(f'{G}/g1_wrapper.npz')
    raw = g['raw_xyz']
    for a in (0, 1):
        for b in (0, 1):
            out = dense_predicton_helper_ref(raw, StubPredictor(), is_zyx=bool(a), return_zyx=bool(b))
            assert out.dtype == np.uint8 and np.array_equal(out, g[f'out_{a}{b}'])
    assert np.array_equal(xyz2zyx_ref(raw), g['xyz2zyx']) and np.array_equal(zyx2xyz_ref(raw), g['zyx2xyz'])


def test_g2_chunkify():
    g = np.load(f'{G}/g2_chunkify.npz')
    for n_items, n in [(75, 8), (3, 8), (8, 8), (10, 3), (0, 4), (1, 1)]:
        parts = chunkify_ref(list(range(n_items)), n)
        assert [len(p) for p in parts] == g[f'n{n_items}_k{n}_len'].tolist()
        assert [v for p in parts for v in p] == g[f'n{n_items}_k{n}_flat'].tolist()
    assert [len(p) for p in chunkify_ref(list(range(75)), 8)] == [10, 10, 10, 9, 9, 9, 9, 9]


def test_g3_label_rule():
    g = np.load(f'{G}/g3_label_rule.npz')
    cases = json.load(open(f'{G}/g3_label_rule_cases.json'))
    pred = g['pred']
    for name, c in cases.items():
        for j, ids in enumerate(c['target_channels']):
            data, raw = label_rule_ref(pred, ids, c['thresholds'])
            assert int(raw) == int(g[f'{name}_{j}_raw'])
            assert data.dtype == g[f'{name}_{j}_data'].dtype and np.array_equal(data, g[f'{name}_{j}_data'])
    # known answers at the default threshold: floor(255 p) > 127.5  <=>  value >= 128
    data, _ = label_rule_ref(pred, (1, 2, 3), [None] * 4)
    assert data[0, 0, :4].tolist() == [2, 0, 3, 3]


@pytest.mark.parametrize('arch', sorted(ARCHS))
def test_g4_unet_logits(arch):
    g = np.load(f'{G}/g4_unet_logits.npz')
    model = build_unet(arch, seed=100)
    for tag in ('even', 'odd'):
        x = torch.from_numpy(g[f'{arch}_{tag}_in'])
        with torch.no_grad():
            y = model((x.float() / 255.)[None, None])[0].numpy()
        ref = g[f'{arch}_{tag}_logits']
        assert y.shape == ref.shape
        assert np.allclose(y, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max()), np.abs(y - ref).max()


def test_g5_tiled_apply():
    g = np.load(f'{G}/g5_tiled_apply.npz')
    vol = g['vol']
    ident = PredictorRef(None, tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6), out_shape=(1, 24, 40, 40),
                         strict_shapes=True, apply_softmax=False, forward=lambda t: t)
    out = ident.predict(vol).numpy()
    assert np.array_equal(out, vol) and np.array_equal(out, g['identity'])      # pins the tile indexing

    def box_sum(t):
        return torch.nn.functional.conv3d(t, torch.ones(1, 1, 3, 3, 3), padding=1)
    full = box_sum(torch.ones(1, 1, 24, 40, 40)).numpy()
    p1 = PredictorRef(None, tile_shape=(12, 20, 20), overlap_shape=(1, 1, 1), out_shape=(1, 24, 40, 40),
                      strict_shapes=True, apply_softmax=False, forward=box_sum)
    assert np.array_equal(p1.predict(np.ones_like(vol)).numpy(), full)            # halo >= receptive field: seamless
    p0 = PredictorRef(None, tile_shape=(12, 20, 20), overlap_shape=(0, 0, 0), out_shape=(1, 24, 40, 40),
                      strict_shapes=True, apply_softmax=False, forward=box_sum)
    o0 = p0.predict(np.ones_like(vol)).numpy()
    assert np.array_equal(o0, g['box_ol0']) and o0[0, 0, 11, 10, 10] == 18 and full[0, 0, 11, 10, 10] == 27
    small = build_unet('myelin', seed=101, n_blocks=3, start_filts=8)
    p = PredictorRef(small, tile_shape=(12, 20, 20), overlap_shape=(4, 6, 6), out_shape=(2, 24, 40, 40),
                     strict_shapes=True, apply_softmax=True)
    assert np.allclose(p.predict(vol).numpy(), g['unet_tiled_probs'], atol=1e-5)
    with pytest.raises(ValueError):
        PredictorRef(small, tile_shape=(11, 20, 20), overlap_shape=(0, 0, 0), out_shape=(2, 24, 40, 40),
                     strict_shapes=True).predict(vol)


def test_g6_config1_end_to_end():
    """BASELINE.json config 1 on CPU: 64^3 uint8, 3-layer CNN, tiles 32^3 + halo 8, uint8 output."""
    g = np.load(f'{G}/g6_config1.npz')
    vol = np.random.default_rng(0).integers(0, 256, (64, 64, 64), dtype=np.uint8)
    p = PredictorRef(build_cnn3(0), tile_shape=(32, 32, 32), overlap_shape=(8, 8, 8), out_shape=(2, 64, 64, 64),
                     strict_shapes=True, apply_softmax=True)
    out = dense_predicton_helper_ref(vol.astype(np.float32) / 255., p, is_zyx=True, return_zyx=True)
    assert out.shape == (2, 64, 64, 64) and out.dtype == np.uint8
    diff = np.abs(out.astype(np.int16) - g['out_u8'].astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3     # floor() may flip on a 1-ulp softmax difference
    if diff.max() == 0:
        assert hashlib.sha256(out.tobytes()).digest() == g['sha256'].tobytes()


def test_g7_map_myelin2coords_restatement_matches_reference_outputs():
    """oracle/myelin_ref.py vs the outputs of the reference's own map_myelin2coords (lifted and executed by
    tests/golden/make_golden_myelin.py): this row of the oracle is PINNED."""
    from oracle.myelin_ref import box_majority_ref
    g = np.load(os.path.join(G, 'g7_myelin2coords.npz'))
    vol4, coords, mag = g['vol4'], g['coords'], int(g['mag'])
    assert np.array_equal(box_majority_ref(vol4, coords, mag=mag), g['default'])
    assert np.array_equal(box_majority_ref(vol4, coords, cube_edge_avg=(5, 7, 3), mag=mag), g['edge_5_7_3'])
    assert np.array_equal(box_majority_ref(vol4, coords, thresh_proba=100, thresh_majority=0.3, mag=mag),
                          g['thresh_100_maj_0p3'])
    assert np.array_equal(box_majority_ref(vol4, coords, thresh_proba=140.5, thresh_majority=0.1, mag=mag),
                          g['thresh_frac_maj_0p1'])
    assert 0 < g['default'].sum() < len(coords)          # both classes occur
