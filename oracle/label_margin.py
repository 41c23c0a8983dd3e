"""ORACLE-SIDE CHECKER (test infrastructure, NOT product code): compares labels of the HIP path with the fp32 oracle
and splits every disagreement into margin-SAFE (the oracle's decision margin exceeds the STATED numeric tolerance of
the storage type: must never happen) and margin-UNSAFE (the oracle itself is closer to the decision boundary than
that tolerance: reported, never silently passed).  SURVEY.md section 7 "bit-exact labels under bf16/fp16".

Only ``tests/``, ``__graft_entry__.smoke()``, ``tools/`` probes and ``bench.py``'s ``cpu_baseline`` leg may import this.

The tolerance is A PRIORI: ``tol = TOL_LOGIT_REL[act] * max|oracle logit|`` with the constants below, fixed per
activation storage type BEFORE anything is measured.  (Round 2 derived the safe set from the error measured on the
very tensors compared, which made "no safe mismatch" true by construction; the measured error is still reported and
must itself stay below the stated tolerance -- if a kernel ever exceeds it, safe mismatches become possible and the
assertion has content.)

Two label definitions are checked:

* the reference's THRESHOLD rule on ``uint8(floor(255*softmax))`` (/root/reference/syconn/handler/prediction.py:813-833,
  864-865): ids applied in order, ``data[pred[l] > t] = l``.  With every logit off by at most ``tol`` the probability of
  class i lies in ``[f(p_i, -tol), f(p_i, +tol)]``, ``f(p, t) = p e^t / (p e^t + (1 - p) e^-t)`` (worst case: class i
  moves by +t, all others by -t).  A voxel is safe iff for every id that interval (widened by one float32 ulp of 255 for
  the truncating cast) does not contain the integer cut;
* ARGMAX over the class logits (the north star's additional demand): safe iff the oracle's top-2 logit margin exceeds
  ``2 * tol``.
"""
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .predictor_ref import label_rule_ref

# stated tolerance of a full-size network per activation storage type: max |logit error| / max |oracle logit|
# (fp32 accumulation everywhere; 'f32' = the fp32 FMA plan, which differs from torch-CPU by summation order only; 'f16x2' = the
# split-fp16 reference-precision plan: values and weights carry 22+ mantissa bits, products lose a 2^-22 cross term)
TOL_LOGIT_REL = {'bf16': 1e-2, 'f16': 1.3e-3, 'f16x2': 1e-5, 'f32': 2e-5}
# The constants are those of the networks they were stated on (round 2): the 4-level BatchNorm U-Nets of BASELINE configs[1..3], in
# which a value passes R0 = 17 stored (= rounded) activation tensors on its longest way from input to logits: 2 convolutions per
# encoder level (8) + up-convolution and 2 convolutions per decoder level (9); BatchNorm is folded into the weights and the final
# 1x1x1 layer reads the last of them.  Every stored tensor is rounded once to the storage type (relative error <= u, independent from
# tensor to tensor), everything between two stores is fp32: the roundings add up like a random walk, so the tolerance of ANY
# architecture is stated as
#       tol(arch, act) = TOL_LOGIT_REL[act] * sqrt(R(arch) / R0)
# with R(arch) counted from the architecture alone (`stored_roundings`): a network with `n` levels has 2n + 3(n - 1) such layers, and
# a GroupNorm network rounds TWICE per layer (the raw convolution output is stored for the statistics, the normalised tensor is what
# the next layer reads).  mivcsj (BASELINE configs[4]: n = 5, 'group8'): R = 2 * 22 = 44 -> fp16 tolerance 1.3e-3 * sqrt(44 / 17) =
# 2.09e-3, fixed here from the layer count before anything is measured.  The reference-precision plans ('f16x2', 'f32') keep their
# constants for every architecture: their error is summation order, not storage.
R0_STORED_ROUNDINGS = 17


def stored_roundings(arch: str) -> int:
    """Stored activation tensors a value passes on the longest path input -> logits, from the architecture alone."""
    from .unet_ref import ARCHS
    a = ARCHS[arch]
    n = int(a['n_blocks'])
    layers = 2 * n + 3 * (n - 1)
    return layers * (2 if str(a.get('normalization', 'batch')).startswith('group') else 1)


def stated_tolerance(arch: str, act: str) -> float:
    if act in ('f16x2', 'f32'):
        return TOL_LOGIT_REL[act]
    return TOL_LOGIT_REL[act] * float(np.sqrt(stored_roundings(arch) / R0_STORED_ROUNDINGS))


def _cut(t: Optional[float]) -> float:
    """prediction.py:824-827: None -> 255/2; t < 1 -> 255*t.  (uint8 p > t) <=> p >= floor(t) + 1."""
    if t is None:
        t = 255 / 2
    if t < 1.:
        t = 255 * t
    return float(np.floor(t) + 1.0)


def _shift_prob(p: torch.Tensor, t: float) -> torch.Tensor:
    """Largest (t > 0) / smallest (t < 0) softmax probability reachable when every logit moves by at most |t|."""
    p = p.double()
    a, b = p * np.exp(t), (1.0 - p) * np.exp(-t)
    return a / (a + b)


@torch.no_grad()
def label_split(ref_logits: torch.Tensor, gpu_logits: torch.Tensor, gpu_probs: torch.Tensor, gpu_labels: torch.Tensor,
                ids: Sequence[int], channel_thresholds: Sequence[Optional[float]],
                tol_logit_rel: float) -> Dict[str, float]:
    """ref_logits / gpu_logits / gpu_probs: (C, ...) float32 CPU tensors; gpu_labels: (...) uint8 labels of the HIP path
    for the multi-id target `ids` with per-CHANNEL thresholds `channel_thresholds` (reference semantics);
    tol_logit_rel: the stated a-priori tolerance (TOL_LOGIT_REL[act])."""
    ref_logits, gpu_logits, gpu_probs = ref_logits.float(), gpu_logits.float(), gpu_probs.float()
    n = int(gpu_labels.numel())
    d = (gpu_logits - ref_logits).abs()
    err_l = float(d.max())
    rms_l = float(d.pow(2).mean().sqrt())
    scale = float(ref_logits.abs().max())
    tol_l = tol_logit_rel * scale                                        # a priori: oracle quantities and a constant only

    top2 = ref_logits.topk(2, dim=0).values
    margin = top2[0] - top2[1]
    safe_a = margin > 2.0 * tol_l
    mis_a = gpu_logits.argmax(0) != ref_logits.argmax(0)

    ref_p = ref_logits.softmax(0)
    ref_u8 = (ref_p.numpy() * 255).astype(np.uint8)                      # prediction.py:864-865
    ref_lab = torch.from_numpy(label_rule_ref(ref_u8, list(ids), list(channel_thresholds))[0].astype(np.uint8))
    ulp = 255.0 * 2.0 ** -22
    safe_t = torch.ones_like(margin, dtype=torch.bool)
    for i in ids:
        cut = _cut(channel_thresholds[i])
        lo, hi = _shift_prob(ref_p[i], -tol_l) * 255.0 - ulp, _shift_prob(ref_p[i], tol_l) * 255.0 + ulp
        safe_t &= (lo >= cut) | (hi < cut)
    mis_t = gpu_labels.cpu().to(torch.uint8) != ref_lab
    # the same margins measured A POSTERIORI with twice the error this comparison observed (reporting only: how much of the a-priori
    # unsafe set is really at risk -- by construction no mismatch can lie outside it)
    tol_m = 2.0 * err_l
    safe_a_m = margin > 2.0 * tol_m
    safe_t_m = torch.ones_like(margin, dtype=torch.bool)
    for i in ids:
        cut = _cut(channel_thresholds[i])
        lo, hi = _shift_prob(ref_p[i], -tol_m) * 255.0 - ulp, _shift_prob(ref_p[i], tol_m) * 255.0 + ulp
        safe_t_m &= (lo >= cut) | (hi < cut)
    return {
        'label_unsafe_frac_2x_measured_err': float((~safe_t_m).float().mean()),
        'argmax_unsafe_frac_2x_measured_err': float((~safe_a_m).float().mean()),
        'voxels': n,
        'tol_logit_rel_stated': float(tol_logit_rel),
        'logit_err_max': err_l, 'logit_err_rms': rms_l, 'logit_err_max_rel': err_l / max(scale, 1e-30),
        'prob255_err_max': float((gpu_probs - ref_p).abs().max()) * 255.0,
        'median_top2_margin_over_tol': float(margin.median()) / max(tol_l, 1e-30),
        'median_top2_margin_over_err': float(margin.median()) / max(err_l, 1e-30),
        'argmax_mismatch_safe': int((mis_a & safe_a).sum()), 'argmax_mismatch_unsafe': int((mis_a & ~safe_a).sum()),
        'argmax_unsafe_frac': float((~safe_a).float().mean()),
        'label_mismatch_safe': int((mis_t & safe_t).sum()), 'label_mismatch_unsafe': int((mis_t & ~safe_t).sum()),
        'label_unsafe_frac': float((~safe_t).float().mean()),
        'label_agreement': 1.0 - float(mis_t.float().mean()),
        'argmax_agreement': 1.0 - float(mis_a.float().mean()),
    }


def merge_splits(parts: Sequence[Dict[str, float]]) -> Dict[str, float]:
    """Combine per-tile results: counts add, error bounds take the max, fractions are voxel-weighted."""
    n = sum(p['voxels'] for p in parts)
    out = {'voxels': n, 'tol_logit_rel_stated': parts[0]['tol_logit_rel_stated']}
    for k in ('logit_err_max', 'logit_err_max_rel', 'prob255_err_max'):
        out[k] = max(p[k] for p in parts)
    out['logit_err_rms'] = float(np.sqrt(sum(p['logit_err_rms'] ** 2 * p['voxels'] for p in parts) / n))
    for k in ('median_top2_margin_over_tol', 'median_top2_margin_over_err'):
        out[k] = float(np.median([p[k] for p in parts]))
    for k in ('argmax_mismatch_safe', 'argmax_mismatch_unsafe', 'label_mismatch_safe', 'label_mismatch_unsafe'):
        out[k] = int(sum(p[k] for p in parts))
    for k in ('argmax_unsafe_frac', 'label_unsafe_frac', 'label_agreement', 'argmax_agreement', 'label_unsafe_frac_2x_measured_err',
              'argmax_unsafe_frac_2x_measured_err'):
        out[k] = float(sum(p[k] * p['voxels'] for p in parts) / n)
    return out
