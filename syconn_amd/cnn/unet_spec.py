"""Hyper-parameters of the dense-prediction U-Nets and seeded random-init ``state_dict``s for them.

The reference writes its architectures down only in the training scripts, as ``elektronn3.models.unet.UNet(...)``
constructor calls:

* /root/reference/syconn/cnn/cnn_myelin.py:93-100          out=2, n_blocks=4, start_filts=32, planar_blocks=(0,2), 'batch'
* /root/reference/syconn/cnn/cnn_er.py:88-96               out=2, 4, 48, (0,2), 'batch'            (er, golgi)
* /root/reference/syconn/cnn/cnn_cellorganelles.py:69-77   out=4, 5, 48, (0,3), 'group8'           (mivcsj)
* /root/reference/syconn/cnn/cnn_synapse_type.py:83-94     out=4, 4, 28, (0,), batch_norm=True     (legacy layout)
* /root/reference/syconn/cnn/cnn_synapse_type_enhanced.py:128-137   same trunk, out=7

plus the two build-defined models of BASELINE.json configs 2 / 3 (the reference's
``semseg_spine`` / ``semseg_axon`` are 2D multi-view nets; the names are reused for 3D U-Nets on the myelin / er trunks
with 5 / 6 classes).  No trained weights exist (/root/reference/.MISSING_LARGE_BLOBS), so benchmarks and tools run on
seeded random weights produced here: a plain ``dict`` of tensors with elektronn3's parameter names, which is what
``syconn_amd.plan.plan_from_model`` consumes (and what loads into an elektronn3 / oracle ``UNet`` unchanged).
"""
from collections import OrderedDict
from typing import Dict, Tuple

import torch

ARCHS = {
    'myelin':       dict(out_channels=2, n_blocks=4, start_filts=32, planar_blocks=(0, 2), normalization='batch'),
    'er':           dict(out_channels=2, n_blocks=4, start_filts=48, planar_blocks=(0, 2), normalization='batch'),
    'golgi':        dict(out_channels=2, n_blocks=4, start_filts=48, planar_blocks=(0, 2), normalization='batch'),
    'syntype':      dict(out_channels=4, n_blocks=4, start_filts=28, planar_blocks=(0,), normalization='batch',
                         full_norm=False),
    'syntype_enh':  dict(out_channels=7, n_blocks=4, start_filts=28, planar_blocks=(0,), normalization='batch',
                         full_norm=False),
    'mivcsj':       dict(out_channels=4, n_blocks=5, start_filts=48, planar_blocks=(0, 3), normalization='group8'),
    'semseg_spine': dict(out_channels=5, n_blocks=4, start_filts=32, planar_blocks=(0, 2), normalization='batch'),
    'semseg_axon':  dict(out_channels=6, n_blocks=4, start_filts=48, planar_blocks=(0, 2), normalization='batch'),
}


def unet_param_shapes(out_channels: int, n_blocks: int, start_filts: int, planar_blocks=(), normalization='batch',
                      full_norm: bool = True, in_channels: int = 1) -> 'OrderedDict[str, Tuple[str, tuple]]':
    """name -> (role, shape) of every parameter / buffer of the U-Net, in elektronn3's naming and order.
    role: 'conv' | 'convT' | 'bias' | 'norm_w' | 'norm_b' | 'mean' | 'var' | 'count'."""
    out: 'OrderedDict[str, Tuple[str, tuple]]' = OrderedDict()

    def conv(prefix, cin, cout, k):
        out[prefix + '.weight'] = ('conv', (cout, cin, *k))
        out[prefix + '.bias'] = ('bias', (cout,))

    def norm(prefix, c):
        if normalization is None or normalization == 'none':
            return
        out[prefix + '.weight'] = ('norm_w', (c,))
        out[prefix + '.bias'] = ('norm_b', (c,))
        if normalization == 'batch':
            out[prefix + '.running_mean'] = ('mean', (c,))
            out[prefix + '.running_var'] = ('var', (c,))
            out[prefix + '.num_batches_tracked'] = ('count', ())

    outs = in_channels
    for i in range(n_blocks):
        ins = in_channels if i == 0 else outs
        outs = start_filts * (2 ** i)
        k3 = (1, 3, 3) if i in planar_blocks else (3, 3, 3)
        p = f'down_convs.{i}'
        conv(p + '.conv1', ins, outs, k3)
        conv(p + '.conv2', outs, outs, k3)
        if full_norm:
            norm(p + '.norm0', outs)
        norm(p + '.norm1', outs)
    for i in range(n_blocks - 1):
        ins, outs = outs, outs // 2
        planar = (n_blocks - 2 - i) in planar_blocks
        k2, k3 = ((1, 2, 2), (1, 3, 3)) if planar else ((2, 2, 2), (3, 3, 3))
        p = f'up_convs.{i}'
        out[p + '.upconv.weight'] = ('convT', (ins, outs, *k2))
        out[p + '.upconv.bias'] = ('bias', (outs,))
        conv(p + '.conv1', 2 * outs, outs, k3)
        conv(p + '.conv2', outs, outs, k3)
        if full_norm:
            norm(p + '.norm0', outs)
            norm(p + '.norm1', outs)
        norm(p + '.norm2', outs)
    conv('conv_final', outs, out_channels, (1, 1, 1))
    return out


def random_state_dict(arch: str, seed: int = 0, final_scale: float = 1.0, **overrides) -> Dict[str, torch.Tensor]:
    """Seeded random weights for `arch`: xavier-normal conv / transposed-conv weights and zero conv bias (elektronn3's
    ``weight_init``), non-trivial normalisation parameters (weight~U(0.5,1.5), bias~N(0,0.1), running_mean~N(0,0.1),
    running_var~U(0.5,1.5)) so that BatchNorm folding / GroupNorm affine are really exercised, a small non-zero class bias,
    and the final 1x1x1 weights scaled by `final_scale` (spreads the class logits: SURVEY.md section 8d)."""
    kw = dict(ARCHS[arch])
    kw.update(overrides)
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for name, (role, shape) in unet_param_shapes(**kw).items():
        if role in ('conv', 'convT'):
            rf = shape[2] * shape[3] * shape[4]
            std = (2.0 / ((shape[0] + shape[1]) * rf)) ** 0.5
            t = torch.randn(shape, generator=g) * std
            if name == 'conv_final.weight':
                t = t * final_scale
        elif role == 'bias':
            t = torch.randn(shape, generator=g) * 0.1 if name == 'conv_final.bias' else torch.zeros(shape)
        elif role in ('norm_w', 'var'):
            t = torch.rand(shape, generator=g) + 0.5
        elif role in ('norm_b', 'mean'):
            t = torch.randn(shape, generator=g) * 0.1
        else:
            t = torch.zeros((), dtype=torch.long)
        sd[name] = t
    return sd
