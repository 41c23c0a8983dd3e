"""ORACLE (test infrastructure, NOT product code): torch-CPU fp32 restatement of the
elektronn3 3D U-Net that SyConn's dense prediction path runs.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

PARITY UNPINNED: the arithmetic of this path lives in the third-party package ``elektronn3``
(git branch ``syconn2``, frozen env ``elektronn3==0+untagged.1035.g9f88766``;
/root/reference/environment.yml:87, examples/working_env_glibc_2_27_2021_11.yml:345) which is neither
vendored in /root/reference nor installed here, and the reference holds no numeric test of the path
(/root/reference/tests/test_models_e3.py:34-36 only loads a model).  This file restates the published
``elektronn3.models.unet.UNet`` block structure from the call sites that pin its hyper-parameters:

* /root/reference/syconn/cnn/cnn_myelin.py:93-100        UNet(out=2, n_blocks=4, start_filts=32, planar=(0,2), 'batch')
* /root/reference/syconn/cnn/cnn_er.py:88-96             UNet(1, 2, 4, 48, (0,2), 'batch')
* /root/reference/syconn/cnn/cnn_cellorganelles.py:69-77 UNet(1, 4, 5, 48, (0,3), 'group8')
* /root/reference/syconn/cnn/cnn_synapse_type.py:83-94   UNet(1, 4, 4, 28, (0,), batch_norm=True)  (legacy API)
* /root/reference/syconn/cnn/cnn_synapse_type_enhanced.py:128-137   same trunk, out=7

Module / parameter names (``down_convs.i.conv1`` ... ``up_convs.i.upconv`` ... ``conv_final``) follow
elektronn3 so that a state_dict of a real SyConn ``model.pts`` loads into :class:`UNet` unchanged.
"""
from typing import Sequence, Tuple

import torch
import torch.nn as nn


def _planar(k: int, planar: bool):
    return (1, k, k) if planar else k


def get_normalization(normtype, num_channels: int) -> nn.Module:
    """elektronn3 ``get_normalization``: 'batch' -> BatchNorm3d, 'group<N>' -> GroupNorm(N, C), None -> Identity."""
    if normtype is None or normtype == 'none':
        return nn.Identity()
    if normtype.startswith('group'):
        tail = normtype[len('group'):]
        num_groups = int(tail) if tail.isdigit() else 8
        return nn.GroupNorm(num_groups=num_groups, num_channels=num_channels)
    if normtype == 'instance':
        return nn.InstanceNorm3d(num_channels)
    if normtype == 'batch':
        return nn.BatchNorm3d(num_channels)
    raise ValueError(f'Unknown normalization type "{normtype}"')


def autocrop(from_down: torch.Tensor, from_up: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """elektronn3 ``autocrop`` for 'same' convolutions: ``MaxPool(ceil_mode=True)`` makes the 2x up-conv
    output one element too large in every dim where the encoder tensor is odd; crop the decoder tensor at
    the high end by ``(u - d) % 2``."""
    ds = from_down.shape[2:]
    us = from_up.shape[2:]
    upcrop = [u - ((u - d) % 2) for d, u in zip(ds, us)]
    from_up = from_up[:, :, :upcrop[0], :upcrop[1], :upcrop[2]]
    ds = from_down.shape[2:]
    us = from_up.shape[2:]
    assert all(d >= u for d, u in zip(ds, us))
    # center-crop of the encoder tensor (only != identity for 'valid' convs, unused by SyConn)
    from_down = from_down[:, :,
                          (ds[0] - us[0]) // 2:(ds[0] + us[0]) // 2,
                          (ds[1] - us[1]) // 2:(ds[1] + us[1]) // 2,
                          (ds[2] - us[2]) // 2:(ds[2] + us[2]) // 2]
    return from_down, from_up


class DownConv(nn.Module):
    """conv3 -> norm0 -> ReLU -> conv3 -> norm1 -> ReLU -> (skip) -> MaxPool(ceil_mode=True)  (row U2)."""

    def __init__(self, in_channels, out_channels, pooling=True, planar=False,
                 normalization=None, full_norm=True):
        super().__init__()
        self.pooling = pooling
        k = _planar(3, planar)
        p = (0, 1, 1) if planar else 1
        self.conv1 = nn.Conv3d(in_channels, out_channels, k, padding=p)
        self.conv2 = nn.Conv3d(out_channels, out_channels, k, padding=p)
        self.pool = nn.MaxPool3d(_planar(2, planar), ceil_mode=True) if pooling else nn.Identity()
        self.act1 = nn.ReLU()
        self.act2 = nn.ReLU()
        self.norm0 = get_normalization(normalization, out_channels) if full_norm else nn.Identity()
        self.norm1 = get_normalization(normalization, out_channels)

    def forward(self, x):
        y = self.act1(self.norm0(self.conv1(x)))
        y = self.act2(self.norm1(self.conv2(y)))
        before_pool = y
        return self.pool(y), before_pool


class UpConv(nn.Module):
    """ConvTranspose(k=s=2) -> autocrop -> norm0 -> ReLU -> cat((up, enc), 1) -> conv3 -> norm1 -> ReLU
    -> conv3 -> norm2 -> ReLU  (row U3)."""

    def __init__(self, in_channels, out_channels, planar=False, normalization=None, full_norm=True):
        super().__init__()
        k2 = _planar(2, planar)
        k3 = _planar(3, planar)
        p = (0, 1, 1) if planar else 1
        self.upconv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=k2, stride=k2)
        self.conv1 = nn.Conv3d(2 * out_channels, out_channels, k3, padding=p)
        self.conv2 = nn.Conv3d(out_channels, out_channels, k3, padding=p)
        self.act0 = nn.ReLU()
        self.act1 = nn.ReLU()
        self.act2 = nn.ReLU()
        self.norm0 = get_normalization(normalization, out_channels) if full_norm else nn.Identity()
        self.norm1 = get_normalization(normalization, out_channels) if full_norm else nn.Identity()
        self.norm2 = get_normalization(normalization, out_channels)

    def forward(self, enc, dec):
        updec = self.upconv(dec)
        enc, updec = autocrop(enc, updec)
        updec = self.act0(self.norm0(updec))
        mrg = torch.cat((updec, enc), 1)
        y = self.act1(self.norm1(self.conv1(mrg)))
        y = self.act2(self.norm2(self.conv2(y)))
        return y


class UNet(nn.Module):
    """elektronn3-style 3D U-Net, 'same' convolutions, transposed-conv upsampling, concat merge (row U1).

    ``full_norm=False`` reproduces the legacy ``batch_norm=True`` API used by cnn_synapse_type.py:90
    (normalisation after the second conv of every block only; row U5)."""

    def __init__(self, in_channels: int = 1, out_channels: int = 2, n_blocks: int = 3, start_filts: int = 32,
                 planar_blocks: Sequence[int] = (), activation: str = 'relu', normalization='batch',
                 full_norm: bool = True):
        super().__init__()
        if activation != 'relu':
            raise ValueError('SyConn only uses activation="relu" on this path')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.n_blocks, self.start_filts = n_blocks, start_filts
        self.planar_blocks = tuple(planar_blocks)
        self.normalization, self.full_norm = normalization, full_norm
        self.down_convs = nn.ModuleList()
        self.up_convs = nn.ModuleList()
        outs = in_channels
        for i in range(n_blocks):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            self.down_convs.append(DownConv(ins, outs, pooling=i < n_blocks - 1, planar=i in self.planar_blocks,
                                            normalization=normalization, full_norm=full_norm))
        for i in range(n_blocks - 1):
            ins = outs
            outs = ins // 2
            self.up_convs.append(UpConv(ins, outs, planar=(n_blocks - 2 - i) in self.planar_blocks,
                                        normalization=normalization, full_norm=full_norm))
        self.conv_final = nn.Conv3d(outs, out_channels, 1)
        self.apply(self.weight_init)

    @staticmethod
    def weight_init(m):
        """row U6: xavier_normal_ on conv / transposed-conv weights, zero bias."""
        if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
            nn.init.xavier_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        encoder_outs = []
        for module in self.down_convs:
            x, before_pool = module(x)
            encoder_outs.append(before_pool)
        for i, module in enumerate(self.up_convs):
            x = module(encoder_outs[-(i + 2)], x)
        return self.conv_final(x)


# ----------------------------------------------------------------------------------------------------
# The architectures SyConn instantiates (hyper-parameters from the cnn_*.py call sites cited above) and
# the two build-defined ones of BASELINE.json configs 2/3.
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


def randomize_norm_stats(model: nn.Module, gen: torch.Generator) -> None:
    """Non-trivial BatchNorm running stats / affine so that BN folding is really exercised:
    weight~U(0.5,1.5), bias~N(0,0.1), running_mean~N(0,0.1), running_var~U(0.5,1.5)."""
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm3d, nn.GroupNorm)):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
                if isinstance(m, nn.BatchNorm3d):
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


def build_unet(arch: str, seed: int = 0, final_scale: float = 1.0, **overrides) -> UNet:
    """Seeded random-init U-Net of one of :data:`ARCHS` in eval mode (there are no trained weights:
    /root/reference/.MISSING_LARGE_BLOBS)."""
    kw = dict(ARCHS[arch])
    kw.update(overrides)
    torch.manual_seed(seed)
    model = UNet(in_channels=1, **kw)
    gen = torch.Generator().manual_seed(seed + 1)
    randomize_norm_stats(model, gen)
    with torch.no_grad():
        # spread the final logits (non-zero bias + scale) so that class decisions have a margin
        model.conv_final.bias.copy_(torch.randn(model.conv_final.bias.shape, generator=gen) * 0.1)
        model.conv_final.weight.mul_(final_scale)
    return model.eval()


def build_cnn3(seed: int = 0) -> nn.Sequential:
    """BASELINE.json config 1: random-init 3-layer 3D CNN."""
    torch.manual_seed(seed)
    m = nn.Sequential(nn.Conv3d(1, 8, 3, padding=1), nn.ReLU(),
                      nn.Conv3d(8, 8, 3, padding=1), nn.ReLU(),
                      nn.Conv3d(8, 2, 1))
    for mod in m:
        if isinstance(mod, nn.Conv3d):
            nn.init.xavier_normal_(mod.weight)
            nn.init.constant_(mod.bias, 0)
    return m.eval()


# ----------------------------------------------------------------------------------------------------
# Reduced-precision emulation: same graph, weights rounded to `dtype`, every stored activation rounded to
# `dtype`, fp32 accumulation.  This is what the HIP path computes (up to summation order) and gives a
# much tighter check than comparing bf16 results with the fp32 oracle.
def _round(t: torch.Tensor, dtype) -> torch.Tensor:
    return t.to(dtype).to(torch.float32)


def fold_bn_conv(conv, bn):
    """Return (W', b') of conv followed by eval-mode BatchNorm: W' = W*s, b' = b*s + (beta - mean*s)."""
    w, b = conv.weight.detach().clone(), conv.bias.detach().clone()
    if isinstance(bn, nn.BatchNorm3d):
        s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        t = bn.bias.detach() - bn.running_mean * s
        if isinstance(conv, nn.ConvTranspose3d):
            w = w * s.view(1, -1, 1, 1, 1)
        else:
            w = w * s.view(-1, 1, 1, 1, 1)
        b = b * s + t
    return w, b


@torch.no_grad()
def unet_forward_emulated(model: UNet, x: torch.Tensor, dtype=torch.bfloat16, first_conv_fp32: bool = True,
                          collect: list = None):
    """Forward of `model` with the storage precision of the HIP path: BN folded into fp32 weights which are
    then rounded to `dtype` (first conv and final 1x1x1 conv keep fp32 weights), activations rounded to
    `dtype` after every conv(+norm+ReLU) / up-conv / GroupNorm-apply, fp32 accumulation.  Returns logits.
    If `collect` is a list, every stored activation is appended in layer order (entry i <-> plan buffer i+1)."""
    import torch.nn.functional as F
    if collect is None:
        collect = []

    def conv_block(conv, norm, t, keep_fp32_w=False):
        if isinstance(norm, nn.GroupNorm):
            w, b = conv.weight, conv.bias
            y = F.conv3d(t, w if keep_fp32_w else _round(w, dtype), b, padding=conv.padding)
            y = _round(y, dtype)              # raw conv output is stored before the statistics pass
            y = F.group_norm(y, norm.num_groups, norm.weight, norm.bias, norm.eps)
        else:
            w, b = fold_bn_conv(conv, norm)
            y = F.conv3d(t, w if keep_fp32_w else _round(w, dtype), b, padding=conv.padding)
        y = _round(F.relu(y), dtype)
        collect.append(y)
        return y

    enc = []
    first = True
    for dc in model.down_convs:
        y = conv_block(dc.conv1, dc.norm0, x, keep_fp32_w=first and first_conv_fp32)
        first = False
        y = conv_block(dc.conv2, dc.norm1, y)
        enc.append(y)
        x = dc.pool(y)
        if dc.pooling:
            collect.append(x)
    for i, uc in enumerate(model.up_convs):
        e = enc[-(i + 2)]
        if isinstance(uc.norm0, nn.GroupNorm):
            up = F.conv_transpose3d(x, _round(uc.upconv.weight, dtype), uc.upconv.bias, stride=uc.upconv.stride)
            e, up = autocrop(e, up)
            up = _round(up, dtype)
            up = F.group_norm(up, uc.norm0.num_groups, uc.norm0.weight, uc.norm0.bias, uc.norm0.eps)
        else:
            w, b = fold_bn_conv(uc.upconv, uc.norm0)
            up = F.conv_transpose3d(x, _round(w, dtype), b, stride=uc.upconv.stride)
            e, up = autocrop(e, up)
        up = _round(F.relu(up), dtype)
        collect.append(up)
        y = conv_block(uc.conv1, uc.norm1, torch.cat((up, e), 1))
        x = conv_block(uc.conv2, uc.norm2, y)
    return F.conv3d(x, model.conv_final.weight, model.conv_final.bias)
