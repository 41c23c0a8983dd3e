"""Translate a torch model of the dense prediction path into the layer plan of the HIP library.

Two model families reach ``Predictor`` in SyConn's dense path:

* an elektronn3 ``UNet`` (usually as a TorchScript ``model.pts``, /root/reference/syconn/handler/prediction.py:777,
  traced at /root/reference/syconn/cnn/cnn_myelin.py:107).  Only its ``state_dict`` is used here: the architecture
  (number of blocks, filters, planar blocks, normalisation kind, legacy ``batch_norm=True`` layout) is recovered
  from the parameter names and shapes, so a real SyConn model file loads without elektronn3 being installed.
* a plain ``nn.Sequential`` of Conv3d(k=3, 'same') / ReLU ending in a 1x1x1 Conv3d (BASELINE.json config 1).

The result is ``(list[OpDesc], weight blob float32, info)``; ``sd_model_create`` does folding / packing / upload.
"""
import re
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

BN_EPS = 1e-5   # torch.nn.BatchNorm3d / GroupNorm default; not stored in a state_dict


class _Blob:
    def __init__(self):
        self.parts: List[np.ndarray] = []
        self.n = 0

    def add(self, t) -> int:
        a = np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy()).reshape(-1)
        off = self.n
        self.parts.append(a)
        self.n += a.size
        return off

    def array(self) -> np.ndarray:
        return np.concatenate(self.parts).astype(np.float32) if self.parts else np.zeros(1, np.float32)


def _desc(**kw) -> L.OpDesc:
    d = L.OpDesc()
    for f, _ in L.OpDesc._fields_:
        setattr(d, f, -1 if f.endswith('_off') or f in ('src1',) else 0)
    d.eps = BN_EPS
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _norm_kind(sd: Dict[str, torch.Tensor], prefix: str) -> str:
    if prefix + '.running_mean' in sd:
        return 'batch'
    if prefix + '.weight' in sd:
        return 'group'
    return 'none'


def plan_from_unet_state_dict(sd: Dict[str, torch.Tensor], group_norm_groups: int = 8):
    """elektronn3 UNet state_dict -> plan.  Block structure: SURVEY.md rows U1-U5."""
    n_blocks = len({int(m.group(1)) for k in sd for m in [re.match(r'down_convs\.(\d+)\.conv1\.weight', k)] if m})
    n_up = len({int(m.group(1)) for k in sd for m in [re.match(r'up_convs\.(\d+)\.upconv\.weight', k)] if m})
    if n_blocks < 1 or n_up != n_blocks - 1 or 'conv_final.weight' not in sd:
        raise ValueError('state_dict is not an elektronn3-style UNet (down_convs/up_convs/conv_final)')
    blob = _Blob()
    ops: List[L.OpDesc] = []
    nxt = [1]

    def new_buf() -> int:
        b = nxt[0]
        nxt[0] += 1
        return b

    def add_norm_fields(d: L.OpDesc, prefix: str, kind: str):
        if kind == 'batch':
            d.norm = 1
            d.gamma_off = blob.add(sd[prefix + '.weight'])
            d.beta_off = blob.add(sd[prefix + '.bias'])
            d.mean_off = blob.add(sd[prefix + '.running_mean'])
            d.var_off = blob.add(sd[prefix + '.running_var'])

    def groupnorm(buf: int, prefix: str, crop_ref: int = -1):
        ops.append(_desc(kind=L.SD_OP_GROUPNORM, src0=buf, src1=crop_ref, dst=buf, groups=group_norm_groups, relu=1,
                         gamma_off=blob.add(sd[prefix + '.weight']), beta_off=blob.add(sd[prefix + '.bias'])))

    def conv(src0: int, src1: int, cin0: int, cin1: int, wkey: str, norm_prefix: str) -> Tuple[int, int]:
        w = sd[wkey + '.weight']
        cout, cin, kz, ky, kx = w.shape
        if cin != cin0 + max(cin1, 0) or (ky, kx) != (3, 3) or kz not in (1, 3):
            raise ValueError(f'{wkey}: unsupported conv shape {tuple(w.shape)}')
        kind = _norm_kind(sd, norm_prefix)
        dst = new_buf()
        d = _desc(kind=L.SD_OP_CONV, src0=src0, src1=src1, dst=dst, cin0=cin0, cin1=max(cin1, 0), cout=cout,
                  kz=kz, ky=3, kx=3, relu=0 if kind == 'group' else 1,
                  w_off=blob.add(w), b_off=blob.add(sd[wkey + '.bias']))
        add_norm_fields(d, norm_prefix, kind)
        ops.append(d)
        if kind == 'group':
            groupnorm(dst, norm_prefix)
        return dst, cout

    x, cx = 0, int(sd['down_convs.0.conv1.weight'].shape[1])
    if cx != 1:
        raise ValueError('the dense path feeds single-channel EM data (in_channels must be 1)')
    enc = []
    for i in range(n_blocks):
        p = f'down_convs.{i}'
        planar = sd[p + '.conv1.weight'].shape[2] == 1
        x, cx = conv(x, -1, cx, -1, p + '.conv1', p + '.norm0')
        x, cx = conv(x, -1, cx, -1, p + '.conv2', p + '.norm1')
        enc.append((x, cx))
        if i < n_blocks - 1:
            dst = new_buf()
            ops.append(_desc(kind=L.SD_OP_POOL, src0=x, dst=dst, kz=1 if planar else 2, ky=2, kx=2, cin0=cx, cout=cx))
            x = dst
    for i in range(n_up):
        p = f'up_convs.{i}'
        e, ce = enc[-(i + 2)]
        w = sd[p + '.upconv.weight']   # ConvTranspose3d: [cin][cout][kz][2][2]
        cin, cout, kz, ky, kx = w.shape
        if cin != cx or (ky, kx) != (2, 2) or kz not in (1, 2):
            raise ValueError(f'{p}.upconv: unsupported shape {tuple(w.shape)}')
        kind = _norm_kind(sd, p + '.norm0')
        u = new_buf()
        d = _desc(kind=L.SD_OP_UPCONV, src0=x, dst=u, cin0=cin, cout=cout, kz=kz, ky=2, kx=2,
                  relu=0 if kind == 'group' else 1, w_off=blob.add(w), b_off=blob.add(sd[p + '.upconv.bias']))
        add_norm_fields(d, p + '.norm0', kind)
        ops.append(d)
        if kind == 'group':
            groupnorm(u, p + '.norm0', crop_ref=e)
        x, cx = conv(u, e, cout, ce, p + '.conv1', p + '.norm1')
        x, cx = conv(x, -1, cx, -1, p + '.conv2', p + '.norm2')
    wf = sd['conv_final.weight']
    if tuple(wf.shape[2:]) != (1, 1, 1) or wf.shape[1] != cx:
        raise ValueError('conv_final must be a 1x1x1 convolution')
    ops.append(_desc(kind=L.SD_OP_FINAL, src0=x, dst=0, cin0=cx, cout=int(wf.shape[0]), kz=1, ky=1, kx=1,
                     w_off=blob.add(wf), b_off=blob.add(sd['conv_final.bias'])))
    info = dict(family='unet', n_blocks=n_blocks, out_channels=int(wf.shape[0]), n_buffers=nxt[0])
    return ops, blob.array(), info


def plan_from_sequential(model: torch.nn.Sequential):
    """Conv3d(k=3|(1,3,3), 'same') [+ BatchNorm3d] + ReLU ... Conv3d(k=1): BASELINE.json config 1."""
    import torch.nn as nn
    mods = [m for m in model if not isinstance(m, (nn.Identity, nn.Dropout, nn.Dropout3d))]
    blob = _Blob()
    ops: List[L.OpDesc] = []
    x, cx, nxt, i = 0, 1, 1, 0
    while i < len(mods):
        m = mods[i]
        if not isinstance(m, nn.Conv3d):
            raise ValueError(f'unsupported layer {type(m).__name__} at position {i}')
        k = tuple(m.kernel_size)
        last = i == len(mods) - 1
        if k == (1, 1, 1) and last:
            if m.in_channels != cx:
                raise ValueError('channel mismatch in final conv')
            ops.append(_desc(kind=L.SD_OP_FINAL, src0=x, dst=0, cin0=cx, cout=m.out_channels, kz=1, ky=1, kx=1,
                             w_off=blob.add(m.weight), b_off=blob.add(m.bias)))
            i += 1
            continue
        if k not in ((3, 3, 3), (1, 3, 3)) or tuple(m.padding) != (k[0] // 2, 1, 1) or tuple(m.stride) != (1, 1, 1) \
                or m.in_channels != cx or m.bias is None:
            raise ValueError(f'unsupported Conv3d {m}')
        d = _desc(kind=L.SD_OP_CONV, src0=x, dst=nxt, cin0=cx, cin1=0, cout=m.out_channels, kz=k[0], ky=3, kx=3,
                  w_off=blob.add(m.weight), b_off=blob.add(m.bias))
        i += 1
        if i < len(mods) and isinstance(mods[i], nn.BatchNorm3d):
            bn = mods[i]
            d.norm, d.eps = 1, bn.eps
            d.gamma_off, d.beta_off = blob.add(bn.weight), blob.add(bn.bias)
            d.mean_off, d.var_off = blob.add(bn.running_mean), blob.add(bn.running_var)
            i += 1
        if i < len(mods) and isinstance(mods[i], nn.ReLU):
            d.relu = 1
            i += 1
        ops.append(d)
        x, cx, nxt = nxt, m.out_channels, nxt + 1
    if not ops or ops[-1].kind != L.SD_OP_FINAL:
        raise ValueError('the network must end with a 1x1x1 Conv3d')
    return ops, blob.array(), dict(family='sequential', out_channels=int(ops[-1].cout), n_buffers=nxt)


# what a state_dict cannot show: the plan assumes ReLU activations and transposed-convolution upsampling (all of SyConn's
# models: activation='relu', syconn/cnn/cnn_*.py).  A traced / scripted model or an nn.Module that uses anything else is
# refused instead of being predicted with the wrong function.
_UNSUPPORTED_ATEN = ('aten::leaky_relu', 'aten::prelu', 'aten::rrelu', 'aten::elu', 'aten::selu', 'aten::celu', 'aten::gelu',
                     'aten::silu', 'aten::mish', 'aten::hardswish', 'aten::softplus', 'aten::tanh', 'aten::sigmoid',
                     'aten::upsample', 'aten::interpolate', 'aten::instance_norm', 'aten::layer_norm')


def _refuse_unsupported_layers(model):
    import torch.nn as nn
    if isinstance(model, torch.jit.ScriptModule):
        try:
            graph = str(model.inlined_graph)
        except Exception:       # (no graph to look at: nothing to check)
            return
        found = sorted({op for op in _UNSUPPORTED_ATEN if op in graph})
        if found:
            raise ValueError(f'the model uses {", ".join(found)}: only ReLU activations, BatchNorm / GroupNorm and transposed-'
                             f'convolution upsampling are implemented (what syconn/cnn/cnn_*.py build)')
    elif isinstance(model, nn.Module):
        ok = (nn.Conv3d, nn.ConvTranspose3d, nn.BatchNorm3d, nn.GroupNorm, nn.ReLU, nn.MaxPool3d, nn.Identity, nn.Dropout,
              nn.Dropout3d, nn.Sequential, nn.ModuleList)
        bad = sorted({type(m).__name__ for m in model.modules()
                      if not list(m.children()) and not isinstance(m, ok) and type(m).__module__.startswith('torch.nn')})
        if bad:
            raise ValueError(f'the model contains {", ".join(bad)}: only Conv3d / ConvTranspose3d / BatchNorm3d / GroupNorm / ReLU '
                             f'/ MaxPool3d networks are implemented (what syconn/cnn/cnn_*.py build)')


def plan_from_model(model, group_norm_groups: Optional[int] = None):
    """Dispatch on the model type: ScriptModule / nn.Module with a UNet state_dict, or nn.Sequential."""
    import torch.nn as nn
    if not isinstance(model, dict):
        _refuse_unsupported_layers(model)
    if isinstance(model, nn.Sequential) and not isinstance(model, torch.jit.ScriptModule):
        return plan_from_sequential(model)
    sd = model if isinstance(model, dict) else model.state_dict()
    groups = group_norm_groups
    if groups is None:
        groups = 8
        for m in (model.modules() if hasattr(model, 'modules') and not isinstance(model, torch.jit.ScriptModule) else []):
            if isinstance(m, nn.GroupNorm):
                groups = m.num_groups
                break
    return plan_from_unet_state_dict(sd, group_norm_groups=groups)
