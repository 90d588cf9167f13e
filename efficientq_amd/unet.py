"""Residual 3D U-Net host graph ("UResQ") with the reference's module tree, so that the
reference's checkpoints load unchanged (identical ``state_dict`` keys) and layers are visited
in the same order.  Reference: src/models/model_blk.py:49-207, factoryQ.py:66-81,182-237,
factory_blk.py:18-166.  Only the quantised convolutions compute through the HIP library; ReLU /
pooling / skip additions are torch device ops (plumbing between layers); the trilinear up-sampling has a channels-last
library kernel (TrilinearUp).

Parity-relevant behaviours kept on purpose:
  * the first ReLU of every conv unit is in-place, so a residual block adds relu(x), not x
    (quirk Q10, factoryQ.py:76-77 with factory_blk.py:162-166);
  * deep-supervision heads are plain ``nn.Conv3d`` and are never quantised (quirk Q17);
  * ``forward`` returns all heads stacked, M x N x C x D x H x W (model_blk.py:201-207).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.nn as nn


class PassModule(nn.Module):
    def forward(self, x):
        return x


class TrilinearUp(nn.Upsample):
    """nn.Upsample(scale_factor, mode='trilinear') of the decoder (factory_blk.py:70-93).  On a HIP device, without
    autograd, for per-axis factors 1 / 2 it runs the library's channels-last kernel (effq_upsample_trilinear: the
    framework's kernel indexes NCDHW and took 18 ms per calibration on channels-last tensors); otherwise the module is
    exactly nn.Upsample.  No parameters: checkpoints are unaffected."""

    def forward(self, x):
        sf = self.scale_factor
        sc = tuple(int(v) for v in sf) if isinstance(sf, (tuple, list)) else (int(sf),) * 3
        plain = (x.is_cuda and x.dim() == 5 and x.dtype == torch.float32 and len(sc) == 3 and
                 all(v in (1, 2) for v in sc) and
                 tuple(float(v) for v in (sf if isinstance(sf, (tuple, list)) else (sf,) * 3)) == tuple(float(v) for v in sc)
                 and not (torch.is_grad_enabled() and x.requires_grad) and FAST_UPSAMPLE)
        if not plain:
            return super().forward(x)
        from .hip_ops import get_ops
        from .qconv import from_ndhwc, to_ndhwc
        return from_ndhwc(get_ops(x.device).upsample_trilinear(to_ndhwc(x), sc))


import os as _os
FAST_UPSAMPLE = _os.environ.get("EFFQ_FAST_UPSAMPLE", "1") != "0"


class ConvUnit(nn.Module):
    """One conv with its norm / non-linearity in the order given by ``kind``:
    'pre' = BN-ReLU-Drop-Conv, 'mid' = ReLU-Drop-Conv-BN, 'post' = Drop-Conv-BN-ReLU
    (factoryQ.py:29-81).  Attribute names (relu, do, conv, bn) are the checkpoint keys."""

    def __init__(self, kind, cin, cout, k, stride, pad, dil, Conv, bn, relu_inplace=True, drop=0.0):
        super().__init__()
        self.kind = kind
        if kind == 'pre':
            self.bn = bn(cin)
            self.relu = nn.ReLU(relu_inplace)
            self.do = nn.Dropout3d(drop) if drop else PassModule()
            self.conv = Conv(cin, cout, k, stride, pad, dil, 1, False)
        elif kind == 'mid':
            self.relu = nn.ReLU(relu_inplace)
            self.do = nn.Dropout3d(drop) if drop > 0 else PassModule()
            self.conv = Conv(cin, cout, k, stride, pad, dil, 1, False)
            self.bn = bn(cout)
        else:
            self.do = nn.Dropout3d(drop) if drop > 0 else PassModule()
            self.conv = Conv(cin, cout, k, stride, pad, dil, 1, False)
            self.bn = bn(cout)
            self.relu = nn.ReLU(relu_inplace)

    def forward(self, x):
        if self.kind == 'pre':
            return self.conv(self.do(self.relu(self.bn(x))))
        if self.kind == 'mid':
            return self.bn(self.conv(self.do(self.relu(x))))
        return self.relu(self.bn(self.conv(self.do(x))))


class ResUnit(nn.Module):
    """Two 3^3 conv units plus identity / 1^3 projection skip (factory_blk.py:147-166)."""

    def __init__(self, kind, cin, cout, drop, dil, Conv, bn):
        super().__init__()
        self.change_dim = cin != cout
        self.block1 = ConvUnit(kind, cin, cout, 3, 1, dil, dil, Conv, bn, True, 0)
        self.block2 = ConvUnit(kind, cout, cout, 3, 1, dil, dil, Conv, bn, True, drop)
        self.projection = Conv(cin, cout, 1, 1, 0, bias=False) if self.change_dim else PassModule()

    def forward(self, x):
        out = self.block2(self.block1(x))       # block1's in-place ReLU has already rewritten x
        return out + self.projection(x)


class Fuser(nn.Module):
    """Up-sample the deep feature and add the skip (factory_blk.py:70-93)."""

    def __init__(self, kind, cin, cskip, scale, Conv, bn):
        super().__init__()
        self.upsampler = nn.Sequential()
        if cin != cskip:
            self.upsampler.add_module('block', ConvUnit(kind, cin, cskip, 1, 1, 0, 1, Conv, bn, False, 0))
        self.upsampler.add_module('trilinear', TrilinearUp(scale_factor=scale, mode='trilinear'))

    def forward(self, x, skip):
        return self.upsampler(x) + skip


def _scaled(t, f):
    return tuple(i * f for i in t) if isinstance(t, (tuple, list)) else t * f


class UResQ(nn.Module):
    def __init__(self, QConv, num_mod, num_classes, depth_config, width_config, dilation_config,
                 init_stride=1, stride=2, drop_rate=0.25, bn=nn.BatchNorm3d, ds=False, blk_type='mid',
                 q_weight=True, qlvl=8, q_act=True, qlvl_act=8, q_first=None, q_last=None,
                 hetero_param: Optional[dict] = None, init_kernel=3, **kwQ):
        super().__init__()
        assert len(depth_config) == len(width_config) == len(dilation_config)
        assert len(depth_config) % 2 == 1, 'Can only have odd number of UBlocks'
        hp = hetero_param or {}
        aniso_depth = hp.get('aniso_pool_depth', 99999)
        aniso_stride = hp.get('aniso_pool_stride', (2, 2, 1))
        drop_cut = hp.get('drop_cut_thres', -1)
        ds_limit = hp.get('ds_depth_limit', 99999)
        self.init_stride = init_stride
        kind = blk_type

        def wrap(qw, lv, qa, lva) -> Callable:
            if QConv in (nn.Conv2d, nn.Conv3d):
                return QConv

            def make(cin, cout, k, s=1, p=0, d=1, g=1, bias=True):
                return QConv(cin, cout, k, s, p, d, g, bias, q_weight=qw, qlvl=lv, q_act=qa, qlvl_act=lva, **kwQ)
            return make

        ConvQ = wrap(q_weight, qlvl, q_act, qlvl_act)
        ConvFirst = wrap(q_first[0] > 0, q_first[0], q_first[1] > 0, q_first[1]) if q_first else nn.Conv3d
        ConvLast = wrap(q_last[0] > 0, q_last[0], q_last[1] > 0, q_last[1]) if q_last else nn.Conv3d

        n = len(depth_config)
        k0 = init_kernel
        self.conv0 = nn.Sequential()
        self.conv0.add_module('conv', ConvFirst(num_mod, width_config[0], k0, init_stride, (k0 - 1) // 2, bias=False))
        if kind != 'pre':
            self.conv0.add_module('bn', bn(width_config[0]))
        if kind == 'post':
            self.conv0.add_module('relu', nn.ReLU(True))

        self.u_blocks = nn.Sequential()
        self.trans_downs = nn.Sequential()
        self.trans_ups = nn.Sequential()
        self.classifiers = nn.Sequential()
        for i in range(n):
            w = width_config[i]
            dr = drop_rate
            if dr > 0 and w < drop_cut:
                dr = min(drop_rate / 2, 0.2)
            stage = nn.Sequential() if depth_config[i] > 0 else PassModule()
            for j in range(depth_config[i]):
                stage.add_module(f'Layer{j + 1}', ResUnit(kind, w, w, dr, dilation_config[i], ConvQ, bn))
            self.u_blocks.add_module(f'UResBlock{i + 1}', stage)
            if i < n // 2:
                pool_k = stride if i < aniso_depth else aniso_stride
                down = nn.Sequential()
                down.add_module('pool', nn.MaxPool3d(pool_k, pool_k))
                down.add_module('block', ConvUnit(kind, w, width_config[i + 1], 1, 1, 0, 1, ConvQ, bn, True, 0))
                self.trans_downs.add_module(f'TransDown{i + 1}', down)
            elif i < n - 1:
                sc = stride if i >= n - 1 - aniso_depth else aniso_stride
                self.trans_ups.add_module(f'TransUp{i + 1}', Fuser(kind, w, width_config[i + 1], sc, ConvQ, bn))
                if ds:
                    if ds != 'simple':
                        raise NotImplementedError("only ds='simple' heads are on the calibrated configs")
                    head = None
                    if n - i <= ds_limit:
                        head = nn.Sequential()
                        head.add_module('classifier', nn.Conv3d(w, num_classes, 1, 1, 0))
                        up = _scaled(init_stride, 2 ** len(width_config[i + 1:]))
                        if up not in (1, (1, 1), (1, 1, 1)):
                            head.add_module('extra_up', TrilinearUp(scale_factor=up, mode='trilinear'))
                    self.classifiers.add_module(f'AuxClassifier{i + 1}', head)
        self.final_cls = nn.Sequential()
        self.final_cls.add_module('cls', ConvLast(width_config[-1], num_classes, 1, 1, 0))
        if init_stride not in (1, (1, 1), (1, 1, 1)):
            self.final_cls.add_module('extra_up', TrilinearUp(scale_factor=init_stride, mode='trilinear'))

    def load_state_dict(self, state_dict, strict=True, init=True):
        r = super().load_state_dict(state_dict, strict)
        if init:
            self.qparam_init()
        return r

    def qparam_init(self):
        for m in self.modules():
            if 'QConv' in m.__class__.__name__ and hasattr(m, 'qparam_init'):
                m.qparam_init()

    def perform_quantization(self):
        for m in self.modules():
            if 'QConv' in m.__class__.__name__:
                m.perform_quantization()

    def forward(self, x, feature_out=False):
        nb, nd = len(self.u_blocks), len(self.trans_downs)
        f = self.conv0(x)
        skips, outs = [], []
        for i in range(nb):
            f = self.u_blocks[i](f)
            if i < nd:
                skips.append(f)
                f = self.trans_downs[i](f)
            elif i < nb - 1:
                j = i - nd
                if len(self.classifiers) and self.classifiers[j] is not None:
                    outs.append(self.classifiers[j](f))
                f = self.trans_ups[j](f, skips[-(j + 1)])
        if feature_out:
            return f
        outs.append(self.final_cls(f))
        return torch.stack(outs, dim=0)
