"""CLI / YAML surface and factories of the ``ptq`` mission, same flag names and semantics as the
reference (src/entrance.py:17-128, src/definer.py:130-248,286-329): YAML values override the
command line for every non-null key (quirk Q15); ``qlvl_*`` are LEVEL counts (4 => 2-bit);
``q_first/q_last "W,A"`` with A=-1 => full-precision activations (quirk Q14); every ``lwq_*``
argument is forwarded to the conv constructor as ``**kwQ``.
``--qconv effq`` (and the explicit alias ``effq_hip``) selects the MI355X calibrator."""
from __future__ import annotations

import argparse

import torch.nn as nn
import yaml

from .qconv import EfficientQConvHIP
from .unet import UResQ

QCONV_REGISTRY = {'conv': nn.Conv3d, 'effq': EfficientQConvHIP, 'effq_hip': EfficientQConvHIP}


def merge_config(cfg: str, args: argparse.Namespace):
    with open(cfg, 'r') as fid:
        config = yaml.load(fid, Loader=yaml.FullLoader)
    for k, v in config.items():
        if v is not None:
            setattr(args, k, v)
    return args


def build_parser():
    p = argparse.ArgumentParser(description='EfficientQ PTQ calibration on MI355X')
    p.add_argument('mission', choices=['ptq'])
    p.add_argument('--pretrain')
    p.add_argument('--resume')
    p.add_argument('--device', default=0, type=int, help='GPU ID.')
    p.add_argument('--task')
    p.add_argument('--suffix', default="", type=str)
    p.add_argument('--test_fp', action='store_true')
    p.add_argument('--config', type=str)
    p.add_argument('--data_dir')
    p.add_argument('--split_dir')
    p.add_argument('--round', default='1', type=str)
    p.add_argument('--patch_size')
    p.add_argument('--bin_label')
    p.add_argument('--multi_label')
    p.add_argument('--model', default='UResQ')
    p.add_argument('--nMod', type=int)
    p.add_argument('--nClass', type=int)
    p.add_argument('--init_stride', type=str, default='1')
    p.add_argument('--depth')
    p.add_argument('--width')
    p.add_argument('--dilation')
    p.add_argument('--nla', default='relu')
    p.add_argument('--norm', type=str, default='bn')
    p.add_argument('--drop_rate', default=0.2, type=float)
    p.add_argument('--ds', type=str, default=None, choices=['simple', 'complex', ''])
    p.add_argument('--init_kernel', default=3, type=int)
    p.add_argument('--hetero_dim', action='store_true')
    p.add_argument('--blk', type=str, default='pre')
    p.add_argument('--no_test', action='store_true')
    p.add_argument('--qconv', default='conv')
    p.add_argument('--qlvl_w', type=int)
    p.add_argument('--qlvl_a', type=int)
    p.add_argument('--q_first')
    p.add_argument('--q_last')
    p.add_argument('--debug', action='store_true')
    p.add_argument('--lwq_dataid', type=int, default=0)
    p.add_argument('--lwq_batchsz', type=int, default=1)
    p.add_argument('--lwq_patchsz')
    p.add_argument('--lwq_verbose', action='store_true')
    p.add_argument('--save_nii', action='store_true')
    # new in this build: synthetic calibration volumes (no dataset is shipped with either repo)
    p.add_argument('--synthetic', action='store_true', help='calibrate on seeded synthetic volumes')
    p.add_argument('--snap_dir', default=None)
    return p


def _pair(s):
    return [int(x) for x in s.split(',')] if s else None


def get_conv_class(args):
    """(QConv class, Qinfo string naming the snapshot dir, kwQ) -- definer.py:286-329."""
    name = args.qconv.lower()
    if name not in QCONV_REGISTRY:
        raise RuntimeError('Unknown QConv name: %s' % args.qconv)
    if name == 'conv':
        return nn.Conv3d, 'FP', {}
    q_weight, q_act = args.qlvl_w > 0, args.qlvl_a > 0
    qlvl, qlvl_act = args.qlvl_w, (args.qlvl_a if q_act else 256)
    kwQ = {a: getattr(args, a) for a in dir(args) if a[:4] == 'lwq_'}
    if q_act and q_weight:
        info = 'bothQw{}a{}'.format(qlvl, qlvl_act)
    elif q_act:
        info = 'actQa{}'.format(qlvl_act)
    else:
        info = 'weightQw{}'.format(qlvl)
    return QCONV_REGISTRY[name], args.qconv + '_' + info, kwQ


def get_model_cube(args, QConv=nn.Conv3d, kwQ=None):
    """definer.py:130-248."""
    kwQ = kwQ or {}
    task = args.task.lower()
    nMod = args.nMod if args.nMod else (4 if task == 'brats' else 1)
    nClass = args.nClass if args.nClass else (4 if task == 'brats' else 3)
    if getattr(args, 'bin_label', None):
        nClass = 2
    if getattr(args, 'multi_label', None):
        nClass -= 1
    if args.model not in ('UResQ',):
        raise RuntimeError('Unknown model name: %s' % args.model)
    st = str(args.init_stride)
    init_stride = tuple(int(x) for x in st.split(',')) if ',' in st else (int(st),) * 3
    if args.qconv.lower() == 'conv':
        q_weight = q_act = False
        q_first = q_last = qlvl = qlvl_act = None
    else:
        q_weight, q_act = args.qlvl_w > 0, args.qlvl_a > 0
        qlvl, qlvl_act = args.qlvl_w, (args.qlvl_a if q_act else 256)
        q_first, q_last = _pair(args.q_first), _pair(args.q_last)
    if args.nla.lower() not in ('relu', 'reluf'):
        raise RuntimeError('Unknown NLA name: %s' % args.nla)
    if args.norm.lower() != 'bn':
        raise NotImplementedError('Norm type should be in BN')
    width = [int(i) for i in args.width.split(',')] if args.width else [32, 64, 128, 256, 128, 64, 32]
    depth = [int(i) for i in args.depth.split(',')] if args.depth else [1] * len(width)
    dil = [int(i) for i in args.dilation.split(',')] if args.dilation else [1] * len(width)
    hp = {'drop_cut_thres': 128, 'ds_depth_limit': 3 if 2 in init_stride else 4}
    if args.hetero_dim:
        hp['aniso_pool_depth'] = 9999 if 2 in init_stride else 4
        hp['aniso_pool_stride'] = (2, 2, 1)
    model = UResQ(QConv, nMod, nClass, depth_config=depth, width_config=width, dilation_config=dil,
                  init_stride=init_stride, stride=2, drop_rate=args.drop_rate, bn=nn.BatchNorm3d, ds=args.ds,
                  blk_type=args.blk, q_weight=q_weight, qlvl=qlvl, q_act=q_act, qlvl_act=qlvl_act,
                  q_first=q_first, q_last=q_last, hetero_param=hp, init_kernel=args.init_kernel, **kwQ)
    num_mo = min(hp['ds_depth_limit'], len(depth) // 2 + 1) if args.ds else 1
    cube = {'model': model, 'init_func': None, 'pretrain': args.pretrain, 'resume': getattr(args, 'resume', None),
            'optimizer_list': None, 'num_mo': num_mo, 'nClass': nClass, 'nMod': nMod}
    return cube, args.model + '_' + args.norm.upper()


BRATS_NET = dict(task='brats', model='UResQ', nMod=4, nClass=4, multi_label='brats', init_stride='2,2,2',
                 depth='1,1,1,1,1,1,1', width='32,64,128,256,128,64,32', dilation='1,1,1,1,1,1,1', nla='relu',
                 norm='bn', drop_rate=0.5, ds='simple', hetero_dim=True, blk='mid', init_kernel=3,
                 qconv='effq', q_first='256,-1', q_last='256,-1')       # config/brats_ptq.yaml
LITS_NET = dict(task='lits', model='UResQ', nMod=1, nClass=3, multi_label=None, init_stride='2,2,1',
                depth='1,1,1,1,1,1,1,1,1', width='32,64,128,256,512,256,128,64,32',
                dilation='1,1,1,1,1,1,1,1,1', nla='relu', norm='bn', drop_rate=0.5, ds='simple', hetero_dim=True,
                blk='mid', init_kernel=3, qconv='effq', q_first='256,-1', q_last='256,-1')   # config/lits_ptq.yaml
TINY_NET = dict(task='lits', model='UResQ', nMod=1, nClass=3, multi_label=None, init_stride='1', depth='1,1,1',
                width='8,16,8', dilation=None, nla='relu', norm='bn', drop_rate=0.5, ds='simple', hetero_dim=True,
                blk='mid', init_kernel=3, qconv='effq', q_first='256,-1', q_last='256,-1')   # BASELINE config 1


def make_args(net: dict, qlvl_w: int, qlvl_a: int, **over):
    base = dict(pretrain=None, resume=None, device=0, round='1', suffix='', config=None, test_fp=False,
                no_test=True, save_nii=False, bin_label=None, lwq_dataid=0, lwq_batchsz=1, lwq_patchsz=None,
                lwq_verbose=False, qlvl_w=qlvl_w, qlvl_a=qlvl_a)
    base.update(net)
    base.update(over)
    return argparse.Namespace(**base)
