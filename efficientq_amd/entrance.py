"""Command-line entry of the ``ptq`` mission, same surface as the reference's ``src/entrance.py``
(`python entrance.py ptq --qlvl_w 4 --qlvl_a 4 --round 1 --config ../config/brats_ptq.yaml --pretrain ...`):

    python -m efficientq_amd.entrance ptq --config config/brats_ptq.yaml --qlvl_w 4 --qlvl_a 4 \
        --pretrain pretrain/brats/round1/mid/state_0500.pkl --synthetic --lwq_batchsz 2 --snap_dir out/

Neither repository ships data, so ``--synthetic`` replaces the reference's data cube by seeded synthetic
volumes (``synth.py``); without ``--pretrain`` a seeded random-init network stands in for the checkpoint.
Real data goes through ``calibrate.do_ptq(args, model_cube, data_cube, tester, snap_dir)`` with the
reference's own ``data_cube`` / ``tester`` objects.  With ``torchrun --nproc-per-node N``
the calibration volumes are sharded over N GPUs.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from . import calibrate as K
from . import config as Cf
from . import synth


class _SnapshotWriter:
    """Stand-in for the reference's PTQTester (utils/tester.py:37-51): the three snapshot files only."""

    def __init__(self, model, root):
        self.model, self.root = model, root

    def test_as_is(self, *a, **k):
        print('[entrance] evaluation (sliding-window Dice) is outside the calibrated hot path: skipped')

    def snapshot(self, name, compress=False):
        sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        path = os.path.join(self.root, name)
        if compress:
            np.savez_compressed(path, **{k: v.numpy() for k, v in sd.items()})
        else:
            torch.save({'state_dict': sd}, path)
        print(f'[entrance] wrote {path}')


class _SyntheticCube:
    def __init__(self, task, n, size):
        vols = synth.calib_batch(task, range(n), size)

        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return n

            def __getitem__(self, i):
                return vols[i], torch.zeros(vols.shape[2:], dtype=torch.long)

            def use_fix_transform(self):
                pass
        self.trainseqloader = torch.utils.data.DataLoader(DS(), 1, shuffle=False)


def main(argv=None):
    args = Cf.build_parser().parse_args(argv)
    if args.config:
        args = Cf.merge_config(args.config, args)
    if args.mission != 'ptq':
        raise NotImplementedError(args.mission)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        # (no device_id: the group is the rendezvous only; the collectives are RCCL calls on the kernels' stream, rccl.py -
        # an eagerly created framework communicator brings streams of its own that slow the calibration by 20 %)
        dist.init_process_group('nccl')
        args.device = local
    QConv, Qinfo, kwQ = Cf.get_conv_class(args)
    cube, info = Cf.get_model_cube(args, QConv, kwQ)
    model = cube['model']
    snap = args.snap_dir or os.path.join('exp_ptq', args.task, f'{info}_{time.strftime("%m%d%H%M")}_{Qinfo}{args.suffix}')
    os.makedirs(snap, exist_ok=True)
    if not args.pretrain:
        synth.randomise_network(model, 0)
        args.pretrain = os.path.join(snap, 'round%s_random_init.pkl' % args.round)
        torch.save({'state_dict': model.state_dict()}, args.pretrain)
        cube['pretrain'] = args.pretrain
        print(f'[entrance] no --pretrain given: seeded random-init network saved to {args.pretrain}')
    if not args.synthetic:
        raise SystemExit('no dataset is shipped: pass --synthetic, or call calibrate.do_ptq with your own data_cube')
    size = [int(v) for v in args.lwq_patchsz.split(',')] if args.lwq_patchsz else (128 if args.task == 'brats' else 160)
    size = size[0] if isinstance(size, list) and len(set(size)) == 1 else size
    data_cube = _SyntheticCube(args.task, args.lwq_batchsz, size)
    with open(os.path.join(snap, 'cmd.txt'), 'w') as f:
        f.write(' '.join(sys.argv) + '\n')
    K.do_ptq(args, cube, data_cube, _SnapshotWriter(model, snap), snap)


if __name__ == '__main__':
    main()
