"""Train CLI with the reference's flag names and plugin lookup (main.py:13-43):

    python main.py --model ConvVAE --trainer VAETrainer \
        --architecture architecture-vae-vcc2016.json

Multi-GPU: launch one process per GPU with torch.distributed.run; the trainer
all-reduces gradients over RCCL.
"""
import argparse
import json
import os
import sys
from importlib import import_module

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--logdir_root', default=None, help='root of log dir')
    p.add_argument('--logdir', default=None, help='log dir')
    p.add_argument('--restore_from', default=None, help='restore from dir (not from *.ckpt)')
    p.add_argument('--gpu_cfg', default=None, help='GPU configuration')
    p.add_argument('--summary_freq', type=int, default=1000, help='Update summary')
    p.add_argument('--ckpt', default=None, help='specify the ckpt in restore_from (if there are multiple ckpts)')
    p.add_argument('--architecture', default='architecture-vae-vcc2016.json', help='network architecture')
    p.add_argument('--model_module', default='model.vae', help='Model module')
    p.add_argument('--model', default=None, help='Model: ConvVAE')
    p.add_argument('--trainer_module', default='trainer.vae', help='Trainer module')
    p.add_argument('--trainer', default=None, help='Trainer: VAETrainer')
    p.add_argument('--seed', type=int, default=0, help='(new) weight-init / shuffle / sampler seed')
    p.add_argument('--precision', default=None, help='(new) bf16x2 (default) | bf16x3 (fp32-exact) | bf16')
    args = p.parse_args(argv)
    if args.model is None or args.trainer is None:          # main.py:33-37
        raise ValueError('\n  Both `model` and `trainer` should be assigned.'
                         '\n  Use `python main.py --help` to see applicable options.')
    return args


def main(argv=None):
    ''' NOTE: The input is rescaled to [-1, 1] '''
    import torch
    import torch.distributed as dist
    from analyzer import read, Tanhize, load_npf
    from util.wrapper import validate_log_dirs

    args = parse_args(argv)
    MODEL = getattr(import_module(args.model_module), args.model)          # main.py:39-43
    TRAINER = getattr(import_module(args.trainer_module), args.trainer)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')

    # the default logdir carries a wall-clock stamp: rank 0 chooses it, everybody else receives it
    box = [validate_log_dirs(args) if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    dirs = box[0]
    os.makedirs(dirs['logdir'], exist_ok=True)
    with open(args.architecture) as f:
        arch = json.load(f)
    if rank == 0:                                                          # main.py:54-55
        with open(os.path.join(dirs['logdir'], os.path.basename(args.architecture)), 'w') as f:
            json.dump(arch, f, indent=4)

    normalizer = Tanhize(xmax=load_npf('./etc/xmax.npf'), xmin=load_npf('./etc/xmin.npf'))
    image, label = read(file_pattern=arch['training']['datadir'], batch_size=arch['training']['batch_size'],
                        capacity=2048, min_after_dequeue=1024, normalizer=normalizer, seed=args.seed,
                        rank=rank, world=world)
    machine = MODEL(arch, seed=args.seed, **({'precision': args.precision} if args.precision else {}))
    loss = machine.loss(image, label)
    trainer = TRAINER(loss, arch, args, dirs)
    trainer.train(nIter=arch['training']['max_iter'], machine=machine)


if __name__ == '__main__':
    main()
