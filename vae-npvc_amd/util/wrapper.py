"""Log-dir and checkpoint helpers (behavioural counterpart of the reference's
util/wrapper.py:32-62,99-132; TF Saver replaced by a flat-buffer checkpoint)."""
import os
import re
from datetime import datetime

import torch


def validate_log_dirs(args):
    """logdir = <logdir_root>/train/<MMDD-HHMM-SS-YYYY> (wrapper.py:99-132).  Unlike the
    reference (which crashes on --logdir/--logdir_root) explicit values work."""
    logdir_root = getattr(args, 'logdir_root', None)
    logdir = getattr(args, 'logdir', None)
    restore_from = getattr(args, 'restore_from', None)
    if logdir and restore_from:
        raise ValueError('You can only specify one of the following: --logdir and --restore_from')
    if logdir and logdir_root:
        raise ValueError('You can only specify either --logdir or --logdir_root')
    if logdir_root is None:
        logdir_root = 'logdir'
    if logdir is None:
        stamp = datetime.now().strftime('%m%d-%H%M-%S-%Y')
        logdir = os.path.join(logdir_root, 'train', stamp)
        print('Using default logdir: {}'.format(logdir))
    if restore_from is None:
        restore_from = logdir
    return {'logdir': logdir, 'logdir_root': logdir_root, 'restore_from': restore_from}


def find_ckpt(logdir, ckpt=None):
    """`<logdir>/<ckpt>` or the newest model.ckpt-N under logdir (wrapper.py:32-62); None if there is none."""
    if ckpt is None:
        if not os.path.isdir(logdir):
            return None
        cands = sorted({re.sub(r'\.index$', '', f) for f in os.listdir(logdir) if re.match(r'model\.ckpt-\d+(\.index)?$', f)})
        if not cands:
            return None
        ckpt = max(cands, key=lambda f: int(f.rsplit('-', 1)[1]))
    path = os.path.join(logdir, ckpt)
    return path if (os.path.exists(path) or os.path.exists(path + '.index')) else None


def read_ckpt(path, layout=None):
    """A checkpoint written by VAETrainer.save: {'params', 'm', 'v', 'step', 'layout'} (flat float32 buffers in
    the tensor order of include/vaenpvc.h) -- or, when `<path>.index` exists, a TensorFlow V2 checkpoint of the
    reference's graph (util/tf_checkpoint.py; needs the engine's `layout` to place the variables)."""
    if not os.path.exists(path) and os.path.exists(path + '.index'):
        from util.tf_checkpoint import import_checkpoint
        if layout is None:
            raise ValueError('a TensorFlow checkpoint needs the parameter layout')
        return import_checkpoint(path, layout)
    return torch.load(path, map_location='cpu')


def load(engine, logdir, ckpt=None):
    """Restore parameters from `<logdir>/<ckpt>` or the newest model.ckpt-N
    (wrapper.py:32-62); returns the global step parsed from the -N suffix."""
    path = find_ckpt(logdir, ckpt)
    if path is None:
        raise FileNotFoundError('no model.ckpt-N under %s' % logdir)
    sd = read_ckpt(path, getattr(engine, 'layout', None))
    engine.load_flat(sd['params'])
    m = re.search(r'-(\d+)$', path)
    return int(m.group(1)) if m else int(sd.get('step', 0))
