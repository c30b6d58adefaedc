"""Conversion CLI with the reference's flag names (convert.py:14-30):

    python convert.py --src SF1 --trg TM3 --model ConvVAE \
        --checkpoint logdir/train/<stamp>/model.ckpt-<N>

Device path per utterance (convert.py:79-89): Tanhize -> encode (z_mu) -> decode with the
target speaker id -> inverse Tanhize.  The log-F0 transform (convert.py:51-57) runs on the
host.  WORLD synthesis needs pyworld (absent here): when it is importable a wav is written,
otherwise the converted features are saved as `<src>-<trg>-<basename>.npz`.
"""
import argparse
import glob
import json
import os
import sys
from datetime import datetime
from importlib import import_module

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FS = 16000


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--checkpoint', default=None, help='root of log dir')
    p.add_argument('--src', default='SF1', help='source speaker [SF1 - TM3]')
    p.add_argument('--trg', default='TM3', help='target speaker [SF1 - TM3]')
    p.add_argument('--output_dir', default='./logdir', help='root of output dir')
    p.add_argument('--module', default='model.vae', help='Module')
    p.add_argument('--model', default=None, help='Model')
    p.add_argument('--file_pattern', default='./dataset/vcc2016/bin/Testing Set/{}/*.bin', help='file pattern')
    p.add_argument('--batch_frames', type=int, default=16384,
                   help='(not in the reference) frames gathered from consecutive utterances into one device launch; '
                        '0 = REFERENCE-PARITY mode: one launch per utterance, like the reference\'s sess.run per file -- an '
                        'utterance of <= 512 frames then runs on the fp32-exact whole-frame kernels and its output does not '
                        'depend on its neighbours in the glob order.  The default gathers files onto the large-batch kernels '
                        '(2-term bf16 operands: within the 1e-4 relative parity bar, not bit-reproducible per file)')
    args = p.parse_args(argv)
    if args.model is None:                                               # convert.py:23-27
        raise ValueError('\n  You MUST specify `model`.'
                         '\n    Use `python convert.py --help` to see applicable options.')
    return args


def make_output_name(output_dir, filename, src, trg, ext):              # convert.py:35-43
    basename = os.path.splitext(os.path.split(str(filename, 'utf8'))[-1])[0]
    print('Processing {}'.format(basename))
    return os.path.join(output_dir, '{}-{}-{}.{}'.format(src, trg, basename, ext))


def get_default_output(logdir_root):                                     # convert.py:45-49
    stamp = datetime.now().strftime('%m%d-%H%M-%S-%Y')
    logdir = os.path.join(logdir_root, 'output', stamp)
    print('Using default logdir: {}'.format(logdir))
    return logdir


def convert_f0(f0, src, trg, etc_dir='./etc'):
    """convert.py:51-57 -- note the thresholds apply to the TRANSFORMED value (quirk kept)."""
    mu_s, std_s = np.fromfile(os.path.join(etc_dir, '{}.npf'.format(src)), np.float32)
    mu_t, std_t = np.fromfile(os.path.join(etc_dir, '{}.npf'.format(trg)), np.float32)
    f0 = np.asarray(f0, np.float32)
    lf0 = np.where(f0 > 1., np.log(np.where(f0 > 1., f0, 1.)), f0).astype(np.float32)
    lf0 = np.where(lf0 > 1., (lf0 - mu_s) / std_s * std_t + mu_t, lf0).astype(np.float32)
    lf0 = np.where(lf0 > 1., np.exp(lf0), lf0).astype(np.float32)
    return lf0


def convert_utterance(machine, normalizer, sp, trg_id):
    """The device tensor path of convert.py:79-89 for one utterance: sp [N,513] -> converted sp."""
    import torch
    x = normalizer.forward_process(sp)                                    # [N,513] in [-1,1]
    x = x.view(x.shape[0], 1, x.shape[1], 1)                              # nh_to_nchw (convert.py:60-63)
    y_t = torch.full((x.shape[0],), int(trg_id), dtype=torch.int64, device=x.device)
    z = machine.encode(x)
    x_t = machine.decode(z, y_t)                                          # NHWC [N,513,1,1]
    x_t = x_t.reshape(x_t.shape[0], -1)                                   # tf.squeeze
    return normalizer.backward_process(x_t)


def convert_utterances(machine, normalizer, sps, trg_id):
    """The same tensor path for SEVERAL utterances in one launch.  The model is frame-wise (W = 1: every frame is an independent
    sample of the network, model/vae.py:72-103), so the frames of consecutive files can share one encode -> decode call and the
    result is cut back at the file boundaries.  One utterance is a few hundred to ~2 000 frames, a size at which a launch
    sequence is latency-bound (0.4 ms for 1 024 frames against 2.5 ms for 32 768); the reference runs one sess.run per file
    (convert.py:105-116).  Returns the converted sp of every utterance, in order."""
    import torch
    sps = [np.ascontiguousarray(sp, np.float32) for sp in sps]
    if not sps:
        return []
    if len(sps) == 1:
        return [convert_utterance(machine, normalizer, sps[0], trg_id)]
    out = convert_utterance(machine, normalizer, np.concatenate(sps, axis=0), trg_id)
    return list(torch.split(out, [sp.shape[0] for sp in sps], dim=0))


def batched(features, batch_frames):
    """Groups the utterance stream: consecutive feature dicts whose frame counts add up to at most `batch_frames` (an utterance
    longer than that goes alone; batch_frames <= 0: every utterance alone).  Order is preserved."""
    group, n = [], 0
    for feat in features:
        k = int(feat['sp'].shape[0])
        if group and (batch_frames <= 0 or n + k > batch_frames):
            yield group
            group, n = [], 0
        group.append(feat)
        n += k
    if group:
        yield group


def main(argv=None):
    from analyzer import read_whole_features, SPEAKERS, Tanhize, load_npf
    from util.wrapper import load

    args = parse_args(argv)
    MODEL = getattr(import_module(args.module), args.model)
    logdir, ckpt = os.path.split(args.checkpoint)
    arch_file = glob.glob(os.path.join(logdir, 'architecture*.json'))[0]  # should only be 1 file
    with open(arch_file) as fp:
        arch = json.load(fp)
    normalizer = Tanhize(xmax=load_npf('./etc/xmax.npf'), xmin=load_npf('./etc/xmin.npf'))
    machine = MODEL(arch)
    load(machine.engine, logdir, ckpt=ckpt)
    output_dir = get_default_output(args.output_dir)
    os.makedirs(output_dir, exist_ok=True)
    trg_id = SPEAKERS.index(args.trg)
    try:
        import pyworld  # noqa: F401  (absent in this image: the features are saved instead of a wav)
        import soundfile as sf
        have_world = True
    except ImportError:
        sf, have_world = None, False
    from analyzer import pw2wav
    machine.engine.validate_ids(_ids(machine, 1, trg_id))
    for group in batched(read_whole_features(args.file_pattern.format(args.src)), args.batch_frames):
        converted = convert_utterances(machine, normalizer, [feat['sp'] for feat in group], trg_id)
        for feat, sp_t in zip(group, converted):
            sp = sp_t.cpu().numpy()
            f0 = convert_f0(feat['f0'], args.src, args.trg)
            feat.update({'sp': sp, 'f0': f0})
            if have_world:
                y = pw2wav(feat)                                           # convert.py:110-112
                sf.write(make_output_name(output_dir, feat['filename'], args.src, args.trg, 'wav'), y, FS)
            else:
                np.savez(make_output_name(output_dir, feat['filename'], args.src, args.trg, 'npz'),
                         sp=sp, f0=f0, ap=feat['ap'], en=feat['en'])
    return output_dir


def _ids(machine, n, trg_id):
    import torch
    return torch.full((max(1, n),), int(trg_id), dtype=torch.int64, device=machine.engine.device)


if __name__ == '__main__':
    main()
