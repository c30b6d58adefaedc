"""Data plane of the hot path (counterpart of the reference's analyzer.py:16-22,75-158).

The TF queue machinery is replaced by an HBM-resident frame store: every matching
`.bin` file (records of 1029 float32 = [sp(513) | ap(513) | f0 | en | speaker]) is
uploaded once; a batch is a device-side gather of shuffled record rows followed by the
`unpack_records` HIP kernel (slice sp, Tanhize, int64 speaker cast).  WORLD feature
extraction / synthesis (pyworld) is out of scope.
"""
import glob
import os

import numpy as np
import torch

from hipvae import lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, 'etc', 'speakers.tsv')) as _fp:
    SPEAKERS = [s.strip() for s in _fp.readlines() if s.strip()]      # analyzer.py:18
FFT_SIZE = 1024
SP_DIM = FFT_SIZE // 2 + 1
FEAT_DIM = SP_DIM + SP_DIM + 1 + 1 + 1      # [sp, ap, f0, en, s]  (analyzer.py:21)
RECORD_BYTES = FEAT_DIM * 4


def load_npf(path, dtype=np.float32, count=SP_DIM):
    """etc/xmin.npf / xmax.npf.  The reference reads them with the NumPy default dtype
    (float64) while build.py's percentile output dtype depends on the NumPy version
    (trap T4); here the on-disk dtype is explicit: float32, or float64 if the file size
    says so."""
    nbytes = os.path.getsize(path)
    if nbytes == count * 8:
        return np.fromfile(path, np.float64).astype(np.float32)
    if nbytes != count * 4:
        raise ValueError('%s: expected %d float32 values' % (path, count))
    return np.fromfile(path, dtype)


class Tanhize(object):
    """Normalizing `x` to [-1, 1] (analyzer.py:75-87) on the GPU."""

    def __init__(self, xmin, xmax, device=None):
        dev = device if device is not None else 'cuda'
        self.xmin = torch.as_tensor(np.asarray(xmin, np.float32)).to(dev).contiguous()
        self.xmax = torch.as_tensor(np.asarray(xmax, np.float32)).to(dev).contiguous()
        self.xscale = self.xmax - self.xmin

    def _run(self, x, fwd):
        lib = L.load_library()
        x = torch.as_tensor(x, dtype=torch.float32).to(self.xmin.device)
        shape = x.shape
        x2 = x.reshape(-1, self.xmin.numel()).contiguous()
        out = torch.empty_like(x2)
        fn = lib.vaenpvc_tanhize_fwd if fwd else lib.vaenpvc_tanhize_bwd
        L.check(fn(x2.data_ptr(), self.xmin.data_ptr(), self.xmax.data_ptr(), out.data_ptr(), x2.shape[0],
                   x2.shape[1], torch.cuda.current_stream().cuda_stream), 'tanhize')
        return out.reshape(shape)

    def forward_process(self, x):
        return self._run(x, True)

    def backward_process(self, x):
        return self._run(x, False)


class _Handle(object):
    """Lazy stand-in for a dequeued graph tensor."""
    def __init__(self, source, which):
        self.source, self.which = source, which


class FrameStore(object):
    """All records resident in HBM; `next_batch()` = shuffle_batch dequeue
    (analyzer.py:128-135) without replacement within an epoch.  With data parallelism
    every rank draws the same permutation (shared seed) and takes its own slice."""

    def __init__(self, records, batch_size, normalizer, seed=0, rank=0, world=1, device='cuda'):
        self.rec = torch.as_tensor(records, dtype=torch.float32).to(device).contiguous()
        if self.rec.dim() != 2 or self.rec.shape[1] != FEAT_DIM:
            raise ValueError('records must be [N, %d] float32' % FEAT_DIM)
        self.batch_size, self.normalizer = int(batch_size), normalizer
        self.rank, self.world = rank, world
        self.gen = torch.Generator(device='cpu')
        self.gen.manual_seed(seed)
        self.perm, self.pos = None, 0
        self.lib = L.load_library()

    def _indices(self):
        n, b = self.rec.shape[0], self.batch_size * self.world
        if self.perm is None or self.pos + b > n:
            self.perm = torch.randperm(n, generator=self.gen)
            self.pos = 0
            if b > n:
                raise ValueError('global batch larger than the data set')
        idx = self.perm[self.pos:self.pos + b]
        self.pos += b
        return idx[self.rank * self.batch_size:(self.rank + 1) * self.batch_size]

    def next_batch(self):
        idx = self._indices().to(self.rec.device)
        rows = self.rec.index_select(0, idx)
        F = rows.shape[0]
        x = torch.empty(F, SP_DIM, dtype=torch.float32, device=rows.device)
        y = torch.empty(F, dtype=torch.int64, device=rows.device)
        nz = self.normalizer
        L.check(self.lib.vaenpvc_unpack_records(rows.data_ptr(), F, FEAT_DIM, SP_DIM, nz.xmin.data_ptr(),
                                                nz.xmax.data_ptr(), x.data_ptr(), y.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), 'unpack_records')
        return x.view(F, 1, SP_DIM, 1), y        # NCHW [F,1,513,1] (analyzer.py:121-122)


def read(file_pattern, batch_size, record_bytes=RECORD_BYTES, capacity=256, min_after_dequeue=128,
         num_threads=8, format='NCHW', normalizer=None, seed=0, rank=0, world=1):
    """analyzer.py:90-135 signature; returns lazy (feature, speaker) handles."""
    files = sorted(glob.glob(file_pattern))
    if not files:
        raise FileNotFoundError('no files match %r' % file_pattern)
    recs = [np.fromfile(f, '<f4').reshape(-1, FEAT_DIM) for f in files]
    store = FrameStore(np.concatenate(recs, 0), batch_size, normalizer, seed=seed, rank=rank, world=world)
    return _Handle(store, 'feature'), _Handle(store, 'speaker')


def read_whole_features(file_pattern, num_epochs=1):
    """analyzer.py:138-158: one dict per utterance file."""
    files = sorted(glob.glob(file_pattern))
    print('{} files found'.format(len(files)))
    for _ in range(num_epochs):
        for f in files:
            print('Processing {}'.format(f), flush=True)
            v = np.fromfile(f, '<f4').reshape(-1, FEAT_DIM)
            yield {
                'sp': v[:, :SP_DIM],
                'ap': v[:, SP_DIM:2 * SP_DIM],
                'f0': v[:, SP_DIM * 2],
                'en': v[:, SP_DIM * 2 + 1],
                'speaker': v[:, SP_DIM * 2 + 2].astype(np.int64),
                'filename': f.encode('utf8'),
            }
