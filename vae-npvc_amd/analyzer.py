"""Data plane of the hot path (counterpart of the reference's analyzer.py:16-22,75-158).

The TF queue machinery is replaced by an HBM-resident frame store: every matching
`.bin` file (records of 1029 float32 = [sp(513) | ap(513) | f0 | en | speaker]) is
uploaded once; a batch is ONE HIP kernel that gathers the shuffled record rows' sp columns,
applies Tanhize and casts the speaker column to int64 (`vaenpvc_gather_unpack_records`); the shuffle
order comes from a host-side BoundedShuffler with the reference queue's capacity /
min_after_dequeue semantics.  WORLD feature
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
        with torch.cuda.device(x2.device):
            L.check(fn(x2.data_ptr(), self.xmin.data_ptr(), self.xmax.data_ptr(), out.data_ptr(), x2.shape[0],
                       x2.shape[1], torch.cuda.current_stream(x2.device).cuda_stream), 'tanhize')
        return out.reshape(shape)

    def forward_process(self, x):
        return self._run(x, True)

    def backward_process(self, x):
        return self._run(x, False)


class _Handle(object):
    """Lazy stand-in for a dequeued graph tensor."""
    def __init__(self, source, which):
        self.source, self.which = source, which


class BoundedShuffler(object):
    """Host-side index stream with the semantics of the reference's input queues
    (analyzer.py:103-135): `tf.train.string_input_producer(files)` reshuffles the FILE order every
    epoch and a FixedLengthRecordReader walks each file's records in order; `tf.train.shuffle_batch`
    keeps a RandomShuffleQueue of at most `capacity` records and dequeues uniformly at random while at
    least `min_after_dequeue` remain.  A record can therefore only move a bounded distance away from its
    file-order position: mixing is LOCAL (capacity 2048 / min_after_dequeue 1024 in main.py:65-66).

    This class produces the same kind of sequence on record NUMBERS (the records themselves stay in
    HBM): a pool is refilled to `capacity` from the epoch stream before every draw and
    min(n, len(pool) - min_after_dequeue) distinct entries leave it uniformly at random per draw.
    Deterministic given `seed`; every data-parallel rank builds the same sequence and slices it."""

    def __init__(self, file_sizes, capacity, min_after_dequeue, seed=0):
        if capacity <= min_after_dequeue:
            raise ValueError('capacity must be larger than min_after_dequeue')   # tf.train.shuffle_batch's own check
        self.sizes = [int(n) for n in file_sizes]
        self.starts = np.concatenate([[0], np.cumsum(self.sizes)[:-1]]).astype(np.int64)
        self.capacity, self.min_after = int(capacity), int(min_after_dequeue)
        ss = np.random.SeedSequence(seed).spawn(2)        # independent streams: file order / picks from the pool
        self.rng_files = np.random.Generator(np.random.PCG64(ss[0]))
        self.rng = np.random.Generator(np.random.PCG64(ss[1]))
        self.pool = np.empty(0, np.int64)
        self._pending = np.empty(0, np.int64)

    def _more(self, n):
        """next n record numbers of the (endless) epoch stream"""
        while self._pending.size < n:
            order = self.rng_files.permutation(len(self.sizes))       # string_input_producer(shuffle=True)
            epoch = np.concatenate([self.starts[i] + np.arange(self.sizes[i], dtype=np.int64) for i in order])
            self._pending = np.concatenate([self._pending, epoch])
        out, self._pending = self._pending[:n], self._pending[n:]
        return out

    def next(self, n):
        out = []
        while n > 0:
            if self.pool.size < self.capacity:
                self.pool = np.concatenate([self.pool, self._more(self.capacity - self.pool.size)])
            k = min(n, self.pool.size - self.min_after)
            pick = self.rng.choice(self.pool.size, size=k, replace=False)
            out.append(self.pool[pick])
            keep = np.ones(self.pool.size, bool)
            keep[pick] = False
            self.pool = self.pool[keep]
            n -= k
        return np.concatenate(out)

    def state_dict(self):
        """Everything the next draw depends on: a restored run continues the SAME record sequence instead of replaying
        the first batches of the original run (the reference's queues are not checkpointed either, but they are not
        seeded: a restarted TF run sees fresh shuffles, never the same ones again)."""
        # (builtin types and tensors only: the checkpoint stays loadable with torch.load's weights_only default)
        return {'rng_files': self.rng_files.bit_generator.state, 'rng': self.rng.bit_generator.state,
                'pool': torch.from_numpy(self.pool.copy()), 'pending': torch.from_numpy(self._pending.copy()),
                'sizes': list(self.sizes)}

    def load_state_dict(self, sd):
        if list(sd['sizes']) != self.sizes:
            raise ValueError('shuffler state belongs to another file list')
        self.rng_files.bit_generator.state = sd['rng_files']
        self.rng.bit_generator.state = sd['rng']
        self.pool = np.asarray(sd['pool'], dtype=np.int64).copy()
        self._pending = np.asarray(sd['pending'], dtype=np.int64).copy()


class IndexAhead(object):
    """The shuffler's index stream drawn AHEAD batches at a time: the device copy of the indices is then one transfer per
    AHEAD iterations instead of one per iteration (a small pageable host-to-device copy waits for the device's queue to
    drain: per iteration it serialised the host loop with the device, 37 us of a 0.23 ms step at batch 16).  The sequence
    is exactly the one of per-iteration draws (`shuffler.next(n)` is called once per batch either way); the state that is
    checkpointed is the shuffler's state BEFORE the block in flight plus the number of its batches already handed out."""

    AHEAD = 32

    def __init__(self, shuffler, n_per_batch, lo, hi):
        self.shuffler, self.n, self.lo, self.hi = shuffler, int(n_per_batch), int(lo), int(hi)
        self._block, self._pos, self._state0 = None, 0, None

    def _refill(self):
        self._state0 = self.shuffler.state_dict()
        self._block = np.stack([np.ascontiguousarray(self.shuffler.next(self.n)[self.lo:self.hi]) for _ in range(self.AHEAD)])
        self._pos = 0
        return self._block

    def next_block_if_due(self):
        """-> the new [AHEAD, hi - lo] index block when the current one is used up (the caller uploads it), else None"""
        if self._block is None or self._pos >= self.AHEAD:
            return self._refill()
        return None

    def take(self):
        """position of the next batch inside the current block"""
        p = self._pos
        self._pos += 1
        return p

    def state_dict(self):
        if self._block is None:
            return {'shuffler': self.shuffler.state_dict(), 'consumed': 0}
        return {'shuffler': self._state0, 'consumed': self._pos}

    def load_state_dict(self, sd):
        self.shuffler.load_state_dict(sd['shuffler'])
        self._block, self._pos, self._state0 = None, 0, None
        consumed = int(sd.get('consumed', 0))
        if consumed:
            self._refill()
            self._pos = consumed


class FrameStore(object):
    """All records resident in HBM; `next_batch()` = the shuffle_batch dequeue (analyzer.py:128-135): the
    BoundedShuffler picks record numbers on the host, ONE HIP kernel gathers those records' sp columns,
    normalises them and casts the speaker column (vaenpvc_gather_unpack_records).  With data parallelism
    every rank draws the same global index sequence (shared seed) and takes its own slice."""

    def __init__(self, records, batch_size, normalizer, seed=0, rank=0, world=1, device='cuda',
                 capacity=None, min_after_dequeue=None, file_sizes=None, y_dim=None):
        self.rec = torch.as_tensor(records, dtype=torch.float32).to(device).contiguous()
        if self.rec.dim() != 2 or self.rec.shape[1] != FEAT_DIM:
            raise ValueError('records must be [N, %d] float32' % FEAT_DIM)
        n = self.rec.shape[0]
        self.batch_size, self.normalizer = int(batch_size), normalizer
        self.rank, self.world = rank, world
        sizes = [n] if file_sizes is None else list(file_sizes)
        assert sum(sizes) == n
        capacity = min(int(capacity), n) if capacity else n
        min_after = min(int(min_after_dequeue), capacity - 1) if min_after_dequeue is not None else 0
        self.shuffler = BoundedShuffler(sizes, capacity, max(0, min_after), seed=seed)
        self.ahead = IndexAhead(self.shuffler, self.batch_size * world, rank * self.batch_size, (rank + 1) * self.batch_size)
        self._idx_dev = None
        self.lib = L.load_library()
        # speaker ids index the embedding table: an id outside [0, y_dim) is an error in TensorFlow; check the
        # (integral float) speaker column once, here, instead of on every batch
        spk = self.rec[:, -1]
        ny = len(SPEAKERS) if y_dim is None else int(y_dim)
        bad = int(((spk < 0) | (spk >= ny) | (spk != spk.floor())).sum().item())
        if bad:
            raise ValueError('%d record(s) carry a speaker id outside [0, %d)' % (bad, ny))

    def state_dict(self):
        a = self.ahead.state_dict()
        return {'shuffler': a['shuffler'], 'consumed': a['consumed'], 'batch_size': self.batch_size, 'world': self.world}

    def load_state_dict(self, sd):
        if (sd.get('batch_size'), sd.get('world')) != (self.batch_size, self.world):
            # another global batch: the draw sequence cannot line up; keep the fresh stream
            return False
        self.ahead.load_state_dict(sd)
        self._idx_dev = None if self.ahead._block is None else torch.from_numpy(self.ahead._block).to(self.rec.device)
        return True

    def next_batch(self):
        dev = self.rec.device
        blk = self.ahead.next_block_if_due()
        if blk is not None:
            self._idx_dev = torch.from_numpy(blk).to(dev)       # [AHEAD, batch] int64: one transfer per AHEAD iterations
        idx = self._idx_dev[self.ahead.take()]
        F = idx.numel()
        x = torch.empty(F, SP_DIM, dtype=torch.float32, device=dev)
        y = torch.empty(F, dtype=torch.int64, device=dev)
        nz = self.normalizer
        with torch.cuda.device(dev):
            L.check(self.lib.vaenpvc_gather_unpack_records(self.rec.data_ptr(), self.rec.shape[0], idx.data_ptr(), F,
                                                           FEAT_DIM, SP_DIM, nz.xmin.data_ptr(), nz.xmax.data_ptr(),
                                                           x.data_ptr(), y.data_ptr(),
                                                           torch.cuda.current_stream(dev).cuda_stream),
                    'gather_unpack_records')
        return x.view(F, 1, SP_DIM, 1), y        # NCHW [F,1,513,1] (analyzer.py:121-122)


def read(file_pattern, batch_size, record_bytes=RECORD_BYTES, capacity=256, min_after_dequeue=128,
         num_threads=8, format='NCHW', normalizer=None, seed=0, rank=0, world=1):
    """analyzer.py:90-135 signature; returns lazy (feature, speaker) handles.  `capacity` and
    `min_after_dequeue` bound the shuffle exactly like the reference's RandomShuffleQueue (BoundedShuffler);
    `num_threads` is meaningless here (no reader threads: the records live in HBM)."""
    if record_bytes != RECORD_BYTES:
        raise ValueError('records are %d bytes' % RECORD_BYTES)
    patterns = [file_pattern] if isinstance(file_pattern, str) else list(file_pattern)   # (the VAWGAN file lists two)
    files = sorted({f for p in patterns for f in glob.glob(p)})
    if not files:
        raise FileNotFoundError('no files match %r' % (file_pattern,))
    recs = [np.fromfile(f, '<f4').reshape(-1, FEAT_DIM) for f in files]
    store = FrameStore(np.concatenate(recs, 0), batch_size, normalizer, seed=seed, rank=rank, world=world,
                       capacity=capacity, min_after_dequeue=min_after_dequeue, file_sizes=[len(r) for r in recs])
    return _Handle(store, 'feature'), _Handle(store, 'speaker')


def write_bin(path, sp, ap, f0, en, speaker):
    """One utterance in the reference's on-disk format (analyzer.py:39-47,62-72): rows of 1029
    little-endian float32 = [sp(513) log10 energy-normalised | ap(513) | f0 | en | speaker id]."""
    sp, ap = np.asarray(sp, np.float32), np.asarray(ap, np.float32)
    n = sp.shape[0]
    if sp.shape != (n, SP_DIM) or ap.shape != (n, SP_DIM):
        raise ValueError('sp and ap must be [N, %d]' % SP_DIM)
    spk = SPEAKERS.index(speaker) if isinstance(speaker, str) else int(speaker)
    rows = np.concatenate([sp, ap, np.asarray(f0, np.float32).reshape(n, 1), np.asarray(en, np.float32).reshape(n, 1),
                           np.full((n, 1), spk, np.float32)], axis=1).astype('<f4')
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'wb') as fp:
        fp.write(rows.tobytes())
    return rows


def pw2wav_inputs(features, feat_dim=SP_DIM):
    """The arrays analyzer.pw2wav (analyzer.py:160-185) hands to pyworld.synthesize(f0, sp, ap, fs): float64,
    C-contiguous, sp de-normalised to the linear spectrum 10^sp * en.  Like the reference, the dict form
    (what convert.py:105-112 passes) does the 10^sp * en arithmetic in the arrays' OWN dtype (float32 there)
    and casts afterwards; the matrix form casts to float64 first.  Returns (f0, sp, ap)."""
    if isinstance(features, dict):
        en = np.reshape(features['en'], [-1, 1])
        sp = en * np.power(10., features['sp'])
        f0, ap = np.asarray(features['f0']), np.asarray(features['ap'])
    else:
        features = np.asarray(features).astype(np.float64)
        ap = features[:, feat_dim:feat_dim * 2]
        f0 = features[:, feat_dim * 2]
        en = np.reshape(features[:, feat_dim * 2 + 1], [-1, 1])
        sp = en * np.power(10., features[:, :feat_dim])
    return (f0.astype(np.float64).copy(order='C'), sp.astype(np.float64).copy(order='C'),
            ap.astype(np.float64).copy(order='C'))


def pw2wav(features, feat_dim=SP_DIM, fs=16000):
    """analyzer.py:160-185.  WORLD synthesis itself is the external pyworld C library (not in this image)."""
    import pyworld as pw
    f0, sp, ap = pw2wav_inputs(features, feat_dim)
    return pw.synthesize(f0, sp, ap, fs)


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
