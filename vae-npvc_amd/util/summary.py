"""TensorBoard event files for the summaries the reference declares in model/vae.py:132-136:

    tf.summary.scalar('KL-div', D_KL); tf.summary.scalar('logPx', logPx)
    tf.summary.histogram('xh', xh);    tf.summary.histogram('x', x)

which its tf.train.Supervisor (trainer/vae.py:76-82) merges and appends to
`<logdir>/events.out.tfevents.*` every 120 s (Supervisor default save_summaries_secs).

No TensorFlow here, so the container format is written by hand: a TFRecord stream (length, masked
CRC-32C of the length, payload, masked CRC-32C of the payload) of `Event` protocol buffers
(wall_time=1 double, step=2 int64, file_version=3 string, summary=5 {value=1 {tag=1, simple_value=2
float, histo=5 {min=1, max=2, num=3, sum=4, sum_squares=5, bucket_limit=6 packed, bucket=7 packed}}}).
The histogram buckets are TensorFlow's default limits (+-1e-12 * 1.1^k ... 1e20, 0, +-DBL_MAX) and the
counting runs on the GPU (`vaenpvc_summary`); only the reduced numbers come to the host.
`read_events` parses a file back (tests, and a quick look without TensorBoard).
"""
import os
import socket
import struct
import time

import numpy as np

# ---------------------------------------------------------------------------- CRC-32C (Castagnoli), masked
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------- protobuf wire format (writer)
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_double(field, v):
    return _key(field, 1) + struct.pack('<d', float(v))


def _f_float(field, v):
    return _key(field, 5) + struct.pack('<f', float(v))


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field, b):
    return _key(field, 2) + _varint(len(b)) + b


def _f_packed_double(field, vals):
    return _f_bytes(field, struct.pack('<%dd' % len(vals), *[float(v) for v in vals]))


def default_bucket_limits():
    """TensorFlow's default histogram bucket limits (core/lib/histogram/histogram.cc: InitDefaultBuckets)."""
    pos = []
    v = 1.0e-12
    while v < 1.0e20:
        pos.append(v)
        v *= 1.1
    pos.append(np.finfo(np.float64).max)
    return np.array([-p for p in reversed(pos)] + [0.0] + pos, np.float64)


def histogram_proto(stats, counts, limits):
    """HistogramProto bytes from (min, max, sum, sum_squares), per-bucket counts (len(limits) + 1, bucket b =
    [limits[b-1], limits[b])) -- empty bucket runs are collapsed the way TensorFlow's EncodeToProto does."""
    counts = np.asarray(counts, np.float64)
    num = counts.sum()
    lim_out, cnt_out = [], []
    n = len(limits)
    # our last bucket (v >= limits[-1] = DBL_MAX) is always empty for finite data; TF has exactly len(limits) buckets
    i = 0
    while i < n:
        c = counts[i]
        if c > 0 or i == n - 1 or counts[i + 1] > 0:
            lim_out.append(limits[i])
            cnt_out.append(c)
        i += 1
    body = (_f_double(1, stats[0]) + _f_double(2, stats[1]) + _f_double(3, num) + _f_double(4, stats[2]) +
            _f_double(5, stats[3]) + _f_packed_double(6, lim_out) + _f_packed_double(7, cnt_out))
    return body


class EventWriter(object):
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self.fp = open(self.path, 'ab')
        self._record(_f_double(1, time.time()) + _f_bytes(3, b'brain.Event:2'))

    def _record(self, data):
        hdr = struct.pack('<Q', len(data))
        self.fp.write(hdr + struct.pack('<I', masked_crc(hdr)) + data + struct.pack('<I', masked_crc(data)))
        self.fp.flush()

    def add(self, step, scalars=(), histograms=()):
        """scalars: [(tag, value)]; histograms: [(tag, HistogramProto bytes)]."""
        summ = b''
        for tag, v in scalars:
            summ += _f_bytes(1, _f_bytes(1, tag.encode()) + _f_float(2, v))
        for tag, h in histograms:
            summ += _f_bytes(1, _f_bytes(1, tag.encode()) + _f_bytes(5, h))
        self._record(_f_double(1, time.time()) + _f_varint(2, step) + _f_bytes(5, summ))

    def close(self):
        self.fp.close()


class SummaryWriter(object):
    """What trainer.VAETrainer writes every `secs` seconds (rank 0): the four summaries of model/vae.py:132-136."""

    def __init__(self, logdir, engine, secs=120):
        import torch
        self.engine, self.secs = engine, secs
        self.events = EventWriter(logdir)
        self.limits = default_bucket_limits()
        # device copy of the limits in float32 (the data is float32; +-DBL_MAX become +-inf, i.e. "everything")
        lim32 = self.limits.astype(np.float32)
        self.d_limits = torch.tensor(lim32, device=engine.device)
        self.t_last = time.time()

    def due(self, now=None):
        return (time.time() if now is None else now) - self.t_last >= self.secs

    def write(self, step, loss3, x, xh):
        """loss3 = {G, D_KL, logP} (device or host), x / xh = float32 CUDA tensors of the last batch."""
        hs = []
        for tag, t in (('xh', xh), ('x', x)):
            stats, counts = self.engine.summary(t, self.d_limits)
            hs.append((tag, histogram_proto(stats.cpu().numpy(), counts.cpu().numpy(), self.limits)))
        l3 = [float(v) for v in loss3]
        self.events.add(step, scalars=[('KL-div', l3[1]), ('logPx', l3[2])], histograms=hs)
        self.t_last = time.time()


# ---------------------------------------------------------------------------- reader (tests / inspection)
def _read_varint(buf, pos):
    n, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _parse(buf, fixed32_int=False):
    """protobuf message bytes -> {field: [values]} (wire types 0, 1, 2, 5; fixed32 as float unless fixed32_int)."""
    out, pos = {}, 0
    while pos < len(buf):
        k, pos = _read_varint(buf, pos)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v = struct.unpack_from('<d', buf, pos)[0]
            pos += 8
        elif wire == 5:
            v = struct.unpack_from('<I' if fixed32_int else '<f', buf, pos)[0]
            pos += 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        else:
            raise ValueError('wire type %d' % wire)
        out.setdefault(field, []).append(v)
    return out


def read_events(path):
    """[{'step', 'wall_time', 'file_version', 'scalars': {tag: v}, 'histograms': {tag: {...}}}]; verifies both CRCs."""
    data = open(path, 'rb').read()
    pos, events = 0, []
    while pos < len(data):
        hdr = data[pos:pos + 8]
        n = struct.unpack('<Q', hdr)[0]
        assert struct.unpack_from('<I', data, pos + 8)[0] == masked_crc(hdr), 'length CRC'
        body = data[pos + 12:pos + 12 + n]
        assert struct.unpack_from('<I', data, pos + 12 + n)[0] == masked_crc(body), 'payload CRC'
        pos += 16 + n
        ev = _parse(body)
        rec = {'wall_time': ev.get(1, [0.0])[0], 'step': ev.get(2, [0])[0],
               'file_version': ev.get(3, [b''])[0].decode(), 'scalars': {}, 'histograms': {}}
        for s in ev.get(5, []):
            for val in _parse(s).get(1, []):
                v = _parse(val)
                tag = v[1][0].decode()
                if 2 in v:
                    rec['scalars'][tag] = v[2][0]
                if 5 in v:
                    h = _parse(v[5][0])
                    lim = np.frombuffer(h[6][0], '<f8') if 6 in h else np.zeros(0)
                    cnt = np.frombuffer(h[7][0], '<f8') if 7 in h else np.zeros(0)
                    rec['histograms'][tag] = {'min': h[1][0], 'max': h[2][0], 'num': h[3][0], 'sum': h[4][0],
                                              'sum_squares': h[5][0], 'bucket_limit': lim, 'bucket': cnt}
        events.append(rec)
    return events
