"""TensorFlow "V2" checkpoints (tensor bundles) without TensorFlow: reader, writer, and the mapping between the
reference's variables and the flat parameter buffer of include/vaenpvc.h.

The reference saves through `tf.train.Supervisor` / `tf.train.Saver` (trainer/vae.py:76-82, util/wrapper.py:10-62,
convert.py:100-103): `model.ckpt-N.index` + `model.ckpt-N.data-00000-of-00001` (+ `.meta`, `checkpoint`).

  * `<prefix>.index` is a LevelDB-format table (tensorflow/core/lib/io/table*.cc, written uncompressed by
    BundleWriter): data blocks of prefix-compressed (key, value) entries with a restart array, a metaindex block, an
    index block, and a 48-byte footer ending in the magic 0xdb4775248b80fb57; every block is followed by a 1-byte
    compression type and a masked CRC-32C.  Key "" holds a BundleHeaderProto (num_shards = 1, endianness = 0,
    version); every other key is a variable name whose value is a BundleEntryProto {dtype = 1, shape = 2,
    shard_id = 3, offset = 4, size = 5, crc32c = 6}.
  * `<prefix>.data-00000-of-00001` is the concatenation of the tensors' raw little-endian bytes.

PARITY UNPINNED: no TensorFlow and no reference checkpoint exist in this environment, so the reader is tested against
this file's own writer and against the format facts above (magic, CRCs, block layout), and the variable names are the
*expected* TF-1 names of SURVEY App. A.5 (they are exactly the names of `vaenpvc_param_info`).  Snappy-compressed
index blocks (not what BundleWriter produces) are rejected with a clear message.
"""
import os
import re
import struct

import numpy as np

from util.summary import _parse, _read_varint, _varint, masked_crc

MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8')}
_DT_OF = {np.dtype('float32'): DT_FLOAT, np.dtype('int32'): DT_INT32, np.dtype('int64'): DT_INT64}


# ---------------------------------------------------------------------------- table reader
def _read_block(buf, offset, size):
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    crc = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if masked_crc(bytes(data) + bytes([ctype])) != crc:
        raise ValueError('index block at %d: CRC mismatch' % offset)
    if ctype != 0:
        raise ValueError('index block at %d is compressed (type %d); BundleWriter writes uncompressed tables' % (offset, ctype))
    nrest = struct.unpack_from('<I', data, len(data) - 4)[0]
    end = len(data) - 4 - 4 * nrest
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(data, pos)
        non_shared, pos = _read_varint(data, pos)
        vlen, pos = _read_varint(data, pos)
        key = key[:shared] + bytes(data[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(data[pos:pos + vlen])))
        pos += vlen
    return out


def _handle(b, pos=0):
    off, pos = _read_varint(b, pos)
    size, pos = _read_varint(b, pos)
    return off, size, pos


def read_index(path):
    """{key bytes: value bytes} of a LevelDB-format table file."""
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow V2 checkpoint index (bad magic)' % path)
    footer = buf[-48:]
    _, _, p = _handle(footer)                 # metaindex handle
    ioff, isize, _ = _handle(footer, p)       # index handle
    entries = {}
    for _, hv in _read_block(buf, ioff, isize):
        boff, bsize, _ = _handle(hv)
        for k, v in _read_block(buf, boff, bsize):
            entries[k] = v
    return entries


def read_bundle(prefix):
    """All tensors of the checkpoint `<prefix>` -> {name: ndarray}."""
    entries = read_index(prefix + '.index')
    header = _parse(entries.get(b'', b''))
    nshards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
        raise ValueError('big-endian bundle')
    shards = [open('%s.data-%05d-of-%05d' % (prefix, i, nshards), 'rb').read() for i in range(nshards)]
    out = {}
    for k, v in entries.items():
        if k == b'':
            continue
        e = _parse(v, fixed32_int=True)
        dt = e.get(1, [0])[0]
        if 7 in e:
            raise ValueError('%s: sliced (partitioned) variables are not supported' % k.decode())
        if dt not in _DTYPES:
            raise ValueError('%s: unsupported dtype %d' % (k.decode(), dt))
        shape = []
        for sh in e.get(2, []):
            for d in _parse(sh).get(2, []):
                shape.append(_parse(d).get(1, [0])[0])
        off, size = e.get(4, [0])[0], e.get(5, [0])[0]
        raw = shards[e.get(3, [0])[0]][off:off + size]
        if 6 in e and masked_crc(raw) != e[6][0]:
            raise ValueError('%s: tensor CRC mismatch' % k.decode())
        out[k.decode()] = np.frombuffer(raw, _DTYPES[dt]).reshape(shape).copy()
    return out


# ---------------------------------------------------------------------------- writer
def _f_varint(field, v):
    return _varint((field << 3) | 0) + _varint(v)


def _f_bytes(field, b):
    return _varint((field << 3) | 2) + _varint(len(b)) + b


def _f_fixed32(field, v):
    return _varint((field << 3) | 5) + struct.pack('<I', v)


def _block(entries, restart_interval=16):
    """LevelDB block of sorted (key, value) pairs + trailer (type 0, masked CRC)."""
    body, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            n = min(len(k), len(last))
            while shared < n and k[shared] == last[shared]:
                shared += 1
        body += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack('<I', r)
    body += struct.pack('<I', len(restarts))
    body = bytes(body)
    return body + b'\x00' + struct.pack('<I', masked_crc(body + b'\x00'))


def write_bundle(prefix, tensors):
    """Writes {name: ndarray} as a single-shard V2 checkpoint (`<prefix>.index`, `<prefix>.data-00000-of-00001`)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    data, entries = bytearray(), []
    for name in sorted(tensors):
        a = np.asarray(tensors[name], order='C')
        if a.dtype not in _DT_OF:
            raise ValueError('%s: unsupported dtype %s' % (name, a.dtype))
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        shape = b''.join(_f_bytes(2, _f_varint(1, d)) for d in a.shape)
        e = (_f_varint(1, _DT_OF[a.dtype]) + _f_bytes(2, shape) + _f_varint(4, len(data)) + _f_varint(5, len(raw)) +
             _f_fixed32(6, masked_crc(raw)))
        entries.append((name.encode(), e))
        data += raw
    header = _f_varint(1, 1) + _f_bytes(3, _f_varint(1, 1))       # num_shards = 1, endianness little, version.producer = 1
    entries = [(b'', header)] + entries
    with open(prefix + '.data-00000-of-00001', 'wb') as fp:
        fp.write(bytes(data))
    out = bytearray()
    blk = _block(entries)
    data_handle = _varint(0) + _varint(len(blk) - 5)
    out += blk
    meta_off = len(out)
    meta = _block([])
    out += meta
    idx_off = len(out)
    idx = _block([(entries[-1][0] + b'\x00', data_handle)])
    out += idx
    footer = _varint(meta_off) + _varint(len(meta) - 5) + _varint(idx_off) + _varint(len(idx) - 5)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    out += footer
    with open(prefix + '.index', 'wb') as fp:
        fp.write(bytes(out))


# ---------------------------------------------------------------------------- reference variables <-> flat buffer
_ENC_CONV = re.compile(r'^(Encoder/Conv2d-(\d+))/(kernel|bias)$')


def tf_name(name):
    """Name of an engine-layout tensor in the reference's TF graph.  They differ only for the encoder convs:
    conv2d_nchw_layernorm (util/layers.py:55-64) opens tf.variable_scope(name) AND passes name=name to tf.layers.conv2d, so
    the conv variables are `Encoder/Conv2d-i/Conv2d-i/{kernel,bias}` while the LayerNorm pair sits one scope higher
    (`Encoder/Conv2d-i/layernorm.{offset,scale}`, util/layers.py:65)."""
    m = _ENC_CONV.match(name)
    return '%s/Conv2d-%s/%s' % (m.group(1), m.group(2), m.group(3)) if m else name


def import_checkpoint(prefix, layout):
    """TF variables of the reference's graph -> state dict of hipvae.dp.Stepper.  `layout` = Engine.layout
    ({name: (offset, shape)}, names = tf.trainable_variables() of model/vae.py).  Adam slots (`<name>/Adam`,
    `<name>/Adam_1`, trainer/vae.py:16-24) and `global_step` are taken when present."""
    import torch
    t = read_bundle(prefix)
    n = sum(int(np.prod(shape)) for _, shape in layout.values())
    flat = {k: np.zeros(n, np.float32) for k in ('params', 'm', 'v')}
    missing = [tf_name(name) for name in layout if tf_name(name) not in t]
    if missing:
        raise KeyError('checkpoint %s lacks %d variables of the ConvVAE, e.g. %s' % (prefix, len(missing), missing[:3]))
    slots = 0
    for name, (off, shape) in layout.items():
        k = int(np.prod(shape))
        tn = tf_name(name)
        for dst, key in (('params', tn), ('m', tn + '/Adam'), ('v', tn + '/Adam_1')):
            if key in t:
                a = np.asarray(t[key])
                # the SHAPE must match, not only the element count: a transposed kernel would load silently otherwise
                if tuple(a.shape) != tuple(shape):
                    raise ValueError('%s: checkpoint shape %s, model shape %s' % (key, tuple(a.shape), tuple(shape)))
                flat[dst][off:off + k] = a.astype(np.float32).reshape(-1)
                slots += dst != 'params'
    m = re.search(r'-(\d+)$', prefix)
    step = int(t['global_step']) if 'global_step' in t else (int(m.group(1)) if m else 0)
    if slots == 0 and step != 0:
        # parameters only (e.g. a Saver over tf.trainable_variables()): bias-correcting zero moments for t = step would
        # shrink the first updates by sqrt(1 - b2^t) / (1 - b1^t) of a run that never happened -- restart the optimiser
        import warnings
        warnings.warn('checkpoint %s holds no Adam slots: optimiser state reset (step %d -> 0)' % (prefix, step))
        step = 0
    elif 0 < slots < 2 * len(layout):
        raise KeyError('checkpoint %s holds Adam slots for only %d of %d tensors' % (prefix, slots // 2, len(layout)))
    return {'params': torch.from_numpy(flat['params']), 'm': torch.from_numpy(flat['m']), 'v': torch.from_numpy(flat['v']),
            'step': step}


def export_checkpoint(prefix, layout, state, lr_betas=(0.5, 0.999)):
    """The reverse: a Stepper state dict as the variables `tf.train.Saver` of the reference's graph would restore."""
    tensors = {}
    p, m, v = (np.asarray(state[k], np.float32).reshape(-1) for k in ('params', 'm', 'v'))
    for name, (off, shape) in layout.items():
        k = int(np.prod(shape))
        tn = tf_name(name)
        tensors[tn] = p[off:off + k].reshape(shape)
        tensors[tn + '/Adam'] = m[off:off + k].reshape(shape)
        tensors[tn + '/Adam_1'] = v[off:off + k].reshape(shape)
    step = int(state.get('step', 0))
    tensors['global_step'] = np.array(step, np.int32)      # tf.Variable(0, name='global_step') is int32 (trainer/vae.py:15)
    tensors['beta1_power'] = np.array(lr_betas[0] ** (step + 1), np.float32)     # tf.train.AdamOptimizer non-slot variables
    tensors['beta2_power'] = np.array(lr_betas[1] ** (step + 1), np.float32)
    write_bundle(prefix, tensors)
