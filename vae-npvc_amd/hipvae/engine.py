"""Engine: owns the flat parameter buffer and the workspace, calls the C-ABI.

Mirrors what the TF graph + session own in the reference (model/vae.py variables,
trainer/vae.py optimizer slots).  One Engine per process / GPU.
"""
import ctypes as C
import math
from collections import OrderedDict

import torch

from . import lib as L


def arch_to_struct(arch):
    """architecture-*.json dict -> vaenpvc_arch (only the keys model/vae.py reads).
    Raises AssertionError like ConvVAE._sanity_check (model/vae.py:37-39)."""
    a = L.Arch()
    enc, gen = arch['encoder'], arch['generator']
    for net in (enc, gen):
        assert len(net['output']) == len(net['kernel']) == len(net['stride'])
    if len(enc['output']) > L.MAX_LAYERS or len(gen['output']) > L.MAX_LAYERS:
        raise ValueError('at most %d layers per net' % L.MAX_LAYERS)
    a.H = int(arch['hwc'][0])
    if int(arch['hwc'][1]) != 1 or int(arch['hwc'][2]) != 1:
        raise ValueError('frame-wise model: hwc must be [H, 1, 1]')
    a.z_dim, a.y_dim = int(arch['z_dim']), int(arch['y_dim'])
    a.n_enc = len(enc['output'])
    for i, (o, k, s) in enumerate(zip(enc['output'], enc['kernel'], enc['stride'])):
        if int(k[1]) != 1 or int(s[1]) != 1:
            raise ValueError('kernels/strides must be [k, 1]')
        a.enc_output[i], a.enc_kernel[i], a.enc_stride[i] = int(o), int(k[0]), int(s[0])
    gh, gw, gc = gen['hwc']
    if int(gw) != 1:
        raise ValueError('generator.hwc must be [h, 1, c]')
    a.gen_h, a.gen_c = int(gh), int(gc)
    a.n_dec = len(gen['output'])
    for i, (o, k, s) in enumerate(zip(gen['output'], gen['kernel'], gen['stride'])):
        if int(k[1]) != 1 or int(s[1]) != 1:
            raise ValueError('kernels/strides must be [k, 1]')
        a.dec_output[i], a.dec_kernel[i], a.dec_stride[i] = int(o), int(k[0]), int(s[0])
    return a


def glorot_init(layout, seed=None, device='cpu'):
    """TF1 default initialisers of the reference: Glorot-uniform kernels/embedding
    (tf.get_variable / tf.layers / slim defaults), zero biases and LN offsets, unit
    LN scales (util/layers.py:33-43).  Returns the flat float32 buffer."""
    gen = torch.Generator(device='cpu')
    if seed is not None:
        gen.manual_seed(int(seed))
    chunks = []
    for name, (off, shape) in layout.items():
        n = 1
        for s in shape:
            n *= s
        if name.endswith('.scale'):
            t = torch.ones(n)
        elif name.endswith('.offset') or name.endswith('bias') or name.endswith('biases'):
            t = torch.zeros(n)
        else:
            if len(shape) == 4:
                rf = shape[0] * shape[1]
                fi, fo = rf * shape[2], rf * shape[3]
            else:
                fi, fo = shape[0], shape[1]
            lim = math.sqrt(6.0 / (fi + fo))
            t = (torch.rand(n, generator=gen) * 2 - 1) * lim
        chunks.append(t.float())
    return torch.cat(chunks).to(device)


class Engine(object):
    def __init__(self, arch, device=None, impl=None, precision=None):
        self.lib = L.load_library()
        if not torch.cuda.is_available():
            raise L.HipVaeError('no GPU visible: the ConvVAE hot path has no CPU implementation')
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.arch = arch
        self._astruct = arch_to_struct(arch)
        ctx = C.c_void_p()
        L.check(self.lib.vaenpvc_ctx_create(C.byref(self._astruct), C.byref(ctx)), 'ctx_create')
        self.ctx = ctx
        if impl is not None:
            self.set_impl(impl)
        if precision is not None:
            self.set_precision(precision)
        self.layout = self._query_layout()
        self.n_params = int(self.lib.vaenpvc_param_floats(self.ctx))
        self.z_dim, self.H = self._astruct.z_dim, self._astruct.H
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self._ws = None
        self._loss3 = torch.zeros(3, dtype=torch.float32, device=self.device)
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._bucket_cb = None      # keeps the ctypes thunk alive while it is registered

    def __del__(self):
        try:
            if getattr(self, 'ctx', None):
                self.lib.vaenpvc_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def set_impl(self, impl):
        code = {'auto': L.IMPL_AUTO, 'generic': L.IMPL_GENERIC}.get(impl, impl)
        L.check(self.lib.vaenpvc_set_impl(self.ctx, int(code)), 'set_impl')

    def set_precision(self, precision):
        """'bf16x3' (fp32-exact), 'bf16x2' (default, 16 mantissa bits per operand) or 'bf16'."""
        code = L.PRECISIONS.get(precision, precision)
        L.check(self.lib.vaenpvc_set_precision(self.ctx, int(code)), 'set_precision')

    @property
    def precision(self):
        return int(self.lib.vaenpvc_get_precision(self.ctx))

    def set_tuned_masks(self, fwd=0xffffffff, bwd=0xffffffff):
        """Per-step tuned/generic kernel selection of THIS engine's context (developer hook, include/vaenpvc_debug.h)."""
        L.check(self.lib.vaenpvc_set_tuned_masks(self.ctx, fwd & 0xffffffff, bwd & 0xffffffff), 'set_tuned_masks')

    def timer_select(self, tag):
        L.check(self.lib.vaenpvc_timer_select(self.ctx, tag.encode() if tag else None), 'timer_select')

    def timer_read(self):
        ms, n = C.c_double(), C.c_int64()
        L.check(self.lib.vaenpvc_timer_read(self.ctx, C.byref(ms), C.byref(n)), 'timer_read')
        return ms.value, n.value

    def set_bucket_callback(self, fn):
        """fn(bucket, offset_floats, count_floats, ready_stream_ptr) is called during train_fwd_bwd as soon as
        a contiguous range of the flat gradient buffer is complete (hipvae.dp overlaps its all-reduce)."""
        if fn is None:
            L.check(self.lib.vaenpvc_set_bucket_callback(self.ctx, L.BUCKET_CB(), None), 'set_bucket_callback')
            self._bucket_cb = None
            return
        thunk = L.BUCKET_CB(lambda user, b, off, cnt, stream: fn(int(b), int(off), int(cnt), stream))
        L.check(self.lib.vaenpvc_set_bucket_callback(self.ctx, thunk, None), 'set_bucket_callback')
        self._bucket_cb = thunk

    def _query_layout(self):
        out = OrderedDict()
        n = self.lib.vaenpvc_param_count(self.ctx)
        buf = C.create_string_buffer(128)
        off, nd = C.c_int64(), C.c_int32()
        shp = (C.c_int64 * 4)()
        for i in range(n):
            L.check(self.lib.vaenpvc_param_info(self.ctx, i, buf, 128, C.byref(off), C.byref(nd), shp), 'param_info')
            out[buf.value.decode()] = (int(off.value), tuple(int(shp[k]) for k in range(nd.value)))
        return out

    def init_params(self, seed=None):
        self.params.copy_(glorot_init(self.layout, seed))

    def load_flat(self, flat):
        flat = torch.as_tensor(flat, dtype=torch.float32).reshape(-1)
        if flat.numel() != self.n_params:
            raise ValueError('expected %d parameters, got %d' % (self.n_params, flat.numel()))
        self.params.copy_(flat.to(self.device))

    def param_views(self, flat=None):
        flat = self.params if flat is None else flat
        out = OrderedDict()
        for name, (off, shape) in self.layout.items():
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off:off + n].view(*shape)
        return out

    # ------------------------------------------------------------------ workspace
    def _workspace(self, F, mode):
        need = int(self.lib.vaenpvc_workspace_bytes(self.ctx, F, mode))
        if need < 0:
            L.check(need, 'workspace_bytes')
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def ws_region(self, F, mode, name):
        """View of a named workspace region (tests / observability)."""
        off, cnt = C.c_int64(), C.c_int64()
        L.check(self.lib.vaenpvc_ws_find(self.ctx, F, mode, name.encode(), C.byref(off), C.byref(cnt)), 'ws_find')
        ws, _ = self._workspace(F, mode)
        return ws.view(torch.float32)[off.value:off.value + cnt.value]

    def _stream(self):
        # the current stream OF THIS ENGINE'S DEVICE (not of whatever device happens to be current)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Launches must happen with the engine's device current (HIP launches go to the current device)."""
        return torch.cuda.device(self.device)

    def _chk_x(self, x):
        if x.dtype != torch.float32 or not x.is_cuda:
            raise TypeError('x must be a float32 CUDA tensor')
        x = x.reshape(x.shape[0], -1)        # [F,1,H,1] NCHW or [F,H]: same memory
        if x.shape[1] != self.H:
            raise ValueError('x must have %d bins per frame' % self.H)
        return x.contiguous()

    def _chk_y(self, y, F):
        if y.dtype != torch.int64 or not y.is_cuda:
            raise TypeError('y must be an int64 CUDA tensor (analyzer.py:127)')
        if y.numel() != F:
            raise ValueError('y must have one speaker id per frame')
        return y.reshape(-1).contiguous()

    # ------------------------------------------------------------------ model ops
    def encode(self, x, want_lv=False):
        x = self._chk_x(x)
        F = x.shape[0]
        ws, nb = self._workspace(F, L.MODE_INFER)
        z_mu = torch.empty(F, self.z_dim, dtype=torch.float32, device=self.device)
        z_lv = torch.empty_like(z_mu) if want_lv else None
        with self._on_device():
            L.check(self.lib.vaenpvc_encode_fwd(self.ctx, self.params.data_ptr(), x.data_ptr(), F, z_mu.data_ptr(),
                                                z_lv.data_ptr() if want_lv else None, ws.data_ptr(), nb,
                                                self._stream()), 'encode_fwd')
        return (z_mu, z_lv) if want_lv else z_mu

    def decode(self, z, y):
        if z.dtype != torch.float32 or not z.is_cuda or z.dim() != 2 or z.shape[1] != self.z_dim:
            raise TypeError('z must be float32 CUDA [F, %d]' % self.z_dim)
        z = z.contiguous()
        F = z.shape[0]
        y = self._chk_y(y, F)
        ws, nb = self._workspace(F, L.MODE_INFER)
        xh = torch.empty(F, self.H, dtype=torch.float32, device=self.device)
        with self._on_device():
            L.check(self.lib.vaenpvc_decode_fwd(self.ctx, self.params.data_ptr(), z.data_ptr(), y.data_ptr(), F,
                                                xh.data_ptr(), ws.data_ptr(), nb, self._stream()), 'decode_fwd')
        return xh

    def _chk_eps(self, eps, F):
        if eps.dtype != torch.float32 or not eps.is_cuda or tuple(eps.shape) != (F, self.z_dim):
            raise TypeError('eps must be float32 CUDA [F, %d]' % self.z_dim)
        return eps.contiguous()

    def loss_fwd(self, x, y, eps=None, out=None, seed=None, offset=0):
        """{G, D_KL, logP}.  eps: injected N(0,1) draw [F, z]; or seed/offset: drawn on the device (Philox)."""
        x = self._chk_x(x)
        F = x.shape[0]
        y = self._chk_y(y, F)
        out = self._loss3 if out is None else out
        ws, nb = self._workspace(F, L.MODE_TRAIN)   # same buffer as training; INFER layout is a prefix
        with self._on_device():
            if eps is None:
                if seed is None:
                    raise TypeError('either eps or seed is required')
                L.check(self.lib.vaenpvc_loss_fwd_seeded(self.ctx, self.params.data_ptr(), x.data_ptr(), y.data_ptr(),
                                                         int(seed) & (2 ** 64 - 1), int(offset), F, out.data_ptr(),
                                                         ws.data_ptr(), nb, self._stream()), 'loss_fwd_seeded')
            else:
                eps = self._chk_eps(eps, F)
                L.check(self.lib.vaenpvc_loss_fwd(self.ctx, self.params.data_ptr(), x.data_ptr(), y.data_ptr(),
                                                  eps.data_ptr(), F, out.data_ptr(), ws.data_ptr(), nb,
                                                  self._stream()), 'loss_fwd')
        return out

    def train_fwd_bwd(self, x, y, eps, grads, out=None, seed=None, offset=0, d_offset=None):
        """Forward + backward.  eps: injected draw; eps=None with seed/offset: the sampler draws on the device
        (vaenpvc_train_fwd_bwd_seeded; the draw is readable afterwards as ws_region(F, MODE_TRAIN, 'eps'))."""
        x = self._chk_x(x)
        F = x.shape[0]
        y = self._chk_y(y, F)
        if grads.dtype != torch.float32 or grads.numel() != self.n_params or not grads.is_cuda:
            raise TypeError('grads must be a flat float32 CUDA buffer of %d elements' % self.n_params)
        out = self._loss3 if out is None else out
        ws, nb = self._workspace(F, L.MODE_TRAIN)
        with self._on_device():
            if eps is None:
                if seed is None:
                    raise TypeError('either eps or seed is required')
                L.check(self.lib.vaenpvc_train_fwd_bwd_seeded(self.ctx, self.params.data_ptr(), x.data_ptr(),
                                                              y.data_ptr(), int(seed) & (2 ** 64 - 1), int(offset),
                                                              d_offset.data_ptr() if d_offset is not None else None, F,
                                                              grads.data_ptr(), out.data_ptr(), ws.data_ptr(), nb,
                                                              self._stream()), 'train_fwd_bwd_seeded')
            else:
                eps = self._chk_eps(eps, F)
                L.check(self.lib.vaenpvc_train_fwd_bwd(self.ctx, self.params.data_ptr(), x.data_ptr(), y.data_ptr(),
                                                       eps.data_ptr(), F, grads.data_ptr(), out.data_ptr(),
                                                       ws.data_ptr(), nb, self._stream()), 'train_fwd_bwd')
        return out

    def train_bwd_target(self, x, y, eps, target, grads, out=None):
        """Backward pass only against `target`, on the activations the preceding train_fwd_bwd call of the SAME
        (x, y, eps) left in the workspace (vaenpvc_train_bwd_target)."""
        return self.train_fwd_bwd_target(x, y, eps, target, grads, out=out, _entry='vaenpvc_train_bwd_target')

    def train_fwd_bwd_target(self, x, y, eps, target, grads, out=None, _entry='vaenpvc_train_fwd_bwd_target'):
        """train_fwd_bwd with the log-density evaluated against `target` [F, H] (vaenpvc_train_fwd_bwd_target)."""
        x = self._chk_x(x)
        F = x.shape[0]
        y = self._chk_y(y, F)
        eps = self._chk_eps(eps, F)
        target = self._chk_x(target)
        if target.shape[0] != F:
            raise ValueError('target must have one row per frame')
        if grads.dtype != torch.float32 or grads.numel() != self.n_params or not grads.is_cuda:
            raise TypeError('grads must be a flat float32 CUDA buffer of %d elements' % self.n_params)
        out = self._loss3 if out is None else out
        ws, nb = self._workspace(F, L.MODE_TRAIN)
        with self._on_device():
            L.check(getattr(self.lib, _entry)(self.ctx, self.params.data_ptr(), x.data_ptr(), y.data_ptr(),
                                              eps.data_ptr(), target.data_ptr(), F, grads.data_ptr(),
                                              out.data_ptr(), ws.data_ptr(), nb, self._stream()), _entry)
        return out

    def philox_uniform(self, n, seed, offset=0):
        """float32 [n] U[0,1) draw for (seed, offset) (vaenpvc_philox_uniform)."""
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        with self._on_device():
            L.check(self.lib.vaenpvc_philox_uniform(int(seed) & (2 ** 64 - 1), int(offset), out.data_ptr(), n,
                                                    self._stream()), 'philox_uniform')
        return out

    def adam_range(self, params, grads, m, v, lo, hi, step, lr, beta1, beta2, eps=1e-8, grad_scale=1.0):
        """TF-Adam apply on elements [lo, hi) of four congruent flat buffers (a `var_list` that is a contiguous range
        of the table: trainer/vae.py:128-130 groups variables by name)."""
        with self._on_device():
            L.check(self.lib.vaenpvc_adam_step(params.data_ptr() + 4 * lo, grads.data_ptr() + 4 * lo,
                                               m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo, hi - lo, int(step),
                                               float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                               self._stream()), 'adam_step')

    def philox_normal(self, rows, seed, offset=0):
        """The N(0,1) tensor [rows, z_dim] the seeded entry points draw for (seed, offset)."""
        out = torch.empty(rows, self.z_dim, dtype=torch.float32, device=self.device)
        with self._on_device():
            L.check(self.lib.vaenpvc_philox_normal(int(seed) & (2 ** 64 - 1), int(offset), out.data_ptr(), out.numel(),
                                                   self._stream()), 'philox_normal')
        return out

    def validate_ids(self, y):
        """Raises HipVaeError if any speaker id is outside [0, y_dim) (TF raises on the CPU; the kernels clamp)."""
        y = self._chk_y(y, y.numel())
        with self._on_device():
            L.check(self.lib.vaenpvc_validate_ids(self.ctx, y.data_ptr(), y.numel(), self._flag.data_ptr(),
                                                  self._stream()), 'validate_ids')

    def adam_step(self, grads, m, v, step, lr, beta1, beta2, eps=1e-8, grad_scale=1.0):
        with self._on_device():
            L.check(self.lib.vaenpvc_adam_step(self.params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                               self.n_params, int(step), float(lr), float(beta1), float(beta2),
                                               float(eps), float(grad_scale), self._stream()), 'adam_step')

    def adam_step_dev(self, grads, m, v, d_step, lr, beta1, beta2, eps=1e-8, grad_scale=1.0):
        """Graph-capturable Adam: the int64 step counter `d_step` lives on the device."""
        with self._on_device():
            L.check(self.lib.vaenpvc_adam_step_dev(self.params.data_ptr(), grads.data_ptr(), m.data_ptr(),
                                                   v.data_ptr(), self.n_params, d_step.data_ptr(), float(lr),
                                                   float(beta1), float(beta2), float(eps), float(grad_scale),
                                                   self._stream()), 'adam_step_dev')

    # ------------------------------------------------------------------ data plane
    def tanhize(self, sp, xmin, xmax, forward=True):
        sp = sp.contiguous()
        out = torch.empty_like(sp)
        fn = self.lib.vaenpvc_tanhize_fwd if forward else self.lib.vaenpvc_tanhize_bwd
        with self._on_device():
            L.check(fn(sp.data_ptr(), xmin.data_ptr(), xmax.data_ptr(), out.data_ptr(), sp.shape[0], sp.shape[1],
                       self._stream()), 'tanhize')
        return out

    def unpack_records(self, rec, xmin, xmax, index=None):
        """x = Tanhize(rec[i, :H]), y = int64(rec[i, -1]) for i in `index` (int64 CUDA tensor; default: all rows).
        The gather runs inside the kernel (analyzer.py:113-135 dequeue + slicing in one pass)."""
        rec = rec.contiguous()
        N, R = rec.shape
        F = N if index is None else int(index.numel())
        x = torch.empty(F, self.H, dtype=torch.float32, device=self.device)
        y = torch.empty(F, dtype=torch.int64, device=self.device)
        with self._on_device():
            if index is None:
                L.check(self.lib.vaenpvc_unpack_records(rec.data_ptr(), F, R, self.H, xmin.data_ptr(), xmax.data_ptr(),
                                                        x.data_ptr(), y.data_ptr(), self._stream()), 'unpack_records')
            else:
                if index.dtype != torch.int64 or not index.is_cuda:
                    raise TypeError('index must be an int64 CUDA tensor')
                index = index.contiguous()
                L.check(self.lib.vaenpvc_gather_unpack_records(rec.data_ptr(), N, index.data_ptr(), F, R, self.H,
                                                               xmin.data_ptr(), xmax.data_ptr(), x.data_ptr(),
                                                               y.data_ptr(), self._stream()), 'gather_unpack_records')
        return x, y

    def summary(self, data, edges):
        """tf.summary.histogram payload of a float32 CUDA tensor over ascending bucket limits `edges` (float32
        CUDA, <= 2048): returns (stats float64[4] = min, max, sum, sum of squares; counts int64[len(edges)+1])."""
        data = data.contiguous().view(-1)
        stats = torch.tensor([float('inf'), float('-inf'), 0.0, 0.0], dtype=torch.float64, device=self.device)
        counts = torch.zeros(edges.numel() + 1, dtype=torch.int64, device=self.device)
        with self._on_device():
            L.check(self.lib.vaenpvc_summary(data.data_ptr(), data.numel(), edges.data_ptr(), edges.numel(),
                                             stats.data_ptr(), counts.data_ptr(), self._stream()), 'summary')
        return stats, counts
