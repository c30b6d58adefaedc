"""Critic: owns the discriminator's flat parameter buffer and workspace, calls the vaenpvc_disc_* entry points
(include/vaenpvc.h).  The counterpart of the 'Discriminator' template of the VAWGAN branch
(trainer/vae.py:128-130 selects its variables by that name)."""
import ctypes as C
from collections import OrderedDict

import torch

from . import lib as L
from .engine import glorot_init


def disc_arch_to_struct(arch):
    """architecture-vawgan-*.json -> vaenpvc_disc_arch (keys hwc, discriminator.{kernel,stride,output})."""
    d = arch['discriminator']
    assert len(d['output']) == len(d['kernel']) == len(d['stride'])
    if len(d['output']) > L.MAX_LAYERS:
        raise ValueError('at most %d layers' % L.MAX_LAYERS)
    a = L.DiscArch()
    a.H = int(arch['hwc'][0])
    a.n_layers = len(d['output'])
    for i, (o, k, s) in enumerate(zip(d['output'], d['kernel'], d['stride'])):
        if int(k[1]) != 1 or int(s[1]) != 1:
            raise ValueError('kernels/strides must be [k, 1]')
        a.output[i], a.kernel[i], a.stride[i] = int(o), int(k[0]), int(s[0])
    return a


class Critic(object):
    def __init__(self, arch, device=None):
        self.lib = L.load_library()
        if not torch.cuda.is_available():
            raise L.HipVaeError('no GPU visible: the critic has no CPU implementation')
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self._astruct = disc_arch_to_struct(arch)
        h = C.c_void_p()
        L.check(self.lib.vaenpvc_disc_create(C.byref(self._astruct), C.byref(h)), 'disc_create')
        self.handle = h
        self.H = self._astruct.H
        self.layout = self._query_layout()
        self.n_params = int(self.lib.vaenpvc_disc_param_floats(self.handle))
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self._ws = None
        self._loss2 = torch.zeros(2, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.vaenpvc_disc_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _query_layout(self):
        out = OrderedDict()
        buf = C.create_string_buffer(128)
        off, nd = C.c_int64(), C.c_int32()
        shp = (C.c_int64 * 4)()
        for i in range(self.lib.vaenpvc_disc_param_count(self.handle)):
            L.check(self.lib.vaenpvc_disc_param_info(self.handle, i, buf, 128, C.byref(off), C.byref(nd), shp),
                    'disc_param_info')
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

    def _workspace(self, F):
        need = int(self.lib.vaenpvc_disc_workspace_bytes(self.handle, F))
        if need < 0:
            L.check(need, 'disc_workspace_bytes')
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _rows(self, x, F=None):
        if x.dtype != torch.float32 or not x.is_cuda:
            raise TypeError('expected a float32 CUDA tensor')
        x = x.reshape(x.shape[0], -1)
        if x.shape[1] != self.H or (F is not None and x.shape[0] != F):
            raise ValueError('expected [F, %d]' % self.H)
        return x.contiguous()

    def values(self, x, xh):
        """(D(x) | D(xh)) float32 [2F] and loss2 = {W_dist, 0}."""
        x = self._rows(x)
        F = x.shape[0]
        xh = self._rows(xh, F)
        ws, nb = self._workspace(F)
        out = torch.empty(2 * F, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.vaenpvc_disc_fwd(self.handle, self.params.data_ptr(), x.data_ptr(), xh.data_ptr(), F,
                                              out.data_ptr(), self._loss2.data_ptr(), ws.data_ptr(), nb,
                                              self._stream()), 'disc_fwd')
        return out, self._loss2

    def critic_fwd_bwd(self, x, xh, t, lam, grads, out=None):
        """d l_D / d critic parameters into `grads`; returns loss2 = {W_dist, gp}."""
        x = self._rows(x)
        F = x.shape[0]
        xh = self._rows(xh, F)
        if t.dtype != torch.float32 or not t.is_cuda or t.numel() != F:
            raise TypeError('t must be float32 CUDA [F]')
        if grads.dtype != torch.float32 or grads.numel() != self.n_params or not grads.is_cuda:
            raise TypeError('grads must be a flat float32 CUDA buffer of %d elements' % self.n_params)
        out = self._loss2 if out is None else out
        ws, nb = self._workspace(F)
        with torch.cuda.device(self.device):
            L.check(self.lib.vaenpvc_disc_critic_fwd_bwd(self.handle, self.params.data_ptr(), x.data_ptr(),
                                                         xh.data_ptr(), t.contiguous().data_ptr(), F, float(lam),
                                                         grads.data_ptr(), out.data_ptr(), ws.data_ptr(), nb,
                                                         self._stream()), 'disc_critic_fwd_bwd')
        return out

    def generator_target(self, x, xh, alpha, out=None):
        """x + alpha (1 + 1e-6) dD(xh)/dxh [F, H] (the `target` of Engine.train_fwd_bwd_target) and loss2."""
        x = self._rows(x)
        F = x.shape[0]
        xh = self._rows(xh, F)
        target = torch.empty(F, self.H, dtype=torch.float32, device=self.device)
        out = self._loss2 if out is None else out
        ws, nb = self._workspace(F)
        with torch.cuda.device(self.device):
            L.check(self.lib.vaenpvc_disc_generator_target(self.handle, self.params.data_ptr(), x.data_ptr(),
                                                           xh.data_ptr(), F, float(alpha), target.data_ptr(),
                                                           out.data_ptr(), ws.data_ptr(), nb, self._stream()),
                    'disc_generator_target')
        return target, out
