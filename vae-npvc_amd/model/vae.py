"""`--model_module model.vae --model ConvVAE` plugin (drop-in for the reference's
model/vae.py:8-145).  Same constructor and method names; the arithmetic runs in
libvaenpvc_hip.so (hand-written gfx950 HIP kernels), not in a TF graph.

Eager-host adaptation of the graph protocol: `loss(x, y)` accepts either
device tensors (evaluated now, eps ~ N(0,1) drawn on the device like
GaussianSampleLayer, util/layers.py:152-156) or the lazy (image, label) handles returned
by `analyzer.read` (evaluated by the trainer each step, like `sess.run`).
"""
import torch

from hipvae.engine import Engine


class LossDict(dict):
    """dict with keys 'G', 'D_KL', 'logP' (model/vae.py:127-130) plus the wiring the
    trainer needs in an eager host."""
    machine = None
    source = None


class ConvVAE(object):
    def __init__(self, arch, is_training=False, device=None, seed=None, impl=None):
        self.arch = arch
        self._sanity_check()
        self.is_training = is_training           # unused, kept like the reference (vae.py:14)
        self.engine = Engine(arch, device=device, impl=impl)
        self.engine.init_params(seed)
        self.generate = self.decode              # vae.py:34 (VAE-GAN extension alias)

    def _sanity_check(self):                     # vae.py:37-39
        for net in ['encoder', 'generator']:
            assert len(self.arch[net]['output']) == len(self.arch[net]['kernel']) == len(self.arch[net]['stride'])

    @property
    def y_emb(self):                             # vae.py:20-24
        return self.engine.param_views()['y_embedding/y_emb']

    def _draw_eps(self, F):
        return torch.randn(F, self.engine.z_dim, dtype=torch.float32, device=self.engine.device)

    def loss(self, x, y, eps=None):
        """vae.py:106-137 -> {'G': -logPx + D_KL, 'D_KL', 'logP'} (0-d device tensors),
        or a lazy LossDict when x/y are input-queue handles."""
        out = LossDict()
        out.machine = self
        if hasattr(x, 'source'):                 # lazy handles from analyzer.read
            out.source = x.source
            out.update({'G': None, 'D_KL': None, 'logP': None})
            return out
        F = x.shape[0]
        eps = self._draw_eps(F) if eps is None else eps
        l3 = self.engine.loss_fwd(x, y, eps).clone()
        out.update({'G': l3[0], 'D_KL': l3[1], 'logP': l3[2]})
        return out

    def encode(self, x):                         # vae.py:139-141 : z_mu only
        return self.engine.encode(x)

    def decode(self, z, y):                      # vae.py:143-145 : NHWC [F, H, 1, 1]
        xh = self.engine.decode(z, y)
        return xh.view(xh.shape[0], xh.shape[1], 1, 1)
