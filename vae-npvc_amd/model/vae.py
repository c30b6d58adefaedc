"""`--model_module model.vae --model ConvVAE` plugin (drop-in for the reference's
model/vae.py:8-145).  Same constructor and method names; the arithmetic runs in
libvaenpvc_hip.so (hand-written gfx950 HIP kernels), not in a TF graph.

Eager-host adaptation of the graph protocol: `loss(x, y)` accepts either
device tensors (evaluated now, eps ~ N(0,1) drawn on the device like
GaussianSampleLayer, util/layers.py:152-156) or the lazy (image, label) handles returned
by `analyzer.read` (evaluated by the trainer each step, like `sess.run`).
"""
from hipvae.engine import Engine


class LossDict(dict):
    """dict with keys 'G', 'D_KL', 'logP' (model/vae.py:127-130) plus the wiring the
    trainer needs in an eager host."""
    machine = None
    source = None


class ConvVAE(object):
    def __init__(self, arch, is_training=False, device=None, seed=None, impl=None, precision=None):
        self.arch = arch
        self._sanity_check()
        self.is_training = is_training           # unused, kept like the reference (vae.py:14)
        self.engine = Engine(arch, device=device, impl=impl, precision=precision)
        self.engine.init_params(seed)
        # sampler stream of eager loss() calls: Philox keyed by (seed, rank), one counter tick per call
        import os
        from hipvae.dp import rank_seed
        self._eps_seed = rank_seed(0 if seed is None else seed, int(os.environ.get('RANK', '0')),
                                   int(os.environ.get('WORLD_SIZE', '1'))) ^ 0x5A17
        self._eps_calls = 0
        self.generate = self.decode              # vae.py:34 (VAE-GAN extension alias)

    def _sanity_check(self):                     # vae.py:37-39
        for net in ['encoder', 'generator']:
            assert len(self.arch[net]['output']) == len(self.arch[net]['kernel']) == len(self.arch[net]['stride'])

    @property
    def y_emb(self):                             # vae.py:20-24
        return self.engine.param_views()['y_embedding/y_emb']

    def _draw_eps(self, F):
        """The N(0,1) draw [F, z_dim] of GaussianSampleLayer (util/layers.py:154) from the library's own
        counter-based generator (one fresh counter value per call; ranks use different keys)."""
        self._eps_calls += 1
        return self.engine.philox_normal(F, self._eps_seed, self._eps_calls)

    def loss(self, x, y, eps=None):
        """vae.py:106-137 -> {'G': -logPx + D_KL, 'D_KL', 'logP'} (0-d device tensors),
        or a lazy LossDict when x/y are input-queue handles."""
        out = LossDict()
        out.machine = self
        if hasattr(x, 'source'):                 # lazy handles from analyzer.read
            out.source = x.source
            out.update({'G': None, 'D_KL': None, 'logP': None})
            return out
        if eps is None:       # drawn inside the sampler kernel
            self._eps_calls += 1
            l3 = self.engine.loss_fwd(x, y, seed=self._eps_seed, offset=self._eps_calls).clone()
        else:
            l3 = self.engine.loss_fwd(x, y, eps).clone()
        out.update({'G': l3[0], 'D_KL': l3[1], 'logP': l3[2]})
        return out

    def encode(self, x):                         # vae.py:139-141 : z_mu only
        return self.engine.encode(x)

    def decode(self, z, y):                      # vae.py:143-145 : NHWC [F, H, 1, 1]
        xh = self.engine.decode(z, y)
        return xh.view(xh.shape[0], xh.shape[1], 1, 1)


def __getattr__(name):
    """`--model VAWGAN` with the default `--model_module model.vae` (main.py:27-31 of the reference loads the model
    class from this module by name); the class lives in model/vawgan.py."""
    if name == 'VAWGAN':
        from model.vawgan import VAWGAN
        return VAWGAN
    raise AttributeError(name)
