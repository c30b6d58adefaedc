"""`--model_module model.vawgan --model VAWGAN` plugin: the model class the reference's VAWGANTrainer
(trainer/vae.py:115-218) trains.  The reference tree does not contain it (README.md:3: "for VAWGAN, please
switch to `vawgan` branch"); what it does fix -- the loss keys the trainer reads, the variable groups, the
discriminator's layer table and the hyper-parameters in architecture-vawgan-vcc2016.json -- is kept, the rest
is specified in DESIGN.md section 9.  Encoder and generator ARE the ConvVAE's (model/vae.py:72-103); the
critic runs in the same library (csrc/disc.hip).
"""
import torch

from hipvae.critic import Critic
from model.vae import ConvVAE, LossDict


class VAWGAN(ConvVAE):
    def __init__(self, arch, is_training=False, device=None, seed=None, impl=None, precision=None):
        super(VAWGAN, self).__init__(arch, is_training=is_training, device=device, seed=seed, impl=impl,
                                     precision=precision)
        self.critic = Critic(arch, device=self.engine.device)
        self.critic.init_params(None if seed is None else seed + 1)
        self.discriminate = self._discriminate

    def _sanity_check(self):
        for net in ['encoder', 'generator', 'discriminator']:
            assert len(self.arch[net]['output']) == len(self.arch[net]['kernel']) == len(self.arch[net]['stride'])

    def _discriminate(self, x):
        """Critic value per frame, float32 [F]."""
        d, _ = self.critic.values(x, x)
        return d[:x.shape[0]]

    def loss(self, x, y, eps=None, t=None):
        """{'l_D', 'l_E', 'l_G', 'D_KL', 'logP', 'W_dist', 'gp'} (trainer/vae.py:141-143,196-201): 0-d device
        tensors, or a lazy LossDict when x / y are input-queue handles."""
        keys = ('l_D', 'l_E', 'l_G', 'D_KL', 'logP', 'W_dist', 'gp')
        out = LossDict()
        out.machine = self
        if hasattr(x, 'source'):
            out.source = x.source
            out.update({k: None for k in keys})
            return out
        from hipvae import lib as L
        tr = self.arch['training']
        F = x.shape[0]
        self._eps_calls += 1
        if eps is None:
            eps = self.engine.philox_normal(F, self._eps_seed, self._eps_calls)
        if t is None:
            t = self.engine.philox_uniform(F, self._eps_seed ^ 0x7157, self._eps_calls)
        l3 = self.engine.loss_fwd(x, y, eps).clone()
        xh = self.engine.ws_region(F, L.MODE_INFER, 'xh').view(F, -1)
        scratch = torch.empty_like(self.critic.params)
        l2 = self.critic.critic_fwd_bwd(x.reshape(F, -1), xh, t, tr['lambda'], scratch).clone()
        out.update({'D_KL': l3[1], 'logP': l3[2], 'W_dist': l2[0], 'gp': l2[1], 'l_E': l3[0],
                    'l_D': -l2[0] + tr['lambda'] * l2[1], 'l_G': -l3[2] + tr['alpha'] * l2[0]})
        return out
