"""`--trainer_module trainer.vae --trainer VAETrainer | VAWGANTrainer` plugins (drop-in for the
reference's trainer/vae.py:9-111 and 115-218 + the GANTrainer base ctor trainer/gan.py:13-23).

Adam(lr, beta1, beta2 from arch['training']) over ALL trainables as one fused HIP
kernel; data-parallel over torch.distributed when WORLD_SIZE > 1.
"""
import logging
import os
import time

import torch

from hipvae.dp import Stepper


class VAETrainer(object):
    def __init__(self, loss, arch, args, dirs):          # trainer/gan.py:14-23
        self.loss = loss
        self.arch = arch
        self.args = args
        self.dirs = dirs
        self.opt = self._optimize()
        os.makedirs(dirs['logdir'], exist_ok=True)
        # trainer/gan.py:20-23 uses logging.basicConfig(filename=...); a dedicated handler does the
        # same job but also works when the root logger is already configured by the host program
        self.log = logging.getLogger('vaenpvc.train.%x' % id(self))
        self.log.setLevel(logging.INFO)
        self.log.propagate = False
        if self.opt['g'].rank == 0:
            self.log.addHandler(logging.FileHandler(os.path.join(dirs['logdir'], 'training.log')))
        self.summary = None

    def _optimize(self):                                  # trainer/vae.py:10-28
        t = self.arch['training']
        machine = getattr(self.loss, 'machine', None)
        if machine is None:
            raise ValueError('loss must come from ConvVAE.loss (it carries the machine)')
        stepper = Stepper(machine.engine, t['lr'], t['beta1'], t['beta2'], seed=getattr(self.args, 'seed', 0) or 0)
        return {'g': stepper, 'global_step': lambda: stepper.step_count}

    def _status_message(self, step, logP, D_KL):          # trainer/vae.py:47-50
        msg = 'Iter {:05d}: '.format(step)
        msg += 'log P(x|z, y) = {:.3e} '.format(logP)
        msg += 'D_KL(z) = {:.3e} '.format(D_KL)
        return msg

    def _refresh_status(self, l3=None):                   # trainer/vae.py:31-52
        """Formats and logs the status line from the LAST step's losses.  Issues NO collective (the loss
        sums arrive with the gradient all-reduce, hipvae/dp.py), so ranks may refresh at different
        wall-clock moments without ever desynchronising the communicator."""
        st = self.opt['g']
        l3 = st.mean_losses(l3).cpu()
        msg = self._status_message(st.step_count, float(l3[2]), float(l3[1]))
        if st.rank == 0:
            print('\r{}'.format(msg), end='', flush=True)
            self.log.info(msg)
        return msg

    def save(self, step=None):
        st = self.opt['g']
        if st.rank != 0:
            return None
        path = os.path.join(self.dirs['logdir'], 'model.ckpt-{}'.format(st.step_count if step is None else step))
        sd = st.state_dict()
        sd['layout'] = list(st.backend.layout.items()) if hasattr(st.backend, 'layout') else None
        self._save_source(sd)
        torch.save(sd, path)
        return path

    def _save_source(self, sd):
        """The input pipeline's state travels with the checkpoint (every rank draws the same global index sequence, so
        rank 0's state is everybody's): a restored run continues the record sequence instead of replaying it."""
        src = getattr(self.loss, 'source', None)
        if src is not None and hasattr(src, 'state_dict'):
            sd['source'] = src.state_dict()

    def _restore_source(self, sd):
        src = getattr(self.loss, 'source', None)
        if src is not None and hasattr(src, 'load_state_dict') and sd.get('source') is not None:
            if src.load_state_dict(sd['source']) is False and getattr(self.opt.get('g'), 'rank', 0) == 0:
                # (another global batch size / world size than the checkpoint's: the record stream starts afresh from the seed)
                import logging
                logging.warning('checkpoint holds the data-source state of another batch size or world size: '
                                'the record stream restarts from its seed (parameters, Adam slots and step are restored)')

    def restore(self, restore_from, ckpt=None):
        """tf.train.Supervisor restore (trainer/vae.py:77-84 + util/wrapper.py:32-62): parameters, the Adam
        slots and global_step of `<restore_from>/<ckpt or newest model.ckpt-N>`.  Returns the restored
        step, or None when the directory holds no checkpoint (fresh run, like the Supervisor)."""
        from util.wrapper import find_ckpt, read_ckpt
        path = find_ckpt(restore_from, ckpt)
        if path is None:
            return None
        sd = read_ckpt(path, getattr(self.opt['g'].backend, 'layout', None))
        self.opt['g'].load_state_dict(sd)
        self._restore_source(sd)
        return self.opt['g'].step_count

    def train(self, nIter, machine=None, summary_op=None, status_secs=60, save_secs=300, summary_secs=120):
        """trainer/vae.py:73-99.  Like the reference, the iteration count comes from
        arch['training']['max_iter'] (nIter is ignored, trap T7) and counts GLOBAL steps: a restored
        run continues at its checkpoint's step."""
        st = self.opt['g']
        machine = machine or self.loss.machine
        source = self.loss.source
        if source is None:
            raise ValueError('loss was not built from analyzer.read() handles')
        restore_from = self.dirs.get('restore_from')
        if restore_from and os.path.isdir(restore_from) and (getattr(self.args, 'restore_from', None)
                                                              or getattr(self.args, 'ckpt', None)):
            step = self.restore(restore_from, getattr(self.args, 'ckpt', None))
            if st.rank == 0 and step is not None:
                self.log.info('restored step {} from {}'.format(step, restore_from))
        st.broadcast_params()
        max_iter = self.arch['training']['max_iter']
        if st.rank == 0 and self.summary is None and hasattr(st.backend, 'summary'):
            from util.summary import SummaryWriter          # Supervisor summary thread (save_summaries_secs = 120)
            self.summary = SummaryWriter(self.dirs['logdir'], st.backend, secs=summary_secs)
        t_status = t_save = time.time()
        l3 = None
        while st.step_count < max_iter:
            x, y = source.next_batch()                    # analyzer.read dequeue
            l3 = st.step(x, y)                            # sess.run(self.opt['g']); the sampler draws on the device
            now = time.time()
            if self.summary is not None and self.summary.due(now):      # model/vae.py:132-136, rank 0 only
                from hipvae import lib as L
                xh = st.backend.ws_region(x.shape[0], L.MODE_TRAIN, 'xh')
                self.summary.write(st.step_count, st.mean_losses(l3).cpu(), x, xh)
            if now - t_status >= status_secs:             # per-rank wall clock is fine: no collective inside
                self._refresh_status(l3)
                t_status = now
            if now - t_save >= save_secs:                 # Supervisor save_model_secs=300
                self.save()
                t_save = now
        if l3 is not None:
            self._refresh_status(l3)
        return self.save()


class VAWGANTrainer(VAETrainer):
    """trainer/vae.py:115-218: three minimize ops on name-filtered variable lists, nIterD critic steps per
    generator step, each `sess.run` on its own batch.  `loss` comes from model.vawgan.VAWGAN.loss."""

    def _optimize(self):                                  # trainer/vae.py:116-147
        from hipvae.adversarial import AdvStepper
        t = self.arch['training']
        machine = getattr(self.loss, 'machine', None)
        if machine is None or not hasattr(machine, 'critic'):
            raise ValueError('loss must come from VAWGAN.loss (it carries the machine and its critic)')
        st = AdvStepper(machine.engine, machine.critic, t['lr'], t['beta1'], t['beta2'], t['alpha'], t['lambda'],
                        seed=getattr(self.args, 'seed', 0) or 0)
        return {'d': st.critic_step, 'g': st, 'e': st.generator_step, 'global_step': lambda: st.step_count}

    def _status_message(self, step, W_dist, logP, D_KL, gp):      # trainer/vae.py:212-216
        msg = 'Iter {:05d}: '.format(step)
        msg += 'W_dist = {:.4e} '.format(W_dist)
        msg += 'log P(x|z, y) = {:.4e} '.format(logP)
        msg += 'D_KL(z) = {:.4e} '.format(D_KL)
        msg += 'GP = {:.4e} '.format(gp)
        return msg

    def _refresh_status(self, l3=None):                           # trainer/vae.py:195-218
        """Status line from the LAST critic / generator steps (the reference evaluates the losses on one more
        dequeued batch; reusing the steps' own values costs no batch and no extra launches)."""
        st = self.opt['g']
        v = {k: float(t) for k, t in st.status.items()}
        msg = self._status_message(st.step_count, v['W_dist'], v['logP'], v['D_KL'], v['gp'])
        if st.rank == 0:
            print('\r{}'.format(msg), end='', flush=True)
            self.log.info(msg)
        return msg

    def save(self, step=None):
        st = self.opt['g']
        if st.rank != 0:
            return None
        path = os.path.join(self.dirs['logdir'], 'model.ckpt-{}'.format(st.step_count if step is None else step))
        sd = st.state_dict()
        sd['layout'] = list(st.backend.layout.items())
        sd['d_layout'] = list(st.critic.layout.items())
        self._save_source(sd)
        torch.save(sd, path)
        return path

    def restore(self, restore_from, ckpt=None):
        from util.wrapper import find_ckpt, read_ckpt
        path = find_ckpt(restore_from, ckpt)
        if path is None:
            return None
        sd = read_ckpt(path)
        self.opt["g"].load_state_dict(sd)
        self._restore_source(sd)
        return self.opt['g'].step_count

    def train(self, nIter, machine=None, summary_op=None, status_secs=60, save_secs=300, summary_secs=120):
        """trainer/vae.py:150-179: `max_iter` iterations of nIterD critic steps + one generator step."""
        st = self.opt['g']
        source = self.loss.source
        if source is None:
            raise ValueError('loss was not built from analyzer.read() handles')
        restore_from = self.dirs.get('restore_from')
        if restore_from and os.path.isdir(restore_from) and (getattr(self.args, 'restore_from', None)
                                                              or getattr(self.args, 'ckpt', None)):
            step = self.restore(restore_from, getattr(self.args, 'ckpt', None))
            if st.rank == 0 and step is not None:
                self.log.info('restored step {} from {}'.format(step, restore_from))
        st.broadcast_params()
        t = self.arch['training']
        if st.rank == 0 and self.summary is None and hasattr(st.backend, 'summary'):
            from util.summary import SummaryWriter          # the Supervisor's summary thread serves this trainer too (trainer/vae.py:160-166)
            self.summary = SummaryWriter(self.dirs['logdir'], st.backend, secs=summary_secs)
        t_status = t_save = time.time()
        while st.step_count < t['max_iter']:
            # trainer/vae.py:177-178: nIterD critic steps, each on its own batch (dequeued in the same order; the
            # generator's forward pass of all of them runs as one, see AdvStepper.critic_steps)
            st.critic_steps([source.next_batch() for _ in range(t['nIterD'])])
            x, y = source.next_batch()                    # trainer/vae.py:179
            st.generator_step(x, y)
            now = time.time()
            if self.summary is not None and self.summary.due(now):      # model/vae.py:132-136 summaries, rank 0 only
                from hipvae import lib as L
                xh = st.backend.ws_region(x.shape[0], L.MODE_TRAIN, 'xh')
                v = st.status
                # SummaryWriter takes {G, D_KL, logP}; this branch has no single G: l_E = -logP + D_KL stands in its place
                l3 = [float(v['D_KL']) - float(v['logP']), float(v['D_KL']), float(v['logP'])]
                self.summary.write(st.step_count, l3, x, xh)
            if now - t_status >= status_secs:
                self._refresh_status()
                t_status = now
            if now - t_save >= save_secs:
                self.save()
                t_save = now
        self._refresh_status()
        return self.save()
