"""Data-parallel train step: one process per GPU, gradients all-reduced (SUM) as ONE
flat float32 buffer over torch.distributed ("nccl" == RCCL over xGMI on ROCm), the
1/world_size mean folded into the fused Adam kernel (grad_scale).

The reference has no multi-GPU code at all; frames are independent
(per-sample LayerNorm, batch-mean loss), so sharding frames over ranks is exactly one
big batch up to summation order.

`backend` is anything with `.params`, `.n_params`, `.train_fwd_bwd(x,y,eps,grads)` and
`.adam_step(grads,m,v,step,lr,b1,b2,eps,grad_scale)`; the only shipped backend is
hipvae.Engine (HIP).  Tests inject a CPU stand-in to exercise this host logic under gloo.
"""
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(F, rank, world):
    """Equal contiguous shards (mean of local means == global mean)."""
    if F % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (F, world))
    per = F // world
    return rank * per, (rank + 1) * per


class Stepper(object):
    def __init__(self, backend, lr, beta1, beta2, eps=1e-8, group=None):
        self.backend = backend
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.group = group
        self.rank, self.world = world_info(group)
        import os
        # VAENPVC_FORCE_DIST=1: run the all-reduce even with one rank (smoke-tests the RCCL path)
        self._force_collective = os.environ.get('VAENPVC_FORCE_DIST') == '1' and dist.is_available() and dist.is_initialized()
        p = backend.params
        self.grads = torch.zeros_like(p)
        self.m = torch.zeros_like(p)      # Adam slots (tf.train.AdamOptimizer "m"/"v")
        self.v = torch.zeros_like(p)
        self.step_count = 0               # global_step (trainer/vae.py:15)

    def broadcast_params(self, src=0):
        if self.world > 1:
            dist.broadcast(self.backend.params, src=src, group=self.group)

    def step(self, x, y, eps):
        """x, y, eps are this rank's LOCAL shard.  Returns the local loss3 tensor."""
        loss3 = self.backend.train_fwd_bwd(x, y, eps, self.grads)
        if self.world > 1 or self._force_collective:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=self.group)
        self.step_count += 1
        self.backend.adam_step(self.grads, self.m, self.v, self.step_count, self.lr, self.beta1, self.beta2,
                               self.eps, 1.0 / self.world)
        return loss3

    # ------------------------------------------------------------------ hipGraph replay
    def capture(self, x, y, eps):
        """Capture one single-GPU train step (forward+backward+Adam, ~130 kernel launches) in a
        hipGraph on static copies of (x, y, eps); `replay()` then runs a step with one launch.
        Launch overhead dominates below a few thousand frames per step (the reference trains
        with batch 16).  The optimiser state is snapshotted around the capture warm-up, so the
        trajectory is unchanged.  Returns the static input tensors to copy new batches into."""
        if self.world != 1:
            raise RuntimeError('graph capture is implemented for single-process training only')
        be = self.backend
        self._gx, self._gy, self._ge = x.clone(), y.clone(), eps.clone()
        self._d_step = torch.full((1,), self.step_count, dtype=torch.int64, device=be.params.device)
        snap = (be.params.clone(), self.m.clone(), self.v.clone())

        def one_step():
            l3 = be.train_fwd_bwd(self._gx, self._gy, self._ge, self.grads)
            be.adam_step_dev(self.grads, self.m, self.v, self._d_step, self.lr, self.beta1, self.beta2, self.eps, 1.0)
            return l3
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up: module loads, attribute calls, workspace
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._gl3 = one_step()
        be.params.copy_(snap[0])
        self.m.copy_(snap[1])
        self.v.copy_(snap[2])
        self._d_step.fill_(self.step_count)
        return self._gx, self._gy, self._ge

    def replay(self):
        """One captured step on the current contents of the static inputs; returns loss3."""
        self._graph.replay()
        self.step_count += 1
        return self._gl3

    def mean_losses(self, loss3):
        """Average {G, D_KL, logP} over ranks (logging only)."""
        out = loss3.clone()
        if self.world > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
            out /= self.world
        return out

    def state_dict(self):
        return {'params': self.backend.params.detach().cpu(), 'm': self.m.cpu(), 'v': self.v.cpu(),
                'step': self.step_count}

    def load_state_dict(self, sd):
        self.backend.params.copy_(sd['params'].to(self.backend.params.device))
        self.m.copy_(sd['m'].to(self.m.device))
        self.v.copy_(sd['v'].to(self.v.device))
        self.step_count = int(sd['step'])
