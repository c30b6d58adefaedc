"""Data-parallel train step: one process per GPU, gradients all-reduced (SUM) over
torch.distributed ("nccl" == RCCL over xGMI on ROCm), the 1/world_size mean folded into the fused
Adam kernel (grad_scale).

The reference has no multi-GPU code at all; frames are independent (per-sample
LayerNorm, batch-mean loss), so sharding frames over ranks is exactly one big batch up to
summation order.

Overlap.  The backward pass finishes the flat gradient buffer back to front in four
contiguous ranges -- decoder convs, merge, heads, embedding + encoder -- and the library reports each
one through a callback the moment its kernels are enqueued (vaenpvc_set_bucket_callback).  The
callback starts that range's all-reduce on the communicator's stream, ordered after the library's
weight-gradient stream, so 3.3 of the 3.76 MB fly while the encoder backward still runs; Adam waits
for all of them.  The three losses ride in the tail of the same buffer (they are final before the
first range), so every rank holds the global mean losses after every step WITHOUT a collective of
its own: logging can never issue a collective that other ranks do not issue.

`backend` is anything with `.params`, `.n_params`, `.train_fwd_bwd(x, y, eps, grads, out=, seed=, offset=)`
and `.adam_step(...)`; the only shipped backend is hipvae.Engine (HIP).  Tests inject a CPU
stand-in to exercise this host logic under gloo.
"""
import os

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


def rank_seed(seed, rank, world):
    """Sampler seed of a rank: every rank must draw DIFFERENT noise for its shard (the reference
    draws independent N(0,1) per frame, util/layers.py:154)."""
    return (int(seed) * int(world) + int(rank)) & (2 ** 64 - 1)


class Stepper(object):
    TAIL = 4    # floats appended to the gradient buffer: {G, D_KL, logP} of the step + padding

    def __init__(self, backend, lr, beta1, beta2, eps=1e-8, group=None, overlap=True, seed=0):
        self.backend = backend
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.group = group
        self.rank, self.world = world_info(group)
        # VAENPVC_FORCE_DIST=1: run the collectives even with one rank (smoke-tests the RCCL path)
        self._force_collective = (os.environ.get('VAENPVC_FORCE_DIST') == '1' and dist.is_available()
                                  and dist.is_initialized())
        self.collective = self.world > 1 or self._force_collective
        p = backend.params
        n = p.numel()
        self._gbuf = torch.zeros(n + self.TAIL, dtype=p.dtype, device=p.device)
        self.grads = self._gbuf[:n]             # flat gradient buffer, same layout as the parameters
        self._loss_tail = self._gbuf[n:n + 3]   # this step's losses; SUM over ranks after the all-reduce
        self.m = torch.zeros_like(p)      # Adam slots (tf.train.AdamOptimizer "m"/"v")
        self.v = torch.zeros_like(p)
        self.step_count = 0               # global_step (trainer/vae.py:15)
        self.seed = rank_seed(seed, self.rank, self.world)
        self.overlap = bool(overlap) and hasattr(backend, 'set_bucket_callback')
        self._works = []
        self._covered = []
        self._cb_error = None
        self._cb_on = False

    def broadcast_params(self, src=0):
        if self.world > 1:
            dist.broadcast(self.backend.params, src=src, group=self.group)

    # ------------------------------------------------------------------ bucketed all-reduce
    def _bucket_ready(self, bucket, off, cnt, ready_stream):
        # This runs as a ctypes callback INSIDE vaenpvc_train_fwd_bwd: ctypes prints and swallows anything raised
        # here, which would leave step() with a short work list and Adam applied to un-reduced gradients (ranks
        # diverge silently).  Keep the first exception and re-raise it from step() after the library call returns.
        try:
            self._bucket_ready_impl(bucket, off, cnt, ready_stream)
        except BaseException as ex:       # noqa: BLE001
            if self._cb_error is None:
                self._cb_error = ex

    def _bucket_ready_impl(self, bucket, off, cnt, ready_stream):
        n = self.grads.numel()
        if off + cnt == n:
            cnt += self.TAIL                   # the losses travel with the range that ends the buffer
        seg = self._gbuf[off:off + cnt]
        if seg.is_cuda and ready_stream:
            # order the collective after the library stream that holds this range (ProcessGroupNCCL
            # makes its own stream wait for the stream that is current when the collective is issued)
            ext = torch.cuda.ExternalStream(int(ready_stream), device=seg.device)
            with torch.cuda.stream(ext):
                w = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)
        self._covered.append((off, off + cnt))

    def _check_coverage(self):
        """The ranges handed to the all-reduce must tile [0, n + TAIL) exactly once (a missed or doubled range would
        leave part of the gradient un-reduced while Adam still scales it by 1/world)."""
        end = self.grads.numel() + self.TAIL
        pos = 0
        for lo, hi in sorted(self._covered):
            if lo != pos:
                raise RuntimeError('gradient all-reduce ranges do not tile the buffer: gap or overlap at %d (next range starts at %d)' % (pos, lo))
            pos = hi
        if pos != end:
            raise RuntimeError('gradient all-reduce ranges cover [0, %d) of %d elements' % (pos, end))

    def _set_cb(self, on):
        if on != self._cb_on:
            self.backend.set_bucket_callback(self._bucket_ready if on else None)
            self._cb_on = on

    def step(self, x, y, eps=None):
        """x, y (and optionally an injected eps) are this rank's LOCAL shard.  Without eps the sampler
        draws on the device, keyed by (rank seed, global step).  Returns {G, D_KL, logP}: this rank's
        values on one rank, the mean over ranks otherwise."""
        be = self.backend
        use_cb = self.collective and self.overlap
        self._set_cb(use_cb)
        self._works = []
        self._covered = []
        self._cb_error = None
        if eps is None:
            loss3 = be.train_fwd_bwd(x, y, None, self.grads, out=self._loss_tail, seed=self.seed,
                                     offset=self.step_count)
        else:
            loss3 = be.train_fwd_bwd(x, y, eps, self.grads, out=self._loss_tail)
        if self._cb_error is not None:         # raised inside the library callback: nothing may be applied
            err, self._cb_error = self._cb_error, None
            for w in self._works:              # (collectives already started still have to be waited for)
                try:
                    w.wait()
                except Exception:              # noqa: BLE001
                    pass
            self._works = []
            raise RuntimeError('gradient-bucket callback failed; the optimiser step was NOT applied') from err
        if self.collective:
            if use_cb:
                self._check_coverage()
                for w in self._works:          # current stream waits for every range
                    w.wait()
                self._works = []
            else:
                dist.all_reduce(self._gbuf, op=dist.ReduceOp.SUM, group=self.group)
        self.step_count += 1
        be.adam_step(self.grads, self.m, self.v, self.step_count, self.lr, self.beta1, self.beta2,
                     self.eps, 1.0 / self.world)
        # (with a collective the tail now holds the SUM over ranks: hand out the mean, as a new tensor)
        return self._loss_tail / float(self.world) if self.collective else loss3

    def mean_losses(self, loss3=None):
        """{G, D_KL, logP} of the LAST step averaged over ranks.  No collective: the sums arrived with the
        gradient all-reduce, so ranks may call this at different times (or not at all)."""
        if self.collective:
            return self._loss_tail / float(self.world)
        return self._loss_tail.clone() if loss3 is None else loss3.clone()

    # ------------------------------------------------------------------ hipGraph replay
    def capture(self, x, y, eps=None, steps=1):
        """Capture one train step (forward + backward [+ all-reduce] + Adam, ~130 kernel launches) in a
        hipGraph on static copies of (x, y[, eps]); `replay()` then runs a step with one launch.
        steps = K > 1: K CONSECUTIVE steps in one graph on K static batches (x [K, F, ...], y [K, F][, eps
        [K, F, z]]): a graph launch has a fixed cost of its own (~10 us on this stack: at 16 / 256 frames a
        one-step graph replays slower than the eager launches it replaces), K steps share it; `replay()` then
        advances K steps and returns the losses of the last one.
        Launch overhead dominates below a few thousand frames per step (the reference trains with
        batch 16).  Without eps the sampler draws inside the graph, keyed by the DEVICE step counter, so
        every replay sees fresh noise.  With more than one rank the (single, unbucketed) gradient
        all-reduce is captured with the step.  The optimiser state is snapshotted around the capture
        warm-up, so the trajectory is unchanged.  Returns the static input tensors to copy new
        batches into."""
        be = self.backend
        self._set_cb(False)
        steps = int(steps)
        if steps < 1:
            raise ValueError('steps must be >= 1')
        if steps > 1 and (x.shape[0] != steps or y.shape[0] != steps or (eps is not None and eps.shape[0] != steps)):
            raise ValueError('capture(steps=%d): the leading dimension of x, y[, eps] is the step' % steps)
        if steps == 1 and y.dim() == 2:
            # a stacked [1, F, ...] input (the K-step calling convention with K = 1): take the one batch, never pass the extra
            # dimension through -- the library would read F = 1 from it
            if y.shape[0] != 1 or x.shape[0] != 1 or (eps is not None and eps.shape[0] != 1):
                raise ValueError('capture(steps=1): x, y[, eps] carry a leading step dimension of %d' % y.shape[0])
            x, y, eps = x[0], y[0], (eps[0] if eps is not None else None)
        self._gsteps = steps
        self._gx, self._gy = x.clone(), y.clone()
        self._ge = eps.clone() if eps is not None else None
        self._d_step = torch.full((1,), self.step_count, dtype=torch.int64, device=be.params.device)
        snap = (be.params.clone(), self.m.clone(), self.v.clone())

        def batch(t, i):
            return t if steps == 1 else t[i]

        def one_step(i=0):
            if self._ge is None:
                l3 = be.train_fwd_bwd(batch(self._gx, i), batch(self._gy, i), None, self.grads, out=self._loss_tail, seed=self.seed,
                                      offset=0, d_offset=self._d_step)
            else:
                l3 = be.train_fwd_bwd(batch(self._gx, i), batch(self._gy, i), batch(self._ge, i), self.grads, out=self._loss_tail)
            if self.collective:
                dist.all_reduce(self._gbuf, op=dist.ReduceOp.SUM, group=self.group)
            be.adam_step_dev(self.grads, self.m, self.v, self._d_step, self.lr, self.beta1, self.beta2, self.eps,
                             1.0 / self.world)
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
            for i in range(steps):
                self._gl3 = one_step(i)
        be.params.copy_(snap[0])
        self.m.copy_(snap[1])
        self.v.copy_(snap[2])
        self._d_step.fill_(self.step_count)
        return (self._gx, self._gy, self._ge) if self._ge is not None else (self._gx, self._gy)

    def replay(self):
        """The captured step(s) on the current contents of the static inputs; returns loss3 of the last one (with more
        than one rank: the SUM over ranks -- use mean_losses())."""
        self._graph.replay()
        self.step_count += getattr(self, '_gsteps', 1)
        return self._gl3

    def state_dict(self):
        return {'params': self.backend.params.detach().cpu(), 'm': self.m.cpu(), 'v': self.v.cpu(),
                'step': self.step_count}

    def load_state_dict(self, sd):
        self.backend.params.copy_(sd['params'].to(self.backend.params.device))
        self.m.copy_(sd['m'].to(self.m.device))
        self.v.copy_(sd['v'].to(self.v.device))
        self.step_count = int(sd['step'])
