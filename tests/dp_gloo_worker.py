"""One rank of the world_size-N gloo data-parallel test (launched by test_dp_gloo.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, 'vae-npvc_amd')):
    sys.path.insert(0, p)
from helpers import SMALL_ARCH  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402


class OracleBackend(object):
    """CPU stand-in for hipvae.Engine built from the oracle (test infrastructure)."""

    def __init__(self, arch, seed):
        self.arch = arch
        self.names = list(O.param_layout(arch).keys())
        self.params = torch.tensor(O.flatten_params(O.init_params(arch, seed)), dtype=torch.float64)
        self.n_params = self.params.numel()

    def train_fwd_bwd(self, x, y, eps, grads):
        P = O.unflatten_params(self.arch, self.params.numpy())
        L, G = O.torch_loss_and_grads(self.arch, P, x.numpy(), y.numpy(), eps.numpy(), torch.float64)
        grads.copy_(torch.tensor(np.concatenate([G[n].ravel() for n in self.names])))
        return torch.tensor([float(L['G']), float(L['D_KL']), float(L['logP'])], dtype=torch.float64)

    def adam_step(self, grads, m, v, step, lr, b1, b2, eps, grad_scale):
        p, mm, vv = O.tf_adam_step(self.params.numpy(), grads.numpy() * grad_scale, m.numpy(), v.numpy(), step,
                                   lr, b1, b2, eps)
        self.params.copy_(torch.tensor(p))
        m.copy_(torch.tensor(mm))
        v.copy_(torch.tensor(vv))


def run(rank, world, F, steps, out):
    from hipvae.dp import Stepper, shard_range
    be = OracleBackend(SMALL_ARCH, seed=10 + rank)     # different init per rank: broadcast must fix it
    st = Stepper(be, 1e-3, 0.5, 0.999)
    assert st.world == world and st.rank == rank
    st.broadcast_params()
    l3 = None
    for t in range(steps):
        x, y, eps = O.make_inputs(SMALL_ARCH, F, 50 + t)
        lo, hi = shard_range(F, rank, world)
        l3 = st.step(torch.tensor(x[lo:hi]), torch.tensor(y[lo:hi]), torch.tensor(eps[lo:hi]))
    gl = st.mean_losses(l3)
    if rank == 0:
        np.save(out, np.concatenate([be.params.numpy(), gl.numpy()]))


if __name__ == '__main__':
    torch.set_num_threads(1)
    rank, world, port, F, steps, out = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]),
                                        int(sys.argv[5]), sys.argv[6])
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = port
        dist.init_process_group('gloo', rank=rank, world_size=world)
    run(rank, world, F, steps, out)
    if world > 1:
        dist.destroy_process_group()
