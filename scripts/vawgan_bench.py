"""Developer tool: wall time of the VAWGAN trainer's steps (config 5 of BASELINE.json: batch 16) on one GPU.
usage: python scripts/vawgan_bench.py [--frames 16] [--iters 50]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    from model.vawgan import VAWGAN
    from hipvae.adversarial import AdvStepper
    arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vawgan-vcc2016.json')))
    t = arch['training']
    m = VAWGAN(arch, seed=1)
    st = AdvStepper(m.engine, m.critic, t['lr'], t['beta1'], t['beta2'], t['alpha'], t['lambda'], seed=2)
    F = a.frames
    x = torch.tanh(torch.randn(F, 513, device='cuda'))
    y = torch.randint(0, 10, (F,), device='cuda')

    def timed(fn, n):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def iteration():
        st.critic_steps([(x, y)] * t['nIterD'])
        st.generator_step(x, y)
    ms_d = timed(lambda: st.critic_step(x, y), a.iters)
    ms_g = timed(lambda: st.generator_step(x, y), a.iters)
    ms_i = timed(iteration, max(5, a.iters // 5))
    print(json.dumps({'frames': F, 'critic_step_ms': round(ms_d, 4), 'generator_step_ms': round(ms_g, 4),
                      'iteration_ms': round(ms_i, 4), 'nIterD': t['nIterD'],
                      'frames_per_s': round(F * (t['nIterD'] + 1) / ms_i * 1e3, 1),
                      'status': {k: float(v) for k, v in st.status.items()}}))


if __name__ == '__main__':
    main()
