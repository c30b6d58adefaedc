#!/usr/bin/env python
"""bench.py -- ConvVAE train-step throughput on MI355X (driver contract).

A "step" = one pass of the hot path over one synthetic batch that is already resident
in HBM: forward + backward (vaenpvc_train_fwd_bwd_seeded: the sampler's N(0,1) draw is generated
on the device, Philox) [+ RCCL all-reduce of the gradient buckets when N > 1] + fused TF-Adam.
Workload at N = 1 (BASELINE.json configs[1], north_star "batch 256 x [1,513,128]"): 256*128 =
32768 independent 513-bin frames per step per GPU (weak scaling: fixed per-GPU work).  The literal
F = 256 and F = 16 (configs[0]) readings are reported in `config.literal_batches`, one iteration of the VAWGAN branch
(configs[4]: 5 critic steps + 1 generator step at 16 frames per step) in `config.vawgan_config5`.

Arithmetic: fp32 tensors everywhere; the GEMM-shaped kernels that run on the bf16 matrix cores split
every fp32 operand into bf16 terms with fp32 accumulation.  The default (`value`, "bf16x2") uses 2 terms
(16 mantissa bits per operand, three products): it holds the parity bars of north_star at the benchmarked
size (tests/test_gpu_parity.py::test_benchmarked_batch_sizes_against_oracle_fixture: 1e-4 activations /
losses, 2e-4 gradients; measured ~7e-6 / <= 1.5e-5).  `modes` reports the fp32-exact 3-term variant and the
plain-bf16 mode (BASELINE config 2's literal dtype; tolerance 3e-2, stated in the tests) beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
sys.path.insert(0, ROOT)

# SURVEY 8(d) / BASELINE.md section 4: algorithmic figures per frame
FLOP_PER_FRAME_TRAIN = 28.85e6          # 3 x effective forward MACs x 2
BYTES_PER_FRAME_TRAIN = 300752.0        # layer-materialised model, fp32 activations
BYTES_PER_STEP_PARAMS = 37.57e6         # weights fwd+bwd, grads, Adam state
HBM_PEAK = 8.0e12
FP32_PEAK = 157.3e12
# dominant kernel family: the last decoder layer (1025-tap conv_transpose = dense Toeplitz GEMM
# [F,4104] x [4104,513]); algorithmic flops per frame for one pass (fwd, or dgrad, or wgrad)
DEC3_FLOP_PER_FRAME = 2.0 * 8 * 513 * 513
BF16_PEAK = 2500e12
PRODUCTS = {3: 6, 2: 3, 1: 1}           # bf16 MFMA products per fp32 product for 3 / 2 / 1 operand terms
PREC_NAME = {3: 'bf16x3', 2: 'bf16x2', 1: 'bf16'}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=100)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--frames', type=int, default=256 * 128, help='frames per step PER GPU')
    p.add_argument('--impl', default='auto', choices=['auto', 'generic'])
    p.add_argument('--precision', default='bf16x2', choices=['auto', 'bf16x2', 'bf16x3', 'bf16'])
    p.add_argument('--timer-tag', default='dec3_wgrad', help='kernel site timed with HIP events for the roofline')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=24.0, help='total budget of the CPU legs')
    p.add_argument('--no-literal', action='store_true', help='skip the extra F=256 / F=16 measurements')
    p.add_argument('--no-modes', action='store_true', help='skip the other precisions')
    return p.parse_args()


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def cpu_baseline(arch, seconds):
    """The oracle's PyTorch-CPU fp32 restatement (CPU stand-in for the TF1 reference path, which cannot
    run here): full train step (fwd + autograd bwd + TF-Adam) at F = 256 (k threads and 1 thread) and at
    F = 16 (configs[0], the reference's own batch), plus forward-only encode -> decode at F = 1024
    (config 4); BASELINE.md section 3.  Bounded: `seconds` of CPU work in total."""
    import numpy as np
    import torch
    from oracle import convvae_oracle as O
    threads = int(os.environ.get('VAENPVC_CPU_THREADS', min(os.cpu_count() or 1, 16)))   # measured on the MI355X box: 16 threads is the fastest (8: 865, 16: 1678, 32: 1237, 64: 857 frames/s)

    def train_leg(F, k, budget):
        torch.set_num_threads(k)
        P = O.torch_params(O.init_params(arch, 0), torch.float32, requires_grad=True)
        x, y, eps = O.make_inputs(arch, F, 0)
        xt, yt, et = torch.tensor(x), torch.tensor(y), torch.tensor(eps)
        m = {n: torch.zeros_like(v) for n, v in P.items()}
        v2 = {n: torch.zeros_like(v) for n, v in P.items()}

        def step(t):
            for p in P.values():
                p.grad = None
            O.torch_loss(arch, P, xt, yt, et)['G'].backward()
            lr_t = 1e-4 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.5 ** t)
            with torch.no_grad():
                for n, p in P.items():
                    g = p.grad
                    m[n].mul_(0.5).add_(g, alpha=0.5)
                    v2[n].mul_(0.999).addcmul_(g, g, value=0.001)
                    p.sub_(lr_t * m[n] / (v2[n].sqrt() + 1e-8))
        step(1)
        times, t0, t = [], time.perf_counter(), 2
        while time.perf_counter() - t0 < budget and len(times) < 200:
            a = time.perf_counter()
            step(t)
            times.append(time.perf_counter() - a)
            t += 1
        med = float(np.median(times))
        return {'frames_per_s': F / med, 'ms_per_step': med * 1e3, 'threads': k, 'frames_per_step': F, 'steps': len(times),
                'p10_ms': float(np.percentile(times, 10)) * 1e3, 'p90_ms': float(np.percentile(times, 90)) * 1e3}

    def fwd_leg(F, k, budget):
        torch.set_num_threads(k)
        P = O.torch_params(O.init_params(arch, 0), torch.float32)
        x, y, _ = O.make_inputs(arch, F, 0)
        xt, yt = torch.tensor(x), torch.full((F,), 9, dtype=torch.int64)
        times, t0 = [], time.perf_counter()
        with torch.no_grad():
            while time.perf_counter() - t0 < budget and len(times) < 100:
                a = time.perf_counter()
                z_mu, _, _ = O.torch_encode(arch, P, xt)
                O.torch_decode(arch, P, z_mu, yt)
                times.append(time.perf_counter() - a)
        med = float(np.median(times[1:] or times))
        return {'frames_per_s': F / med, 'ms': med * 1e3, 'threads': k, 'frames': F, 'runs': len(times)}
    main = train_leg(256, threads, 0.42 * seconds)
    legs = {'train_F256_1thread': train_leg(256, 1, 0.25 * seconds),
            'train_F16': train_leg(16, threads, 0.12 * seconds),
            'train_F16_1thread': train_leg(16, 1, 0.08 * seconds),
            'convert_fwd_F1024': fwd_leg(1024, threads, 0.13 * seconds)}
    return {'value': main['frames_per_s'], 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': 'oracle torch-CPU fp32 train step (fwd+bwd+TF-Adam), F=256 frames/step, %d steps, median' % main['steps'],
            'ms_per_step': main['ms_per_step'], 'p10_ms': main['p10_ms'], 'p90_ms': main['p90_ms'],
            'cpu_model': cpu_model(), 'host_threads': os.cpu_count(), 'legs': legs}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from hipvae import Engine
    from hipvae import lib as L
    from hipvae.dp import Stepper

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    torch.cuda.set_device(local)
    force_dist = os.environ.get('VAENPVC_FORCE_DIST') == '1'      # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    with open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')) as fp:
        arch = json.load(fp)
    t = arch['training']
    eng = Engine(arch, impl=args.impl, precision=args.precision)
    eng.init_params(seed=0)
    st = Stepper(eng, t['lr'], t['beta1'], t['beta2'], seed=0)
    st.broadcast_params()
    planes = L.PRECISIONS[args.precision]

    def make_batch(F, seed):
        g = torch.Generator(device='cpu').manual_seed(seed)
        x = (torch.rand(F, 513, generator=g) * 2 - 1).to(eng.device)
        y = torch.randint(0, 10, (F,), generator=g, dtype=torch.int64).to(eng.device)
        return x, y

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(F, steps, warmup, tag=None):
        x, y = make_batch(F, 1234 + rank)
        for _ in range(warmup):
            st.step(x, y)
        barrier()
        if tag:
            eng.timer_select(tag)
        t0 = time.perf_counter()
        for _ in range(steps):
            st.step(x, y)
        barrier()
        dt = time.perf_counter() - t0
        kern = None
        if tag:
            ms, n = eng.timer_read()
            eng.timer_select(None)
            if n:
                kern = (ms / n, n)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=eng.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, kern

    F = args.frames
    dt, kern = timed(F, args.steps, args.warmup, None if args.impl == 'auto' else args.timer_tag)
    frames_per_s = world * F * args.steps / dt
    steps_per_s = args.steps / dt
    out = {
        'metric': 'SP frames/sec (train step: fwd+bwd+Adam%s)' % ('+RCCL all-reduce' if world > 1 else ''),
        'value': frames_per_s, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32',
        'arithmetic': ('fp32 tensors, fp32 accumulation; GEMM operands on the bf16 matrix cores split into bf16 terms: %s'
                       % {3: '3 terms (fp32-exact)',
                          2: '2 terms (16 mantissa bits per operand; holds the 1e-4 / 2e-4 parity bars with >10x margin)',
                          1: 'plain bf16 operands: the reduced-precision bf16 mode'}[planes]),
        'data': 'synthetic (x~U(-1,1), y~randint(10) resident in HBM; eps~N(0,1) drawn on the device per step, Philox4x32-10; random-init weights)',
        'config': {'workload': 'ConvVAE architecture-vae-vcc2016 train step, 256x[1,513,128] = %d frames/step/GPU' % F,
                   'frames_per_step_per_gpu': F, 'global_frames_per_step': F * world, 'impl': args.impl,
                   'precision': args.precision, 'parallelism': 'dp%d' % world,
                   'all_reduce': 'four gradient buckets overlapped with the backward pass' if world > 1 else None},
        'step_fraction_of_rooflines': {
            'hbm_model_B': (frames_per_s / world * BYTES_PER_FRAME_TRAIN + steps_per_s * BYTES_PER_STEP_PARAMS) / HBM_PEAK,
            'fp32_flops': frames_per_s / world * FLOP_PER_FRAME_TRAIN / FP32_PEAK},
    }
    # ---- roofline of the dominant kernel: a separate short pass with the weight-gradient stream
    #      serialised (backward-mask bit 30 cleared), so that the HIP-event duration of a kernel is
    #      not inflated by kernels running concurrently on the other stream.  Not part of `value`.
    def kernel_ms(tag, steps=6):
        eng.set_tuned_masks(0xffffffff, 0xbfffffff)
        try:
            _, k = timed(F, steps, 1, tag)
        finally:
            eng.set_tuned_masks(0xffffffff, 0xffffffff)
        return k
    kern = kernel_ms(args.timer_tag) if args.impl == 'auto' else kern
    if kern:
        avg_ms, n = kern
        ach = DEC3_FLOP_PER_FRAME * F / (avg_ms * 1e-3) / 1e12
        # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (separate
        # --pmc runs of this same command, FETCH_SIZE/WRITE_SIZE in KiB; see profiles/README.md)
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fp:
                tj = json.load(fp).get('%s/%s' % (args.timer_tag, args.precision))
                if tj and tj.get('frames') == F:
                    traffic = tj['hbm_bytes_per_launch']
        except (OSError, ValueError):
            pass
        bf16 = args.timer_tag in ('dec3_fwd', 'dec3_dgrad', 'dec3_wgrad') and F >= 8192 and args.impl == 'auto'
        peak = (BF16_PEAK / PRODUCTS[planes] if bf16 else FP32_PEAK) / 1e12
        out['roofline'] = {'bound': 'mfma', 'kernel': args.timer_tag, 'achieved': ach, 'peak': peak,
                           'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
                           'avg_kernel_ms': avg_ms, 'launches': n,
                           'algorithmic_flops_per_launch': DEC3_FLOP_PER_FRAME * F,
                           'peak_basis': ('dense bf16 MFMA peak / %d (%d-term operand split: %d bf16 products per fp32 product)'
                                          % (PRODUCTS[planes], planes, PRODUCTS[planes]) if bf16
                                          else 'exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'),
                           'measured': 'HIP events on the launch stream, side stream serialised'}
        if args.impl == 'auto' and F >= 8192 and args.timer_tag == 'dec3_wgrad':
            # the two sibling GEMMs of the same layer (same algorithmic flops)
            sib = {}
            for tag in ('dec3_fwd', 'dec3_dgrad'):
                k = kernel_ms(tag, 4)
                if k:
                    a2 = DEC3_FLOP_PER_FRAME * F / (k[0] * 1e-3) / 1e12
                    sib[tag] = {'avg_kernel_ms': k[0], 'achieved_tflops_fp32_equiv': a2, 'peak': peak, 'frac': a2 / peak}
            out['roofline']['sibling_kernels'] = sib
    # ---- the other precisions beside the default (never instead of it)
    if not args.no_modes and args.impl == 'auto':
        modes = {('bf16x2' if args.precision == 'auto' else args.precision): {'ms_per_step': dt / args.steps * 1e3, 'frames_per_s': frames_per_s}}
        for prec in ('bf16x2', 'bf16x3', 'bf16'):
            if prec in modes:
                continue
            eng.set_precision(prec)
            d2, _ = timed(F, max(10, args.steps // 4), 3)
            n2 = max(10, args.steps // 4)
            modes[prec] = {'ms_per_step': d2 / n2 * 1e3, 'frames_per_s': world * F * n2 / d2}
        eng.set_precision(args.precision)
        modes['note'] = ('bf16x2 (default): 2-term operand split; bf16x3: 3 terms, fp32-exact; '
                         'bf16: plain bf16 operands on the kernels that run on the bf16 matrix cores (tolerance 3e-2, tests)')
        out['modes'] = modes
    if not args.no_literal:
        lits = {}
        for Fl, nst in ((256, 200), (16, 200)):
            dt2, _ = timed(Fl, nst, 10)
            lit = {'frames_per_s': world * Fl * nst / dt2, 'ms_per_step': dt2 / nst * 1e3, 'launch': 'eager'}
            # same step captured in a hipGraph (one launch per step instead of ~130); with N > 1 the
            # (unbucketed) gradient all-reduce is captured with it
            try:
                x, y = make_batch(Fl, 99)
                st.capture(x, y)
                for _ in range(10):
                    st.replay()
                barrier()
                t0 = time.perf_counter()
                for _ in range(nst):
                    st.replay()
                barrier()
                dtg = time.perf_counter() - t0
                lit['hipgraph'] = {'frames_per_s': world * Fl * nst / dtg, 'ms_per_step': dtg / nst * 1e3}
            except Exception as ex:       # noqa: BLE001  (capture support differs between RCCL builds)
                lit['hipgraph'] = {'error': str(ex)[:200]}
            lits['F%d' % Fl] = lit
        out['config']['literal_batches'] = lits
        out['config']['literal_batch256'] = lits['F256']
    if not args.no_literal:
        # BASELINE.json configs[4]: the VAWGAN branch (nIterD critic steps + one generator step per iteration, 16 frames per
        # step and GPU; hipvae/adversarial.py, data parallel over the same process group).  Reported beside the headline.
        try:
            from hipvae.critic import Critic
            from hipvae.adversarial import AdvStepper
            with open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vawgan-vcc2016.json')) as fp:
                varch = json.load(fp)
            vt = varch['training']
            veng, vcr = Engine(varch, precision=args.precision), Critic(varch)
            veng.init_params(seed=0)
            vcr.init_params(seed=1)
            vst = AdvStepper(veng, vcr, vt['lr'], vt['beta1'], vt['beta2'], vt['alpha'], vt['lambda'], seed=0)
            vst.broadcast_params()
            Fv = vt['batch_size']
            g = torch.Generator(device='cpu').manual_seed(7 + rank)
            xv = torch.rand(Fv, eng.H, generator=g).mul_(2).sub_(1).cuda()
            yv = torch.randint(0, varch['y_dim'], (Fv,), generator=g).cuda()

            def iteration():
                vst.critic_steps([(xv, yv)] * vt['nIterD'])
                vst.generator_step(xv, yv)
            for _ in range(3):
                iteration()
            barrier()
            t0 = time.perf_counter()
            nit = 30
            for _ in range(nit):
                iteration()
            barrier()
            dtv = time.perf_counter() - t0
            out['config']['vawgan_config5'] = {
                'ms_per_iteration': dtv / nit * 1e3, 'frames_per_s': world * Fv * (vt['nIterD'] + 1) * nit / dtv,
                'frames_per_step_per_gpu': Fv, 'nIterD': vt['nIterD'],
                'status': {k: float(v) for k, v in vst.status.items()},
                'note': 'model specified in DESIGN.md section 9 (the reference tree holds only its trainer)'}
            del veng, vcr, vst
        except Exception as ex:       # noqa: BLE001
            out['config']['vawgan_config5'] = {'error': str(ex)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(arch, args.cpu_seconds)
    if rank == 0:
        # anything native libraries left in the C stdio buffer (RCCL prints a version banner with printf when a
        # communicator is created) goes out FIRST: the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
