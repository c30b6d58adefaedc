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
losses, 2e-4 gradients; measured 1.4e-5 / <= 3.0e-5 at 32 768 frames, gpurun_out/parity_report.txt).  `modes`
reports the fp32-exact 3-term variant (with its own `roofline`) and the plain-bf16 mode (BASELINE config 2's
literal dtype; tolerance 3e-2, stated in the tests) beside it.  `config.convert_config4` = the conversion
path (encode -> decode, configs[3]) on the GPU next to the CPU leg of `cpu_baseline`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: bench.py launches its N ranks ITSELF
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`,
one process per GPU, RCCL).  It refuses to run (non-zero exit status) when fewer than N devices are visible
or when the process group does not have exactly N ranks.  Launched by torch.distributed.run directly (the
driver's form) it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

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
# Kernel GROUPS of the layered train step (roofline.sites): every tagged launch site of csrc/gfx950_layers.hip belongs to one;
# each group is timed in its own short pass (HIP events around every launch of the group, weight-gradient stream serialised).
#   bound 'mfma': algorithmic FLOP per frame = 3 (forward, input gradient, weight gradient) x 2 x effective MACs of the group's
#                 layers, priced against the bf16 matrix-core peak / products of the operand split;
#   bound 'hbm' : bytes per frame that the group's kernels read + write ONCE through their own interfaces (fp32 tensors),
#                 priced against 8 TB/s.  (The separate LayerNorm-backward passes exist only because layers are materialised:
#                 their bytes are moved bytes, not algorithmic ones -- stated in the entry.)
_E = dict(x=513, e0=2736, e1=1824, e2=1216, e3=896, e4=768, d0=1824, d1=2736, d2=4104)
SITE_GROUPS = {
    'tap_layer (1025-tap conv_transpose: forward, input gradient, weight gradient)': dict(
        tags='dec3_fwd dec3_dgrad dec3_wgrad dec3_row512 dec3_bias dxh_post', bound='mfma', mac=2105352),
    'dense_shaped (encoder layer 4, heads, sampler, merge: GEMMs + their plane producers)': dict(
        tags=('enc4_split enc4_fwd stats_enc4 heads_split heads_fwd reparam merge_split merge_fwd loss merge_dsplit merge_wgrad merge_segsum '
              'merge_small merge_dgrad reparam_bwd heads_dsplit heads_wgrad heads_dgrad enc4_dsplit enc4_wgrad enc4_dgrad stats_enc3'),
        bound='mfma', mac=688128 + 196608 + 196992),
    'mid_conv (encoder layers 2-3, decoder layer 0)': dict(
        tags=('enc2_split enc2_fwd stats_enc2 enc3_split enc3_fwd dec0_split dec0_fwd stats_dec0 dec0_gsplit dec0_asplit dec0_wgrad dec0_dgrad '
              'enc3_gsplit enc3_asplit enc3_wgrad enc3_dgrad enc2_gsplit enc2_asplit enc2_wgrad enc2_dgrad'),
        bound='mfma', mac=272384 + 401408 + 427680),
    'thin_conv (encoder layers 0-1, decoder layers 1-2: fused conv kernels, fused layer-backward kernels)': dict(
        tags=('enc0_fwd enc1_split enc1_fwd stats_enc1 dec1_split dec1_fwd stats_dec1 dec2_split dec2_fwd dec2_stats_planes dec2_bwd dec1_bwd enc1_bwd '
              'dec2_gsplit dec2_asplit dec2_wgrad dec2_dgrad dec1_gsplit dec1_asplit dec1_wgrad dec1_dgrad enc1_gsplit enc1_asplit enc1_wgrad '
              'enc1_dgrad enc0_wgrad enc0_bwd enc0_reduce lnb_dec2 lnb_dec1 lnb_enc0'),
        bound='hbm',
        bytes=4 * ((_E['x'] + _E['e0']) + (_E['e0'] + _E['e1']) + (_E['d0'] + _E['d1']) + (_E['d1'] + _E['d2']) + (_E['d2'] + 4224)
                   + (2 * _E['d2'] + 2 * _E['d1']) + (2 * _E['d1'] + _E['d0']) + (2 * _E['e1'] + 2 * _E['e0'])
                   + (_E['e0'] + _E['x']))
              + 2 * 2 * 63 * 32),   # (decoder layer 1's backward writes layer 0's gradient planes, 2 x bf16 [63][32], instead of the fp32 d(y0))
    'layernorm_backward (separate LayerNorm + lrelu backward passes: encoder layers 3-4; second-stage reductions of the fused ones)': dict(
        tags='lnb_dec0 lnb_enc4 lnb_enc3 lnb_enc2 lnb_enc1', bound='hbm',
        # reads: d(activated output) + pre-LN tensor (fp32); writes: the consumers' operand planes (2 x bf16; channel-last with halo rows for
        # decoder 0: 63 x 32, encoder 3: 10 x 128; plain rows for encoder 4) or the fp32 gradient (encoder 2)
        # (round 5: encoder layer 2's pass is gone -- it runs in the epilogue of layer 3's input-gradient GEMM; `lnb_enc2` is its 8 us second stage)
        # (... and decoder layer 0's inside decoder layer 1's fused backward kernel: `lnb_dec0` is its 7 us second stage)
        bytes=4 * 2 * (_E['e4'] + _E['e3']) + 2 * 2 * (10 * 128 + _E['e4']),
        moved_not_algorithmic=True),
    'weight_packing (per-step packed / split copies of the parameters)': dict(tags='prep', bound='hbm', bytes_per_step=10 * 939162 * 4),
}


# Rows of the large-batch kernel trace that can LEAD it (rocprofv3 --kernel-trace --stats of this command, profiles/r05_kernel_trace_stats.txt):
# kernel name as the profiler prints it (%d = operand planes), the launch sites it serves, its bound and its ALGORITHMIC work per frame.
# bench.py times every row (HIP events around each launch, weight-gradient stream serialised) and reports the row with the largest
# time per step as the top-level `roofline` -- the dominant kernel is measured, not chosen.
KERNEL_ROWS = [
    dict(name='k_gemm_nt_ring', tags='enc4_fwd heads_fwd enc4_dgrad heads_dgrad merge_dgrad', bound='mfma', mac=2 * 688128 + 2 * 196608 + 196992,
         what='C = A B^T on the four-wave LDS-DMA ring kernel (round 6): encoder layer 4 as a dense layer and the heads, forward + input gradient, '
              'and the merge input gradient: five launches per step'),
    dict(name='k_gemm_nt<%d, false>', tags='merge_fwd', bound='mfma', mac=196992,
         what='C = A B^T plane GEMM on the two-barrier 128 x 128 loop: the merge forward (K = 128, speaker table in the epilogue: a store stream)'),
    dict(name='k_fbwd<%d, 0, 0, 516, false>', tags='dec2_bwd', bound='hbm', bytes=4 * (_E['d2'] + _E['d2'] + _E['d1'] + _E['d1']),
         what='whole backward step of decoder layer 2 in one kernel: dy + pre-LN output + input activation read, input gradient written, once'),
    dict(name='k_fbwd<%d, 2, 0, 0, false>', tags='enc1_bwd', bound='hbm', bytes=4 * (_E['e1'] + _E['e1'] + _E['e0'] + _E['e0']),
         what='whole backward step of encoder layer 1'),
    dict(name='k_toep_gemm_bf16<false, %d, false, 516, false>', tags='dec3_dgrad', bound='mfma', mac=8 * 513 * 513, what='1025-tap layer, input gradient'),
    dict(name='k_toep_gemm_bf16<true, %d, false, 513, true>', tags='dec3_fwd', bound='mfma', mac=8 * 513 * 513,
         what='1025-tap layer, forward, with LayerNorm + lrelu + operand split of its input in the staging (round 6: no producer pass in front)'),
    dict(name='k_toep_wgrad_bf16_w4<%d>', tags='dec3_wgrad', bound='mfma', mac=8 * 513 * 513, what='1025-tap layer, weight gradient'),
    dict(name='k_fbwd<%d, 1, 0, 0, true>', tags='dec1_bwd', bound='hbm', bytes=4 * (_E['d1'] + _E['d1'] + _E['d0']) + 2 * 2 * 63 * 32,
         what='whole backward step of decoder layer 1 + the LayerNorm backward of layer 0 (result: its bf16 operand planes)'),
]
# layer-materialised (Model B) bytes per frame of the thin conv group's tensors: outputs of encoder 0-1 and decoder 1-2, each
# written + read once forward and its gradient written + read once backward, x read twice
THIN_MODEL_B_BYTES = 4 * 4 * (_E['e0'] + _E['e1'] + _E['d1'] + _E['d2']) + 2 * 4 * _E['x']


def parse(argv=None):
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
    p.add_argument('--no-convert', action='store_true', help='skip the conversion-path (config 4) measurement')
    p.add_argument('--no-traffic', action='store_true',
                   help='do not measure step_traffic in the run (two rocprofv3 --pmc passes of a 4-step child of this command when rocprofv3 '
                        'is on the box); the committed constant of profiles/pmc_traffic.json is reported instead, labelled as such')
    p.add_argument('--headline-only', action='store_true', help='(the child of the traffic passes) time the steps, print the line, no side legs')
    p.add_argument('--all-legs', action='store_true',
                   help='N > 1: also run the legs that no multi-rank RCCL run has exercised yet (hipGraph capture of a step with '
                        'its all-reduce, the VAWGAN iteration); by default they are skipped there and said so in the line')
    p.add_argument('--side-leg-seconds', type=float, default=900.0,
                   help='watchdog of everything after the headline measurement: when it expires rank 0 prints the line it has '
                        '(with "side_legs": "timed out ...") and every rank exits')
    p.add_argument('--master-port', type=int, default=0, help='rendezvous port of the self-launched ranks (0 = pick a free one)')
    p.add_argument('--standin', default=None,
                   help='TEST ONLY: python file providing make_engine(arch, args) -> CPU stand-in engine; the ranks then use '
                        'the gloo backend and only the launcher / process-group / timing logic of this file runs')
    return p.parse_args(argv)


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


# algorithmic figures of the conversion path: encode (z_mu only) + decode, forward only
FLOP_PER_FRAME_CONVERT = 9.419e6
BYTES_PER_FRAME_CONVERT = 150384.0


def measure_step_traffic(F, precision, seconds=240):
    """HBM-side bytes of ONE train step, measured in this run: two `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE -- together they exceed the TCC counter slots; never combined with a sys / runtime trace) of a 4-step
    `--headline-only` child of this command, summed over every kernel dispatch of the trace and divided by the number of steps
    (= launches of the optimiser kernel).  Corrections as guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
    FETCH_SIZE x 2 (a 128-B request is tallied as 64 B), both counters in KiB.  Returns None when rocprofv3 is absent, a pass
    fails or runs out of time: the caller then keeps the committed constant, labelled measured_in_run = false."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if not exe:
        return None
    tot, steps_seen = {}, None
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='vaenpvc_pmc_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp', VAENPVC_SIDE_STREAM='0')
            cmd = [exe, '--kernel-trace', '--pmc', ctr, '-d', d, '--', sys.executable, os.path.abspath(__file__), '--headline-only',
                   '--no-traffic', '--frames', str(F), '--precision', precision, '--steps', '3', '--warmup', '1']
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=seconds / 2)
            dbs = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            c = sqlite3.connect(dbs[0])
            ks = [row[1] for row in c.execute('pragma table_info(rocpd_info_kernel_symbol)')]
            name_col = 'display_name' if 'display_name' in ks else 'kernel_name'
            q = ('select s.%s, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id '
                 'join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id '
                 'where p.name = ?' % name_col)
            kib, adam = 0.0, set()
            for name, val in c.execute(q, (ctr,)):
                kib += val
            adam = c.execute('select count(*) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id '
                             "where s.%s like '%%k_adam%%'" % name_col).fetchone()[0]
            c.close()
            if not adam:
                return None
            tot[ctr], steps_seen = kib / adam, adam
        except (subprocess.TimeoutExpired, OSError, sqlite3.Error):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    nbytes = (2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0
    return {'hbm_bytes_per_step': nbytes, 'fetch_kib_per_step': tot['FETCH_SIZE'], 'write_kib_per_step': tot['WRITE_SIZE'],
            'steps_in_trace': steps_seen,
            'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) of a 4-step child of this command, in this run: '
                      'sum over all kernel dispatches of 2 x FETCH_SIZE + WRITE_SIZE (KiB) / steps'}


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices(args):
    if args.standin:
        return args.gpus          # the CPU stand-in ranks need no device
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): start the N ranks with torch.distributed.run and
    hand its exit status back.  The child ranks run this same file with WORLD_SIZE set."""
    import subprocess
    nvis = visible_devices(args)
    if nvis < args.gpus:
        sys.stderr.write('bench.py: --gpus %d but only %d device(s) visible: refusing to run\n' % (args.gpus, nvis))
        return 3
    port = args.master_port or free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this driver (RCCL needs it)
    env['VAENPVC_BENCH_SELF_LAUNCHED'] = '1'
    return subprocess.call(cmd, env=env)


def load_standin(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_standin', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args, argv))
    import torch
    import torch.distributed as dist
    from hipvae.dp import Stepper

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        # one rank asked for N, or N ranks asked for another N: never print a line that claims a size it did not run
        sys.stderr.write('bench.py: --gpus %d does not match WORLD_SIZE %d: refusing to run\n' % (args.gpus, world))
        raise SystemExit(3)
    standin = load_standin(args.standin) if args.standin else None
    if standin is None:
        from hipvae import Engine
        from hipvae import lib as L
        nvis = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if nvis < world or local >= nvis:
            sys.stderr.write('bench.py: %d rank(s) but %d device(s) visible: refusing to run\n' % (world, nvis))
            raise SystemExit(3)
        torch.cuda.set_device(local)
    force_dist = os.environ.get('VAENPVC_FORCE_DIST') == '1'      # exercise the RCCL path with one rank
    backend = 'gloo' if standin else 'nccl'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if standin:
            dist.init_process_group(backend)
        else:
            dist.init_process_group(backend, device_id=torch.device('cuda', local))
        if dist.get_world_size() != args.gpus:
            sys.stderr.write('bench.py: process group has %d ranks, --gpus %d: refusing to run\n' % (dist.get_world_size(), args.gpus))
            raise SystemExit(3)

    with open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')) as fp:
        arch = json.load(fp)
    t = arch['training']
    if standin:
        arch = standin.ARCH if hasattr(standin, 'ARCH') else arch
        t = arch['training']
        eng = standin.make_engine(arch, args)
        planes = 2
    else:
        eng = Engine(arch, impl=args.impl, precision=args.precision)
        eng.init_params(seed=0)
        planes = L.PRECISIONS[args.precision]
    st = Stepper(eng, t['lr'], t['beta1'], t['beta2'], seed=0)
    st.broadcast_params()
    dev = eng.params.device
    H = arch['hwc'][0]

    def make_batch(F, seed):
        g = torch.Generator(device='cpu').manual_seed(seed)
        x = (torch.rand(F, H, generator=g) * 2 - 1).to(dev).to(eng.params.dtype)
        y = torch.randint(0, arch['y_dim'], (F,), generator=g, dtype=torch.int64).to(dev)
        return x, y

    def sync():
        if dev.type == 'cuda':
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    def max_over_ranks(dt):
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

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
                # duration of the SITE per step: a site may be several launches (the 1025-tap weight gradient is one launch per
                # q tile since round 3); `n` = launches counted
                kern = (ms / steps, n, steps)
        return max_over_ranks(dt), kern

    F = args.frames
    dt, kern = timed(F, args.steps, args.warmup, None if (args.impl == 'auto' or standin) else args.timer_tag)
    frames_per_s = world * F * args.steps / dt
    steps_per_s = args.steps / dt
    out = {
        'metric': 'SP frames/sec (train step: fwd+bwd+Adam%s)' % ('+RCCL all-reduce' if world > 1 else ''),
        'value': frames_per_s, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {3: 'f32 (tensors and accumulation; matrix-core operands split into 3 bf16 terms = fp32-exact)',
                  2: 'f32 tensors and accumulation / bf16x2 operands (16 mantissa bits) on the matrix cores',
                  1: 'bf16 operands on the matrix cores, f32 tensors and accumulation'}[planes],
        'arithmetic': ('fp32 tensors, fp32 accumulation; GEMM operands on the bf16 matrix cores split into bf16 terms: %s'
                       % {3: '3 terms (fp32-exact)',
                          2: '2 terms (16 mantissa bits per operand; holds the 1e-4 / 2e-4 parity bars at 32 768 frames: measured 1.4e-5 / 3.0e-5)',
                          1: 'plain bf16 operands: the reduced-precision bf16 mode'}[planes]),
        'data': 'synthetic (x~U(-1,1), y~randint(10) resident in HBM; eps~N(0,1) drawn on the device per step, Philox4x32-10; random-init weights)',
        'rccl_ranks': (dist.get_world_size() if dist.is_initialized() else 1),
        'launch': ('self-launched torch.distributed.run' if os.environ.get('VAENPVC_BENCH_SELF_LAUNCHED') == '1'
                   else 'torch.distributed.run (external)' if world > 1 else 'single process'),
        'config': {'workload': 'ConvVAE architecture-vae-vcc2016 train step, 256x[1,513,128] = %d frames/step/GPU' % F,
                   'frames_per_step_per_gpu': F, 'global_frames_per_step': F * world, 'impl': args.impl,
                   'precision': args.precision, 'parallelism': 'dp%d' % world, 'backend': backend if (world > 1 or force_dist) else None,
                   'all_reduce': ('four gradient buckets (3.76 MB fp32 in all, SUM) started from a library callback while the backward pass runs, '
                                  '1/N folded into Adam; losses ride in the buffer tail') if world > 1 else None},
        'step_fraction_of_rooflines': {
            'hbm_model_B': (frames_per_s / world * BYTES_PER_FRAME_TRAIN + steps_per_s * BYTES_PER_STEP_PARAMS) / HBM_PEAK,
            'mfma_%s' % PREC_NAME[planes]: frames_per_s / world * FLOP_PER_FRAME_TRAIN / (BF16_PEAK / PRODUCTS[planes]),
            'note': ('hbm_model_B: layer-materialised algorithmic bytes (SURVEY 8d) / step time / 8 TB/s; mfma_%s: 28.85 MFLOP per frame against the '
                     'bf16 matrix-core peak / the products of the operand split -- the instruction the GEMM-shaped kernels issue.  (The exact-fp32 '
                     'MFMA figure of earlier rounds is gone: no large-batch kernel issues that instruction any more, so it could pass 1.)'
                     % PREC_NAME[planes])},
    }
    if args.headline_only and not standin:
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if standin:
        out['data'] = 'CPU stand-in engine (test of the launcher / process-group logic only; not a measurement)'
        out['config']['workload'] = 'stand-in'
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    # ---- from here on: side legs.  The headline is measured; nothing below may swallow it.  A watchdog thread prints the line
    #      as it stands and ends the process when the side legs take longer than --side-leg-seconds (a hung collective in a
    #      leg that only one rank entered cannot be cancelled from Python)
    import threading

    def _emit(extra=None):
        if extra:
            out['side_legs'] = extra
        if rank == 0:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)      # (RCCL prints a banner with printf: the JSON line must be the LAST line)
            except Exception:
                pass
            print(json.dumps(out), flush=True)

    def _expired():
        try:
            _emit('timed out after %.0f s: the line holds what was measured until then' % args.side_leg_seconds)
        finally:
            os._exit(0)
    watchdog = threading.Timer(args.side_leg_seconds, _expired)
    watchdog.daemon = True
    watchdog.start()
    risky = world == 1 or args.all_legs      # legs never run under multi-rank RCCL (see --all-legs)
    # whole-step HBM traffic from the committed PMC passes of this command (sum over all kernels of
    # 2 x FETCH_SIZE + WRITE_SIZE; profiles/README.md), next to the layer-materialised algorithmic bytes
    alg_b = F * BYTES_PER_FRAME_TRAIN + BYTES_PER_STEP_PARAMS
    live = None
    if world == 1 and not args.no_traffic and args.impl == 'auto' and F >= 1024:
        sync()
        live = measure_step_traffic(F, args.precision)
    if live:
        out['step_traffic_bytes'] = live['hbm_bytes_per_step']
        out['step_traffic'] = dict(live, algorithmic_bytes_model_B=alg_b, ratio=live['hbm_bytes_per_step'] / alg_b, measured_in_run=True)
    else:
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fp:
                tj = json.load(fp).get('step/%s' % ('bf16x2' if args.precision == 'auto' else args.precision))
            if tj and tj.get('frames') == F:
                out['step_traffic_bytes'] = tj['hbm_bytes_per_step']
                out['step_traffic'] = {'hbm_bytes_per_step': tj['hbm_bytes_per_step'], 'algorithmic_bytes_model_B': alg_b,
                                       'ratio': tj['hbm_bytes_per_step'] / alg_b, 'measured_in_run': False, 'source': tj.get('source')}
        except (OSError, ValueError):
            pass

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

    def roofline_of(prec_name, npl, kern):
        avg_ms, n, nsteps = kern
        ach = DEC3_FLOP_PER_FRAME * F / (avg_ms * 1e-3) / 1e12
        # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (separate
        # --pmc runs of this same command, FETCH_SIZE/WRITE_SIZE in KiB; see profiles/README.md)
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fp:
                tj = json.load(fp).get('%s/%s' % (args.timer_tag, prec_name))
                if tj and tj.get('frames') == F:
                    traffic = tj['hbm_bytes_per_launch']
        except (OSError, ValueError):
            pass
        bf16 = args.timer_tag in ('dec3_fwd', 'dec3_dgrad', 'dec3_wgrad') and F >= 16 and args.impl == 'auto'
        peak = (BF16_PEAK / PRODUCTS[npl] if bf16 else FP32_PEAK) / 1e12
        return {'bound': 'mfma', 'kernel': args.timer_tag, 'precision': prec_name, 'achieved': ach, 'peak': peak,
                'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
                'avg_kernel_ms': avg_ms, 'launches': n, 'launches_per_step': n / nsteps,
                'avg_kernel_ms_note': 'duration of the kernel site per step = sum of its launches (the 1025-tap weight gradient: one launch per q tile)',
                'algorithmic_flops_per_launch': DEC3_FLOP_PER_FRAME * F,
                'peak_basis': ('dense bf16 MFMA peak / %d (%d-term operand split: %d bf16 products per fp32 product)'
                               % (PRODUCTS[npl], npl, PRODUCTS[npl]) if bf16
                               else 'exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'),
                'measured': 'HIP events on the launch stream, side stream serialised'}
    kern = kernel_ms(args.timer_tag) if args.impl == 'auto' else kern
    if kern:
        out['roofline'] = roofline_of('bf16x2' if args.precision == 'auto' else args.precision, planes, kern)
        # ---- which row LEADS the kernel trace: every candidate row timed, the largest reported as the top-level roofline (round 5;
        #      until round 4 the line advertised dec3_wgrad, the best kernel of the step, which is row 8 of the trace)
        if args.impl == 'auto' and F >= 8192 and args.timer_tag == 'dec3_wgrad':
            prec_name = 'bf16x2' if args.precision == 'auto' else args.precision
            try:
                with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fp:
                    tjall = json.load(fp)
            except (OSError, ValueError):
                tjall = {}
            rows = []
            for r in KERNEL_ROWS:
                k = kern if r['tags'] == 'dec3_wgrad' else kernel_ms(','.join(r['tags'].split()), 4)
                if not k:
                    continue
                ms = k[0]
                name = 'void ' + (r['name'] % planes if '%d' in r['name'] else r['name'])
                ent = {'kernel': name, 'sites': r['tags'].split(), 'what': r['what'], 'ms_per_step': ms, 'launches_per_step': k[1] / k[2],
                       'avg_launch_ms': ms * k[2] / k[1], 'bound': r['bound'], 'precision': prec_name}
                if r['bound'] == 'mfma':
                    fl = 2.0 * r['mac'] * F
                    ent.update(achieved=fl / (ms * 1e-3) / 1e12, peak=BF16_PEAK / PRODUCTS[planes] / 1e12, unit='TFLOP/s',
                               algorithmic_flops_per_step=fl,
                               peak_basis='dense bf16 MFMA peak / %d (%d-term operand split)' % (PRODUCTS[planes], planes))
                else:
                    nb = float(r['bytes']) * F
                    ent.update(achieved=nb / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s', algorithmic_bytes_per_step=nb,
                               peak_basis='HBM3E 8 TB/s')
                ent['frac'] = ent['achieved'] / ent['peak']
                tj = tjall.get('row/%s/%s' % (name, prec_name)) or tjall.get('row/%s/%s' % (name[5:], prec_name))
                ent['traffic'] = tj['hbm_bytes_per_step'] if (tj and tj.get('frames') == F) else None
                ent['traffic_measured_in_run'] = False
                rows.append(ent)
            rows.sort(key=lambda e: -e['ms_per_step'])
            if rows:
                best_kernel = out['roofline']       # dec3_wgrad: the best GEMM of the step, kept beside the dominant one
                top = dict(rows[0])
                top.update(avg_kernel_ms=top['ms_per_step'], launches=int(round(top['launches_per_step'])),
                           measured='HIP events on the launch stream around every launch of the row, weight-gradient stream serialised',
                           selected='largest time per step among the candidate trace rows (trace_rows); agrees with the top row of '
                                    'profiles/r06_kernel_trace_stats.txt')
                top['trace_rows'] = rows
                top['best_kernel'] = best_kernel
                out['roofline'] = top
        if args.impl == 'auto' and F >= 8192 and args.timer_tag == 'dec3_wgrad':
            # the two sibling GEMMs of the same layer (same algorithmic flops)
            sib = {}
            pk = BF16_PEAK / PRODUCTS[planes] / 1e12
            bk = out['roofline'].get('best_kernel')
            if bk:
                sib['dec3_wgrad'] = {'avg_kernel_ms': bk['avg_kernel_ms'], 'achieved_tflops_fp32_equiv': bk['achieved'], 'peak': pk,
                                     'frac': bk['frac'], 'traffic': bk.get('traffic')}
            for r in out['roofline'].get('trace_rows', []):
                if r['sites'] in (['dec3_fwd'], ['dec3_dgrad']):
                    sib[r['sites'][0]] = {'avg_kernel_ms': r['ms_per_step'], 'achieved_tflops_fp32_equiv': r['achieved'], 'peak': pk, 'frac': r['frac']}
            for tag in ('dec3_fwd', 'dec3_dgrad'):
                if tag in sib:
                    continue
                k = kernel_ms(tag, 4)
                if k:
                    a2 = DEC3_FLOP_PER_FRAME * F / (k[0] * 1e-3) / 1e12
                    sib[tag] = {'avg_kernel_ms': k[0], 'achieved_tflops_fp32_equiv': a2, 'peak': pk, 'frac': a2 / pk}
            out['roofline']['sibling_kernels'] = sib
    # ---- roofline.sites: what the step is made of.  One short pass per kernel group (SITE_GROUPS), the five largest by time
    #      reported with their own bound; `serialised_ms_per_step` = the same step with the weight-gradient stream serialised
    #      (the sum the groups add up to; `value` is measured with the two streams overlapping)
    if kern and args.impl == 'auto' and F >= 1024:
        eng.set_tuned_masks(0xffffffff, 0xbfffffff)
        try:
            ser, _ = timed(F, 6, 2)
            groups = []
            for name, gdef in SITE_GROUPS.items():
                _, k = timed(F, 3, 1, ','.join(gdef['tags'].split()))
                if not k:
                    continue
                ms = k[0]
                ent = {'group': name, 'ms_per_step': ms, 'launches_per_step': k[1] / k[2], 'bound': gdef['bound']}
                if gdef['bound'] == 'mfma':
                    ach = 3 * 2.0 * gdef['mac'] * F / (ms * 1e-3) / 1e12
                    ent.update(achieved=ach, peak=BF16_PEAK / PRODUCTS[planes] / 1e12, unit='TFLOP/s (fp32-equivalent)',
                               algorithmic_flops_per_step=3 * 2.0 * gdef['mac'] * F)
                else:
                    nbytes = gdef.get('bytes_per_step', gdef.get('bytes', 0) * F)
                    ach = nbytes / (ms * 1e-3) / 1e9
                    ent.update(achieved=ach, peak=HBM_PEAK / 1e9, unit='GB/s', bytes_per_step=nbytes,
                               bytes_are=('moved by passes that exist only because layers are materialised (not algorithmic)'
                                          if gdef.get('moved_not_algorithmic') else 'what the kernels of the group read + write once through their interfaces'))
                    # the same time priced on the ALGORITHMIC (layer-materialised, SURVEY 8d Model B) bytes of the group's tensors
                    if name.startswith('thin_conv'):
                        ent['frac_model_B'] = THIN_MODEL_B_BYTES * F / (ms * 1e-3) / HBM_PEAK
                        ent['model_B_bytes_per_step'] = THIN_MODEL_B_BYTES * F
                    elif gdef.get('moved_not_algorithmic'):
                        ent['frac_model_B'] = 0.0       # no algorithmic bytes: these passes only exist because layers are materialised
                    elif 'bytes_per_step' in gdef:
                        ent['frac_model_B'] = ent['achieved'] / ent['peak']
                ent['frac'] = ent['achieved'] / ent['peak']
                groups.append(ent)
            groups.sort(key=lambda e: -e['ms_per_step'])
            tot = sum(e['ms_per_step'] for e in groups)
            out['roofline']['sites'] = groups[:5]
            out['roofline']['sites_note'] = {
                'groups_sum_ms': tot, 'groups_beyond_the_five_ms': sum(e['ms_per_step'] for e in groups[5:]),
                'serialised_ms_per_step': ser / 6 * 1e3, 'untagged_ms': ser / 6 * 1e3 - tot,
                'ms_per_step': out['ms_per_step'],
                'measured': 'HIP events around every launch of the group, one pass per group, weight-gradient stream serialised; '
                            'untagged = Adam, fills, host gaps; ms_per_step is lower than the serialised sum by what the second stream overlaps'}
        finally:
            eng.set_tuned_masks(0xffffffff, 0xffffffff)
    # ---- the other precisions beside the default (never instead of it)
    if not args.no_modes and args.impl == 'auto':
        cur = 'bf16x2' if args.precision == 'auto' else args.precision
        modes = {cur: {'ms_per_step': dt / args.steps * 1e3, 'frames_per_s': frames_per_s}}
        for prec in ('bf16x2', 'bf16x3', 'bf16'):
            if prec in modes:
                continue
            eng.set_precision(prec)
            n2 = max(10, args.steps // 4)
            d2, _ = timed(F, n2, 3)
            fps2 = world * F * n2 / d2
            modes[prec] = {'ms_per_step': d2 / n2 * 1e3, 'frames_per_s': fps2,
                           'fraction_of_hbm_model_B': (fps2 / world * BYTES_PER_FRAME_TRAIN + n2 / d2 * BYTES_PER_STEP_PARAMS) / HBM_PEAK}
            if prec == 'bf16':
                # the bf16-activation reading of the layer-materialised model (152 432 B/frame)
                modes[prec]['fraction_of_hbm_model_B_bf16_activations'] = (fps2 / world * 152432.0 + n2 / d2 * BYTES_PER_STEP_PARAMS) / HBM_PEAK
            if prec == 'bf16x3' and args.timer_tag == 'dec3_wgrad':
                k3 = kernel_ms(args.timer_tag, 4)      # the fp32-exact mode's own roofline line
                if k3:
                    out['roofline_bf16x3'] = roofline_of('bf16x3', 3, k3)
                # the thin-conv group (HBM-bound, the largest group of the step) at the reference's own precision, on the same two byte models
                tname = next(n for n in SITE_GROUPS if n.startswith('thin_conv'))
                kt = kernel_ms(','.join(SITE_GROUPS[tname]['tags'].split()), 3)
                if kt:
                    modes[prec]['thin_conv'] = {'ms_per_step': kt[0], 'frac': SITE_GROUPS[tname]['bytes'] * F / (kt[0] * 1e-3) / HBM_PEAK,
                                                'frac_model_B': THIN_MODEL_B_BYTES * F / (kt[0] * 1e-3) / HBM_PEAK, 'bound': 'hbm'}
        eng.set_precision(args.precision)
        modes['note'] = ('bf16x2 (default): 2-term operand split; bf16x3: 3 terms, fp32-exact; '
                         'bf16: plain bf16 operands on the kernels that run on the bf16 matrix cores (tolerance 3e-2, tests)')
        out['modes'] = modes
        # the fp32-exact figure beside the headline (the default feeds the matrix cores 16-mantissa-bit operand pairs)
        if 'bf16x3' in modes:
            out['reference_precision_value'] = {'value': modes['bf16x3']['frames_per_s'], 'unit': 'frames/s',
                                                'ms_per_step': modes['bf16x3']['ms_per_step'],
                                                'precision': 'bf16x3 = 3-term operand split, fp32-exact (measured error vs float64: 2e-6)'}
            out['headline_choice'] = (
                '`value` (2-term operands) is the config-2 headline: BASELINE.json configs[1] names bf16 and north_star asks for 1e-4 '
                'activation / loss parity, which this mode holds at the benchmarked size in the GPU gate (activations 1.4e-5, losses 1e-7, '
                'gradients 3e-5 of their scale on kink-safe data; the lrelu kink flips it adds on plain data are counted and bounded: '
                '2.5e-6 of the units).  `reference_precision_value` (3-term operands) is the figure for a reader who wants arithmetic '
                'indistinguishable from the reference\'s fp32 kernels (2e-6); `modes.bf16` is config 2\'s literal dtype (tolerance 3e-2).')
    if not args.no_literal:
        # the literal batch sizes: 256 frames per GPU (configs[1]; with N = 8 ranks the global batch is configs[2]'s
        # 2048) and 16 (configs[0], the reference's own batch_size)
        lits = {}
        for Fl, nst in ((256, 200), (16, 200)):
            dt2, _ = timed(Fl, nst, 10)
            lit = {'frames_per_s': world * Fl * nst / dt2, 'ms_per_step': dt2 / nst * 1e3, 'launch': 'eager',
                   'frames_per_step_per_gpu': Fl, 'global_frames_per_step': Fl * world,
                   # these batch sizes run the frame kernels (fp32 vector arithmetic, one workgroup per frame: DESIGN.md section 11):
                   # priced against the packed-fp32 VECTOR peak (= the exact-fp32 MFMA figure, 157.3 TFLOP/s) and the HBM model
                   'path': ('frame kernels: 7 launches per step (the 1025-tap layer as two launches of its own)' if Fl <= 128 else
                            'frame kernels: 6 launches per step' if Fl <= 512 else 'layered kernels'),
                   'fraction_of_fp32_vector_peak': Fl * nst / dt2 * FLOP_PER_FRAME_TRAIN / FP32_PEAK,
                   'fraction_of_hbm_model_B': (Fl * nst / dt2 * BYTES_PER_FRAME_TRAIN + nst / dt2 * BYTES_PER_STEP_PARAMS) / HBM_PEAK}
            # same step captured in a hipGraph (one launch per step instead of one per kernel); with N > 1 the
            # (unbucketed) gradient all-reduce is captured with it
            try:
                if not risky:
                    raise RuntimeError('skipped at N > 1 (--all-legs runs it)')
                x, y = make_batch(Fl, 99)
                st.capture(x, y)
                for _ in range(10):
                    st.replay()
                barrier()
                t0 = time.perf_counter()
                for _ in range(nst):
                    st.replay()
                barrier()
                dtg = max_over_ranks(time.perf_counter() - t0)
                lit['hipgraph'] = {'frames_per_s': world * Fl * nst / dtg, 'ms_per_step': dtg / nst * 1e3}
                # K = 8 consecutive steps per graph (Stepper.capture(steps=8), eight static batches): the graph launch's own fixed cost,
                # which makes the one-step graph slower than the eager launches at these sizes, is shared by eight steps
                GK = 8
                xs, ys = zip(*[make_batch(Fl, 200 + i) for i in range(GK)])
                st.capture(torch.stack(xs), torch.stack(ys), steps=GK)
                for _ in range(3):
                    st.replay()
                barrier()
                t0 = time.perf_counter()
                for _ in range(nst // GK):
                    st.replay()
                barrier()
                dtk = max_over_ranks(time.perf_counter() - t0)
                nk = (nst // GK) * GK
                lit['hipgraph_x8'] = {'frames_per_s': world * Fl * nk / dtk, 'ms_per_step': dtk / nk * 1e3, 'steps_per_graph': GK}
            except Exception as ex:       # noqa: BLE001  (capture support differs between RCCL builds)
                lit.setdefault('hipgraph', {'error': str(ex)[:200]})
                lit.setdefault('hipgraph_x8', {'error': str(ex)[:200]})
            # the trainer's own hot loop (trainer/vae.py:94-99: shuffle_batch dequeue -> sess.run(opt['g'])): the same step fed by
            # the frame store's dequeue (host-side index draw + one gather / normalise kernel on resident records) instead of
            # a resident batch
            try:
                import numpy as np
                from analyzer import FrameStore, Tanhize
                rng = np.random.default_rng(5 + rank)
                recs = rng.standard_normal((20000, 1029)).astype(np.float32)
                recs[:, :513] = rng.uniform(-12, -3, (20000, 513))
                recs[:, -1] = rng.integers(0, arch['y_dim'], 20000)
                store = FrameStore(recs, Fl, Tanhize(xmax=np.full(513, -3, np.float32), xmin=np.full(513, -12, np.float32)),
                                   seed=11, rank=0, world=1, device=dev, capacity=256, min_after_dequeue=128, y_dim=arch['y_dim'])
                for _ in range(10):
                    st.step(*store.next_batch())
                barrier()
                t0 = time.perf_counter()
                for _ in range(nst):
                    st.step(*store.next_batch())
                barrier()
                dtl = max_over_ranks(time.perf_counter() - t0)
                lit['with_dequeue'] = {'ms_per_step': dtl / nst * 1e3, 'frames_per_s': world * Fl * nst / dtl}
                del store
            except Exception as ex:       # noqa: BLE001
                lit['with_dequeue'] = {'error': str(ex)[:200]}
            lits['F%d' % Fl] = lit
        out['config']['literal_batches'] = lits
        out['config']['literal_batch256'] = lits['F256']
    if not args.no_convert and args.impl == 'auto':
        # BASELINE.json configs[3]: the conversion path convert.py:79-89 (encode -> z_mu, decode towards speaker 9 = TM3),
        # forward only, frames resident in HBM; every rank converts its own frames
        conv = {}
        for Fc, nit in ((300, 50), (1024, 50), (32768, 10)):
            xc, _ = make_batch(Fc, 4321 + rank)
            yc = torch.full((Fc,), 9, dtype=torch.int64, device=dev)
            for _ in range(3):
                eng.decode(eng.encode(xc), yc)
            barrier()
            t0 = time.perf_counter()
            for _ in range(nit):
                eng.decode(eng.encode(xc), yc)
            barrier()
            dtc = max_over_ranks(time.perf_counter() - t0) / nit
            fps = world * Fc / dtc
            conv['F%d' % Fc] = {'frames': Fc, 'ms': dtc * 1e3, 'frames_per_s': fps,
                                'fraction_of_fp32_flops': fps / world * FLOP_PER_FRAME_CONVERT / FP32_PEAK,
                                'fraction_of_mfma_%s' % PREC_NAME[planes]: fps / world * FLOP_PER_FRAME_CONVERT / (BF16_PEAK / PRODUCTS[planes]),
                                'fraction_of_hbm_model_B': fps / world * BYTES_PER_FRAME_CONVERT / HBM_PEAK}
        # the utterance sizes convert.py actually runs (a few hundred to ~2 000 frames per file, convert.py:105-116) are latency-bound as
        # single launches; convert.py gathers consecutive files into one launch (--batch_frames, default 16 384: convert.convert_utterances).
        # Measured here: 16 synthetic utterances of 1 024 (resp. 54 of 300) frames converted in ONE encode -> decode call, per utterance
        for Fu, nu, nit in ((1024, 16, 20), (300, 54, 20)):
            xc, _ = make_batch(Fu * nu, 4321 + rank)
            yc = torch.full((Fu * nu,), 9, dtype=torch.int64, device=dev)
            for _ in range(3):
                eng.decode(eng.encode(xc), yc)
            barrier()
            t0 = time.perf_counter()
            for _ in range(nit):
                torch.split(eng.decode(eng.encode(xc), yc), Fu)      # (the cut back to utterances: views, no copy)
            barrier()
            dtc = max_over_ranks(time.perf_counter() - t0) / nit
            conv['F%d_batched_x%d' % (Fu, nu)] = {'frames_per_utterance': Fu, 'utterances_per_launch': nu, 'ms_per_launch': dtc * 1e3,
                                                  'ms_per_utterance': dtc * 1e3 / nu, 'frames_per_s': world * Fu * nu / dtc}
        conv['note'] = ('encode (z_mu) + decode, 9.419 MFLOP and 150 384 layer-materialised bytes per frame (SURVEY 8d); '
                        'the CPU leg of the same path is cpu_baseline.legs.convert_fwd_F1024; F300 / F1024 = ONE utterance per launch (the '
                        'reference\'s sess.run per file), F*_batched_x* = what convert.py does by default')
        out['config']['convert_config4'] = conv
    if not args.no_literal:
        # BASELINE.json configs[4]: the VAWGAN branch (nIterD critic steps + one generator step per iteration, 16 frames per
        # step and GPU; hipvae/adversarial.py, data parallel over the same process group).  Reported beside the headline.
        try:
            if not risky:
                raise RuntimeError('skipped at N > 1 (--all-legs runs it)')
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
            dtv = max_over_ranks(time.perf_counter() - t0)
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
    watchdog.cancel()
    _emit()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
