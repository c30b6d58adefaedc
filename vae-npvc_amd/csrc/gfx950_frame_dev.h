// gfx950_frame_dev.h -- DEVICE runners of the phase-structured frame kernels (gfx950_frame.h, gfx950_frame_wgrad.h):
// what a "phase" is on the GPU.  Shared by gfx950_frame.hip (ConvVAE small-batch path) and disc.hip (the critic's thin
// conv layers, same scheme).  The host emulation of the same headers has its own runners (tests/frame_emu/frame_emu.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include "gfx950_frame.h"
#include "gfx950_frame_wgrad.h"

namespace vaenpvc {
namespace tuned {
using namespace frame;

// PROF: developer instrumentation (VAENPVC_FRAME_PROF=1), a separate instantiation: shader-clock stamp of every phase
// boundary of block 0.  The product instantiation has NO members: the runner is handed to the stage functions by
// reference, and a member would be re-read from memory (scratch) behind every barrier.
template <bool PROF>
struct DevRunnerT;
template <>
struct DevRunnerT<false> {
  __device__ __forceinline__ DevRunnerT(long long*, int) {}
  __device__ __forceinline__ void stamp() {}
};
template <>
struct DevRunnerT<true> {
  long long* prof;
  int n;
  __device__ __forceinline__ DevRunnerT(long long* p, int n0) : prof(p), n(n0) {}
  __device__ __forceinline__ void stamp() {
    if (blockIdx.x == 0 && threadIdx.x == 0 && n < 511) prof[++n] = clock64();
  }
};
template <bool PROF>
struct DevRunner : DevRunnerT<PROF> {
  __device__ __forceinline__ DevRunner(long long* p, int n0) : DevRunnerT<PROF>(p, n0) {}
  using DevRunnerT<PROF>::stamp;
  // Workgroup barrier that orders LDS only: __syncthreads() also waits for every outstanding global STORE (vmcnt(0)),
  // i.e. a full HBM write round trip per phase, and nothing a pass writes to HBM is read again inside the pass.
  __device__ __forceinline__ void sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp();
  }
  template <class F>
  __device__ __forceinline__ void phase(F&& f) {
    f((int)threadIdx.x);
    sync();
  }
  // dst[wave] = sum over the wave's lanes of f(tid); a workgroup-wide sum is sum16(dst) in the next phase
  template <class F>
  __device__ __forceinline__ void reduce(float* dst, F&& f) {
    float v = f((int)threadIdx.x);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) dst[threadIdx.x >> 6] = v;
    sync();
  }
  template <class F>
  __device__ __forceinline__ void reduce2(float* da, float* db, F&& f) {
    float a = 0.f, b = 0.f;
    f((int)threadIdx.x, a, b);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    if ((threadIdx.x & 63) == 0) {
      da[threadIdx.x >> 6] = a;
      db[threadIdx.x >> 6] = b;
    }
    sync();
  }
};
// The pass's argument block is copied ONCE from the kernel-argument segment into LDS and the stages read it there.
// (Handing the by-value kernel argument to the stage functions by reference made every lane keep a private copy:
//  ~0.6 MB of scratch writes per workgroup, 16 us at the head of each pass.)
template <class A>
__device__ __forceinline__ const A& args_to_lds(float* lds) {
  static_assert(sizeof(A) % 4 == 0 && sizeof(A) <= ARGS_FLOATS * 4, "argument block");
  const unsigned* ka = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();   // first argument: offset 0
  unsigned* la = reinterpret_cast<unsigned*>(lds + L_ARGS);
  if (threadIdx.x < sizeof(A) / 4) la[threadIdx.x] = ka[threadIdx.x];
  __syncthreads();
  return *reinterpret_cast<const A*>(la);
}
// runner of the job-list kernels (256-thread workgroups; gfx950_frame_wgrad.h)
struct WRunner {
  template <class F>
  __device__ __forceinline__ void phase(F&& f) {
    f((int)threadIdx.x);
    __syncthreads();
  }
  // per-thread accumulators across a loop over trips of `fb` frames: stage (all threads) | barrier | accumulate | barrier;
  // `spill` runs once per thread after the last trip, then a barrier
  template <class St, class Z, class Ac, class Sp, class AccT>
  __device__ __forceinline__ void frames(int f0, int f1, int fb, St&& st, Z&& z, Ac&& ac, Sp&& sp, AccT& acc) {
    const int tid = (int)threadIdx.x;
    z(tid, acc);
    for (int f = f0; f < f1; f += fb) {
      const int n = f1 - f < fb ? f1 - f : fb;
      st(tid, f, n);
      __syncthreads();
      ac(tid, acc, n);
      __syncthreads();
    }
    sp(tid, acc);
    __syncthreads();
  }
};

}  // namespace tuned
}  // namespace vaenpvc
