// misc_kernels.hip -- optimiser and data-plane kernels (HBM-bound, float4 streams).
#include "kernels.h"

namespace vaenpvc {

// tf.train.AdamOptimizer apply (trainer/vae.py:16-24): ONE pass over the
// flat buffers; reads g,p,m,v and writes p,m,v (28 B/param).
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps, float gs) {
  int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    float4 pp = *(float4*)(p + i4), gg = *(const float4*)(g + i4), mm = *(float4*)(m + i4), vv = *(float4*)(v + i4);
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w};
    float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = ga[k] * gs;
      ma[k] = b1 * ma[k] + (1.0f - b1) * gk;
      va[k] = b2 * va[k] + (1.0f - b2) * gk * gk;
      pa[k] = pa[k] - lr_t * ma[k] / (sqrtf(va[k]) + eps);
    }
    *(float4*)(p + i4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *(float4*)(m + i4) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *(float4*)(v + i4) = make_float4(va[0], va[1], va[2], va[3]);
  } else {
    for (int64_t i = i4; i < n; ++i) {
      float gk = g[i] * gs;
      float mk = b1 * m[i] + (1.0f - b1) * gk;
      float vk = b2 * v[i] + (1.0f - b2) * gk * gk;
      m[i] = mk;
      v[i] = vk;
      p[i] = p[i] - lr_t * mk / (sqrtf(vk) + eps);
    }
  }
}

void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
                 float eps, float gscale, hipStream_t s) {
  int64_t nt = (n + 3) / 4;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps,
                     gscale);
}

// graph-replayable variant: step counter in device memory
__global__ void k_step_inc(int64_t* d_step) { *d_step += 1; }

__global__ void k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                           float* __restrict__ v, int64_t n, const int64_t* __restrict__ d_step, float lr, float b1,
                           float b2, float eps, float gs) {
  __shared__ float s_lrt;
  if (threadIdx.x == 0) {
    double t = (double)*d_step;
    s_lrt = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = s_lrt;
  int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (int64_t i = i4; i < n && i < i4 + 4; ++i) {
    float gk = g[i] * gs;
    float mk = b1 * m[i] + (1.0f - b1) * gk;
    float vk = b2 * v[i] + (1.0f - b2) * gk * gk;
    m[i] = mk;
    v[i] = vk;
    p[i] = p[i] - lr_t * mk / (sqrtf(vk) + eps);
  }
}

void launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, int64_t* d_step, float lr, float b1,
                     float b2, float eps, float gscale, hipStream_t s) {
  hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, s, d_step);
  int64_t nt = (n + 3) / 4;
  hipLaunchKernelGGL(k_adam_dev, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, d_step, lr, b1, b2, eps,
                     gscale);
}

// Tanhize (analyzer.py:77-87)
__global__ void k_tanhize(const float* __restrict__ in, const float* __restrict__ xmin,
                          const float* __restrict__ xmax, float* __restrict__ out, int64_t N, int H, bool fwd) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int b = (int)(i % H);
  float lo = xmin[b], sc = xmax[b] - lo, v = in[i];
  if (fwd) {
    float u = (v - lo) / sc;
    u = fminf(fmaxf(u, 0.f), 1.f);
    out[i] = u * 2.f - 1.f;
  } else {
    out[i] = (v * .5f + .5f) * sc + lo;
  }
}

void launch_tanhize(const float* in, const float* xmin, const float* xmax, float* out, int64_t F, int H,
                    bool forward, hipStream_t s) {
  int64_t N = F * H;
  hipLaunchKernelGGL(k_tanhize, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, in, xmin, xmax, out, N, H, forward);
}

// analyzer.py:113-127: x = Tanhize(record[:H]) ; y = int64(record[-1]).  `idx` (optional) selects the record
// of every output row: the shuffled-batch gather of analyzer.py:128-135 happens inside this pass, so a batch
// costs one read of the H + 1 floats it uses instead of a full-record gather followed by a second pass.
__global__ void k_unpack(const float* __restrict__ rec, const int64_t* __restrict__ idx, int64_t F, int R, int H,
                         const float* __restrict__ xmin, const float* __restrict__ xmax, float* __restrict__ x,
                         int64_t* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * (H + 1)) return;
  int b = (int)(i % (H + 1));
  int64_t f = i / (H + 1);
  const int64_t r = idx ? idx[f] : f;
  if (b == H) {
    y[f] = (int64_t)rec[r * R + R - 1];  // tf.cast(float32 -> int64): truncation toward zero
  } else {
    float lo = xmin[b], sc = xmax[b] - lo;
    float u = (rec[r * R + b] - lo) / sc;
    u = fminf(fmaxf(u, 0.f), 1.f);
    x[f * H + b] = u * 2.f - 1.f;
  }
}

void launch_unpack(const float* rec, const int64_t* idx, int64_t F, int rec_floats, int H, const float* xmin,
                   const float* xmax, float* x, int64_t* y, hipStream_t s) {
  int64_t N = F * (H + 1);
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rec, idx, F, rec_floats, H, xmin, xmax, x, y);
}

// count of speaker ids outside [0, ny)
__global__ void k_check_ids(const int64_t* __restrict__ y, int64_t F, int ny, int* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F && (y[i] < 0 || y[i] >= ny)) atomicAdd(flag, 1);
}
void launch_check_ids(const int64_t* y, int64_t F, int ny, int* flag, hipStream_t s) {
  (void)hipMemsetAsync(flag, 0, sizeof(int), s);
  hipLaunchKernelGGL(k_check_ids, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, y, F, ny, flag);
}

// stand-alone N(0,1) draw, identical to what the seeded train step draws inside its sampler kernel
__global__ void k_philox_normal(float* __restrict__ out, int64_t n, PhiloxKey key) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_normal(philox_resolve(key), (uint64_t)i);
}
void launch_philox_normal(float* out, int64_t n, PhiloxKey key, hipStream_t s) {
  hipLaunchKernelGGL(k_philox_normal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, key);
}
__global__ void k_philox_uniform(float* __restrict__ out, int64_t n, PhiloxKey key) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_uniform(philox_resolve(key), (uint64_t)i);
}
void launch_philox_uniform(float* out, int64_t n, PhiloxKey key, hipStream_t s) {
  hipLaunchKernelGGL(k_philox_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, key);
}

// tf.summary.histogram payload of model/vae.py:134-135 (x, xh): min / max / sum / sum of squares and bucket
// counts over caller-supplied ascending bucket limits.  Edges sit in LDS; a workgroup keeps a private
// histogram and flushes it with one atomic per non-empty bucket.
constexpr int kMaxEdges = 2048;
__device__ __forceinline__ void atomic_min_d(double* a, double v) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(a);
  unsigned long long old = *p;
  while (v < __longlong_as_double((long long)old)) {
    unsigned long long prev = atomicCAS(p, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}
__device__ __forceinline__ void atomic_max_d(double* a, double v) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(a);
  unsigned long long old = *p;
  while (v > __longlong_as_double((long long)old)) {
    unsigned long long prev = atomicCAS(p, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}
__global__ void __launch_bounds__(256) k_summary(const float* __restrict__ d, int64_t n, const float* __restrict__ edges,
                                                 int n_edges, double* __restrict__ stats,
                                                 unsigned long long* __restrict__ counts) {
  __shared__ float se[kMaxEdges];
  __shared__ unsigned sh[kMaxEdges + 1];
  __shared__ double red[4][4];
  for (int i = threadIdx.x; i < n_edges; i += 256) se[i] = edges[i];
  for (int i = threadIdx.x; i <= n_edges; i += 256) sh[i] = 0u;
  __syncthreads();
  double mn = 1e300, mx = -1e300, sm = 0.0, sq = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = d[i];
    int lo = 0, hi = n_edges;  // first edge > v  (bucket index)
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (se[mid] > v) hi = mid; else lo = mid + 1;
    }
    atomicAdd(&sh[lo], 1u);
    mn = fmin(mn, (double)v);
    mx = fmax(mx, (double)v);
    sm += v;
    sq += (double)v * v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, o));
    mx = fmax(mx, __shfl_xor(mx, o));
    sm += __shfl_xor(sm, o);
    sq += __shfl_xor(sq, o);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wv] = mn; red[1][wv] = mx; red[2][wv] = sm; red[3][wv] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_min_d(stats + 0, fmin(fmin(red[0][0], red[0][1]), fmin(red[0][2], red[0][3])));
    atomic_max_d(stats + 1, fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3])));
    atomicAdd(stats + 2, (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]));
    atomicAdd(stats + 3, (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]));
  }
  for (int i = threadIdx.x; i <= n_edges; i += 256)
    if (sh[i]) atomicAdd(counts + i, (unsigned long long)sh[i]);
}
void launch_summary(const float* d, int64_t n, const float* edges, int n_edges, double* stats, unsigned long long* counts,
                    hipStream_t s) {
  int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_summary, dim3((unsigned)blocks), dim3(256), 0, s, d, n, edges, n_edges, stats, counts);
}

}  // namespace vaenpvc
