// BatchNorm1d over the node axis (reference models/gcn.py:23,28; graphsage.py:24,29; gat.py:23,29;
// appnp_stack.py:21,27: nn.BatchNorm1d between conv layers, statistics over ALL N nodes).
// Four streaming kernels: column statistics (sum, sum of squares accumulated in fp64, so
// E[x^2] - mean^2 does not cancel), the affine apply, the backward column reductions
// (sum gy, sum gy * xhat) and the backward apply. Column reductions write one partial record per
// workgroup and a finish kernel adds the records in workgroup order (reproducible, no atomics).
// With `partial_only` the statistics kernels return raw sums so that a node-partitioned run can
// all-reduce them before finishing (dist/nn.py).
#include "rgbx_common.h"

namespace rgbx {
namespace {

constexpr int kMaxBlocks = 1024;

struct ColMap {
  int cv;    // column vectors per row (ceil(d / VEC)), capped at 256
  int rg;    // row groups per block = 256 / cv
};

__host__ __device__ inline ColMap col_map(int d, int vec) {
  int cv = (d + vec - 1) / vec;
  if (cv > 256) cv = 256;
  ColMap m{cv, 256 / cv};
  return m;
}

// part[block, k, c] for k in {0,1}: per-block column sums of f0 and f1, where
//   STATS:  f0 = x,          f1 = x * x
//   BWD:    f0 = gy,         f1 = gy * (x - mean[c]) * rstd[c]
template <int VEC, bool BWD>
__global__ void __launch_bounds__(256)
col_reduce_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                  const float* __restrict__ mean, const float* __restrict__ rstd, int64_t N, int d,
                  double* __restrict__ part) {
  extern __shared__ double sh[];  // [rg][2][cv * VEC]
  const ColMap M = col_map(d, VEC);
  const int tid = threadIdx.x;
  const int cvi = tid % M.cv, rgi = tid / M.cv;
  const bool thread_ok = rgi < M.rg;
  const int stride_c = M.cv * VEC;
  for (int cbase = 0; cbase < d; cbase += stride_c) {
    const int c = cbase + cvi * VEC;
    const bool active = thread_ok && c < d;
    double s0[VEC], s1[VEC];
    float mu[VEC], rs[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s0[i] = s1[i] = 0.0; mu[i] = 0.f; rs[i] = 1.f; }
    if (BWD && active) {
      load_vec<VEC>(mu, mean + c);
      load_vec<VEC>(rs, rstd + c);
    }
    if (active) {
      for (int64_t r = (int64_t)blockIdx.x * M.rg + rgi; r < N; r += (int64_t)gridDim.x * M.rg) {
        float va[VEC];
        load_vec<VEC>(va, a + r * lda + c);
        if constexpr (BWD) {
          float vb[VEC];
          load_vec<VEC>(vb, b + r * ldb + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            s0[i] += (double)va[i];
            s1[i] += (double)(va[i] * ((vb[i] - mu[i]) * rs[i]));
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            s0[i] += (double)va[i];
            s1[i] += (double)va[i] * (double)va[i];
          }
        }
      }
    }
    __syncthreads();
    if (thread_ok) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sh[(rgi * 2 + 0) * stride_c + cvi * VEC + i] = s0[i];
        sh[(rgi * 2 + 1) * stride_c + cvi * VEC + i] = s1[i];
      }
    }
    __syncthreads();
    if (rgi == 0 && active) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (c + i < d) {
          double t0 = 0.0, t1 = 0.0;
          for (int g = 0; g < M.rg; ++g) {
            t0 += sh[(g * 2 + 0) * stride_c + cvi * VEC + i];
            t1 += sh[(g * 2 + 1) * stride_c + cvi * VEC + i];
          }
          part[((int64_t)blockIdx.x * 2 + 0) * d + c + i] = t0;
          part[((int64_t)blockIdx.x * 2 + 1) * d + c + i] = t1;
        }
      }
    }
  }
}

// sums[k, c] = sum over blocks of part[b, k, c]; 8 columns per block, 32 split lanes per column (lane q adds
// partials q, q + 32, ...; the 32 sub-sums are then added in lane order: a fixed order, reproducible).
__global__ void __launch_bounds__(256)
col_finish_kernel(const double* __restrict__ part, int n_blocks, int d, double* __restrict__ sums) {
  __shared__ double sh[32][8];
  const int j = threadIdx.x & 7, q = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + j;
  double s = 0.0;
  if (c < 2 * d) {
    const int which = c / d, col = c % d;
    for (int b = q; b < n_blocks; b += 32) s += part[((int64_t)b * 2 + which) * d + col];
  }
  sh[q][j] = s;
  __syncthreads();
  if (q == 0 && c < 2 * d) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][j];
    sums[c] = t;
  }
}

// y = x * scale[c] + shift[c]
template <int VEC>
__global__ void __launch_bounds__(256)
affine_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
              const float* __restrict__ shift, float* __restrict__ y, int64_t ldy, int64_t N, int d) {
  const int per_row = (d + VEC - 1) / VEC;
  const int64_t total = N * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / per_row;
    const int c = (int)(idx % per_row) * VEC;
    float v[VEC], sc[VEC], sf[VEC];
    load_vec<VEC>(v, x + r * ldx + c);
    load_vec<VEC>(sc, scale + c);
    load_vec<VEC>(sf, shift + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = fmaf(v[i], sc[i], sf[i]);
    store_vec<VEC>(y + r * ldy + c, v);
  }
}

// gx = (gy - ca[c] - xhat * cb[c]) * ck[c],  xhat = (x - mean[c]) * rstd[c]
template <int VEC>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ gy, int64_t ldg, const float* __restrict__ x, int64_t ldx,
                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ ca,
                    const float* __restrict__ cb, const float* __restrict__ ck, float* __restrict__ gx,
                    int64_t ldgx, int64_t N, int d) {
  const int per_row = (d + VEC - 1) / VEC;
  const int64_t total = N * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / per_row;
    const int c = (int)(idx % per_row) * VEC;
    float g[VEC], v[VEC], mu[VEC], rs[VEC], a[VEC], b[VEC], k[VEC];
    load_vec<VEC>(g, gy + r * ldg + c);
    load_vec<VEC>(v, x + r * ldx + c);
    load_vec<VEC>(mu, mean + c);
    load_vec<VEC>(rs, rstd + c);
    load_vec<VEC>(a, ca + c);
    load_vec<VEC>(b, cb + c);
    load_vec<VEC>(k, ck + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = (g[i] - a[i] - (v[i] - mu[i]) * rs[i] * b[i]) * k[i];
    store_vec<VEC>(gx + r * ldgx + c, g);
  }
}

int reduce_grid(int64_t N, int d, int vec) {
  const ColMap M = col_map(d, vec);
  int64_t b = cdiv(N, (int64_t)M.rg * 16);  // >= 16 rows per thread before a partial record is written
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)(b < 1 ? 1 : b);
}

bool vec4_ok(int64_t d, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
  bool ok = d % 4 == 0;
  for (const void* p : ptrs) ok = ok && aligned16(p);
  for (int64_t ld : lds) ok = ok && ld % 4 == 0;
  return ok;
}

template <bool BWD>
int run_reduce(const float* a, int64_t lda, const float* b, int64_t ldb, const float* mean, const float* rstd,
               int64_t N, int64_t d, double* sums, double* scratch, int64_t scratch_doubles, hipStream_t s,
               const char* name) {
  if (N < 0 || d <= 0 || !sums) return fail(RGBX_E_ARG, "%s: bad argument", name);
  if (d >= INT32_MAX) return fail(RGBX_E_RANGE, "%s: d exceeds int32", name);
  if (N == 0) {
    RGBX_HIP(hipMemsetAsync(sums, 0, 2 * d * sizeof(double), s));
    return RGBX_OK;
  }
  if (!a || lda < d || (BWD && (!b || ldb < d || !mean || !rstd)) || !scratch)
    return fail(RGBX_E_ARG, "%s: null pointer or leading dimension < d", name);
  const bool v4 = BWD ? vec4_ok(d, {a, b, mean, rstd}, {lda, ldb}) : vec4_ok(d, {a}, {lda});
  const int vec = v4 ? 4 : 1;
  const int grid = reduce_grid(N, (int)d, vec);
  if (scratch_doubles < (int64_t)grid * 2 * d)
    return fail(RGBX_E_WS, "%s: scratch %lld < %lld doubles", name, (long long)scratch_doubles,
                (long long)grid * 2 * d);
  const ColMap M = col_map((int)d, vec);
  const size_t lds = (size_t)M.rg * 2 * M.cv * vec * sizeof(double);
  if (v4)
    col_reduce_kernel<4, BWD><<<grid, 256, lds, s>>>(a, lda, b, ldb, mean, rstd, N, (int)d, scratch);
  else
    col_reduce_kernel<1, BWD><<<grid, 256, lds, s>>>(a, lda, b, ldb, mean, rstd, N, (int)d, scratch);
  RGBX_CHECK_LAUNCH(name);
  col_finish_kernel<<<(int)cdiv(2 * d, 8), 256, 0, s>>>(scratch, grid, (int)d, sums);
  RGBX_CHECK_LAUNCH("col_finish_kernel");
  return RGBX_OK;
}

// Everything between the (possibly all-reduced) raw sums and the apply pass, one thread per column:
// packed = [sum x (d), sum x^2 (d), row count (1)] in fp64 -> mean, biased variance, rstd, the affine map of the
// training forward, and the running-statistics update (unbiased variance, momentum) of nn.BatchNorm1d.
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const double* __restrict__ packed, const float* __restrict__ weight,
                   const float* __restrict__ bias, float eps, float momentum, float* __restrict__ running_mean,
                   float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ rstd,
                   float* __restrict__ scale, float* __restrict__ shift, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const double n = packed[2 * d];
  const double m64 = packed[c] / n;
  double v64 = packed[d + c] / n - m64 * m64;  // biased variance; fp64, so the difference does not cancel
  v64 = v64 > 0.0 ? v64 : 0.0;
  const float m = (float)m64, var = (float)v64;
  const float r = 1.0f / sqrtf(var + eps);
  const float sc = weight[c] * r;
  mean[c] = m;
  rstd[c] = r;
  scale[c] = sc;
  shift[c] = bias[c] - m * sc;
  if (running_mean) {
    const double nm1 = n > 1.0 ? n - 1.0 : 1.0;
    const float unbiased = var * (float)(n / nm1);
    running_mean[c] = running_mean[c] * (1.0f - momentum) + momentum * m;
    running_var[c] = running_var[c] * (1.0f - momentum) + momentum * unbiased;
  }
}

// Backward: the per-column coefficients of the apply pass from the (all-reduced) sums, and the parameter
// gradients from this rank's own sums.
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const double* __restrict__ glob, const double* __restrict__ local,
                       const double* __restrict__ count, const float* __restrict__ weight,
                       const float* __restrict__ rstd, float* __restrict__ ca, float* __restrict__ cb,
                       float* __restrict__ ck, float* __restrict__ g_weight, float* __restrict__ g_bias, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const double n = *count;
  ca[c] = (float)(glob[c] / n);
  cb[c] = (float)(glob[d + c] / n);
  ck[c] = weight[c] * rstd[c];
  g_bias[c] = (float)local[c];
  g_weight[c] = (float)local[d + c];
}

// Eval-mode BatchNorm folded into the linear layer in front of it, and the operand layout of the fused kernels made
// in the same pass: wt[k, n] = W[n, k] * scale[n] (W [Nout, K] row-major -> W'^T [K, Nout]), 32 x 32 tiles through LDS
// so that both sides move contiguous rows; the workgroups of the first K-tile also write b'[n].
__global__ void __launch_bounds__(256)
fold_bn_linear_kernel(const float* __restrict__ W, int64_t ldw, const float* __restrict__ Wr, int64_t ldwr,
                      const float* __restrict__ bias, const float* __restrict__ bias2,
                      const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                      float* __restrict__ wt, float* __restrict__ wrt, float* __restrict__ b_out, int Nout, int K) {
  __shared__ float tile[2][32][33];
  __shared__ float sc[32];
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (threadIdx.x < 32) {
    const int n = n0 + threadIdx.x;
    float s = 1.f, t = 0.f;
    if (n < Nout && gamma) {
      s = gamma[n] * (1.0f / sqrtf(rvar[n] + eps));
      t = beta[n] - rmean[n] * s;
    }
    sc[threadIdx.x] = s;
    if (n < Nout && b_out && blockIdx.y == 0) {
      float b = bias ? bias[n] : 0.f;
      if (bias2) b += bias2[n];
      b_out[n] = b * s + t;
    }
  }
  for (int r = ty; r < 32; r += 8) {  // rows n of W, columns k
    const int n = n0 + r, k = k0 + tx;
    const bool ok = n < Nout && k < K;
    tile[0][r][tx] = ok ? W[(int64_t)n * ldw + k] : 0.f;
    if (Wr) tile[1][r][tx] = ok ? Wr[(int64_t)n * ldwr + k] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {  // rows k of the output, columns n
    const int k = k0 + r, n = n0 + tx;
    if (k < K && n < Nout) {
      wt[(int64_t)k * Nout + n] = tile[0][tx][r] * sc[tx];
      if (Wr) wrt[(int64_t)k * Nout + n] = tile[1][tx][r] * sc[tx];
    }
  }
}

int elt_grid(int64_t total) {
  int64_t b = cdiv(total, 256);
  return (int)(b < kMaxGrid ? (b < 1 ? 1 : b) : kMaxGrid);
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_fold_bn_linear_f32(const float* W, int64_t ldw, const float* Wr, int64_t ldwr, const float* bias,
                                       const float* bias2, const float* gamma, const float* beta,
                                       const float* running_mean, const float* running_var, float eps, float* wt,
                                       float* wrt, float* b_out, int64_t Nout, int64_t K, rgbx_stream_t stream) {
  if (Nout <= 0 || K <= 0) return fail(RGBX_E_ARG, "fold_bn_linear: bad size");
  if (!W || !wt || (Wr && !wrt)) return fail(RGBX_E_ARG, "fold_bn_linear: null pointer");
  const bool bn = gamma || beta || running_mean || running_var;
  if (bn && !(gamma && beta && running_mean && running_var))
    return fail(RGBX_E_ARG, "fold_bn_linear: gamma, beta, running_mean and running_var go together");
  if (ldw < K || (Wr && ldwr < K)) return fail(RGBX_E_ARG, "fold_bn_linear: leading dimension too small");
  if (Nout >= INT32_MAX || K >= INT32_MAX) return fail(RGBX_E_RANGE, "fold_bn_linear: size exceeds int32");
  dim3 grid((unsigned)cdiv(Nout, 32), (unsigned)cdiv(K, 32));
  fold_bn_linear_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(W, ldw, Wr, ldwr, bias, bias2, gamma, beta, running_mean,
                                                            running_var, eps, wt, wrt, b_out, (int)Nout, (int)K);
  RGBX_CHECK_LAUNCH("fold_bn_linear_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_bn_scratch_doubles(int64_t N, int64_t d, int64_t* count) {
  if (!count || N < 0 || d <= 0) return fail(RGBX_E_ARG, "bn_scratch_doubles: bad argument");
  *count = (int64_t)kMaxBlocks * 2 * d;
  return RGBX_OK;
}

extern "C" int rgbx_bn_stats_f32(const float* x, int64_t ldx, int64_t N, int64_t d, double* sums,
                                 double* scratch, int64_t scratch_doubles, rgbx_stream_t stream) {
  return run_reduce<false>(x, ldx, nullptr, 0, nullptr, nullptr, N, d, sums, scratch, scratch_doubles,
                           (hipStream_t)stream, "bn_stats");
}

extern "C" int rgbx_bn_bwd_reduce_f32(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, int64_t N, int64_t d, double* sums, double* scratch,
                                      int64_t scratch_doubles, rgbx_stream_t stream) {
  return run_reduce<true>(gy, ldg, x, ldx, mean, rstd, N, d, sums, scratch, scratch_doubles, (hipStream_t)stream,
                          "bn_bwd_reduce");
}

extern "C" int rgbx_bn_finalize_f32(const double* packed, const float* weight, const float* bias, float eps,
                                    float momentum, float* running_mean, float* running_var, float* mean,
                                    float* rstd, float* scale, float* shift, int64_t d, rgbx_stream_t stream) {
  if (d <= 0 || d >= INT32_MAX) return fail(RGBX_E_ARG, "bn_finalize: bad width");
  if (!packed || !weight || !bias || !mean || !rstd || !scale || !shift || (running_mean != nullptr) != (running_var != nullptr))
    return fail(RGBX_E_ARG, "bn_finalize: null pointer");
  bn_finalize_kernel<<<(int)cdiv(d, 256), 256, 0, (hipStream_t)stream>>>(packed, weight, bias, eps, momentum, running_mean,
                                                                    running_var, mean, rstd, scale, shift, (int)d);
  RGBX_CHECK_LAUNCH("bn_finalize_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_bn_bwd_finalize_f32(const double* glob, const double* local, const double* count,
                                        const float* weight, const float* rstd, float* ca, float* cb, float* ck,
                                        float* g_weight, float* g_bias, int64_t d, rgbx_stream_t stream) {
  if (d <= 0 || d >= INT32_MAX) return fail(RGBX_E_ARG, "bn_bwd_finalize: bad width");
  if (!glob || !local || !count || !weight || !rstd || !ca || !cb || !ck || !g_weight || !g_bias)
    return fail(RGBX_E_ARG, "bn_bwd_finalize: null pointer");
  bn_bwd_finalize_kernel<<<(int)cdiv(d, 256), 256, 0, (hipStream_t)stream>>>(glob, local, count, weight, rstd, ca, cb, ck,
                                                                        g_weight, g_bias, (int)d);
  RGBX_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_affine_cols_f32(const float* x, int64_t ldx, const float* scale, const float* shift, float* y,
                                    int64_t ldy, int64_t N, int64_t d, rgbx_stream_t stream) {
  if (N < 0 || d < 0) return fail(RGBX_E_ARG, "affine_cols: negative size");
  if (N == 0 || d == 0) return RGBX_OK;
  if (!x || !scale || !shift || !y || ldx < d || ldy < d) return fail(RGBX_E_ARG, "affine_cols: null pointer or ld < d");
  if (d >= INT32_MAX) return fail(RGBX_E_RANGE, "affine_cols: d exceeds int32");
  hipStream_t s = (hipStream_t)stream;
  if (vec4_ok(d, {x, scale, shift, y}, {ldx, ldy}))
    affine_kernel<4><<<elt_grid(N * (d / 4)), 256, 0, s>>>(x, ldx, scale, shift, y, ldy, N, (int)d);
  else
    affine_kernel<1><<<elt_grid(N * d), 256, 0, s>>>(x, ldx, scale, shift, y, ldy, N, (int)d);
  RGBX_CHECK_LAUNCH("affine_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_bn_bwd_apply_f32(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* mean,
                                     const float* rstd, const float* ca, const float* cb, const float* ck, float* gx,
                                     int64_t ldgx, int64_t N, int64_t d, rgbx_stream_t stream) {
  if (N < 0 || d < 0) return fail(RGBX_E_ARG, "bn_bwd_apply: negative size");
  if (N == 0 || d == 0) return RGBX_OK;
  if (!gy || !x || !mean || !rstd || !ca || !cb || !ck || !gx || ldg < d || ldx < d || ldgx < d)
    return fail(RGBX_E_ARG, "bn_bwd_apply: null pointer or ld < d");
  if (d >= INT32_MAX) return fail(RGBX_E_RANGE, "bn_bwd_apply: d exceeds int32");
  hipStream_t s = (hipStream_t)stream;
  if (vec4_ok(d, {gy, x, mean, rstd, ca, cb, ck, gx}, {ldg, ldx, ldgx}))
    bn_bwd_apply_kernel<4><<<elt_grid(N * (d / 4)), 256, 0, s>>>(gy, ldg, x, ldx, mean, rstd, ca, cb, ck, gx, ldgx, N,
                                                              (int)d);
  else
    bn_bwd_apply_kernel<1><<<elt_grid(N * d), 256, 0, s>>>(gy, ldg, x, ldx, mean, rstd, ca, cb, ck, gx, ldgx, N, (int)d);
  RGBX_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return RGBX_OK;
}
