// Also the same loss taken straight from the logits (cross-entropy = NLL of log_softmax): the model's forward
// ends in F.log_softmax (models/gcn.py:31) and the loop applies NLLLoss to it; when the loop only wants the loss
// value, its gradient and the accuracy, log-softmax never has to be written out: rgbx_masked_ce_fwd_f32 reads the
// selected rows once (logsumexp, label logit, arg-max), rgbx_masked_ce_bwd_f32 writes
// scale * (softmax - onehot) in one pass.
//
// Masked NLL over log-probabilities + accuracy, forward and backward: the reference's
// `criterion(out[mask], y[mask])` with nn.NLLLoss() and `out[mask].max(dim=1)[1].eq(y[mask])`
// (itexperiments.py:400,429,434,467,472,624-626,643) without materialising out[mask]: one pass over
// the mask, one gathered element (or one row, for the arg-max) per selected node.
#include "rgbx_common.h"

namespace rgbx {
namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// per block: sum of -logp[i, y_i] and number of selected rows (mask set, label in range)
__global__ void __launch_bounds__(256)
nll_sum_kernel(const float* __restrict__ logp, int64_t ld, const int64_t* __restrict__ y,
               const uint8_t* __restrict__ mask, int64_t N, int C, double* __restrict__ partials) {
  __shared__ double sh[4];
  double loss = 0.0, cnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (mask && !mask[i]) continue;
    const int64_t t = y[i];
    if (t < 0 || t >= C) continue;
    loss -= (double)logp[i * ld + t];
    cnt += 1.0;
  }
  loss = block_sum(loss, sh);
  cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) {  // one record per block; summed in block order by nll_finish_kernel
    partials[3 * blockIdx.x + 0] = loss;
    partials[3 * blockIdx.x + 1] = cnt;
    partials[3 * blockIdx.x + 2] = 0.0;
  }
}

// A wave takes 64 consecutive rows per step: mask and label are read coalesced (one lane per row), the
// selected rows come out of a ballot, and the wave then sweeps them two at a time (both row loads issued
// before either reduction), so no row waits on a dependent mask -> label -> row chain.
__device__ __forceinline__ void row_argmax(const float* __restrict__ row, int C, int lane, float& best, int& arg) {
  best = -INFINITY;
  arg = INT32_MAX;
  for (int c = lane; c < C; c += 64) {
    const float v = row[c];
    if (v > best) { best = v; arg = c; }
  }
}

__device__ __forceinline__ void wave_argmax(float& best, int& arg) {
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oa = __shfl_xor(arg, off);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
}

__global__ void __launch_bounds__(256)
nll_acc_kernel(const float* __restrict__ logp, int64_t ld, const int64_t* __restrict__ y,
               const uint8_t* __restrict__ mask, int64_t N, int C, double* __restrict__ partials) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  double loss = 0.0, cnt = 0.0, hit = 0.0;
  for (int64_t base = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * 64; base < N;
       base += (int64_t)gridDim.x * wpb * 64) {
    const int64_t i = base + lane;
    int t = -1;
    if (i < N && (!mask || mask[i])) {
      const int64_t ti = y[i];
      if (ti >= 0 && ti < C) t = (int)ti;
    }
    unsigned long long sel = __ballot(t >= 0);
    while (sel) {
      const int r0 = __builtin_ctzll(sel);
      sel &= sel - 1;
      const int r1 = sel ? __builtin_ctzll(sel) : -1;
      if (r1 >= 0) sel &= sel - 1;
      const float* row0 = logp + (base + r0) * ld;
      const float* row1 = logp + (base + (r1 >= 0 ? r1 : r0)) * ld;
      float b0, b1;
      int a0, a1;
      row_argmax(row0, C, lane, b0, a0);
      row_argmax(row1, C, lane, b1, a1);
      wave_argmax(b0, a0);
      wave_argmax(b1, a1);
      const int t0 = __shfl(t, r0);
      const int t1 = __shfl(t, r1 >= 0 ? r1 : r0);
      if (lane == 0) {
        loss -= (double)row0[t0];
        cnt += 1.0;
        hit += a0 == t0 ? 1.0 : 0.0;
        if (r1 >= 0) {
          loss -= (double)row1[t1];
          cnt += 1.0;
          hit += a1 == t1 ? 1.0 : 0.0;
        }
      }
    }
  }
  loss = block_sum(loss, sh);
  cnt = block_sum(cnt, sh);
  hit = block_sum(hit, sh);
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x + 0] = loss;
    partials[3 * blockIdx.x + 1] = cnt;
    partials[3 * blockIdx.x + 2] = hit;
  }
}

// logsumexp of one row by a wave (lane-strided), plus the arg-max (lowest index on ties); every lane gets both
__device__ __forceinline__ void row_lse_argmax(const float* __restrict__ row, int C, int lane, float& lse, int& arg) {
  float best;
  row_argmax(row, C, lane, best, arg);
  wave_argmax(best, arg);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(row[c] - best);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  lse = best + logf(s);
}

// Cross-entropy from logits over the selected rows: loss = sum(lse_i - z[i, y_i]), count, arg-max hits.
__global__ void __launch_bounds__(256)
ce_fwd_kernel(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ y,
              const uint8_t* __restrict__ mask, int64_t N, int C, double* __restrict__ partials) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  double loss = 0.0, cnt = 0.0, hit = 0.0;
  for (int64_t base = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * 64; base < N;
       base += (int64_t)gridDim.x * wpb * 64) {
    const int64_t i = base + lane;
    int t = -1;
    if (i < N && (!mask || mask[i])) {
      const int64_t ti = y[i];
      if (ti >= 0 && ti < C) t = (int)ti;
    }
    unsigned long long sel = __ballot(t >= 0);
    while (sel) {
      const int r0 = __builtin_ctzll(sel);
      sel &= sel - 1;
      const float* row = z + (base + r0) * ld;
      float lse;
      int arg;
      row_lse_argmax(row, C, lane, lse, arg);
      const int t0 = __shfl(t, r0);
      if (lane == 0) {
        loss += (double)(lse - row[t0]);
        cnt += 1.0;
        hit += arg == t0 ? 1.0 : 0.0;
      }
    }
  }
  loss = block_sum(loss, sh);
  cnt = block_sum(cnt, sh);
  hit = block_sum(hit, sh);
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x + 0] = loss;
    partials[3 * blockIdx.x + 1] = cnt;
    partials[3 * blockIdx.x + 2] = hit;
  }
}

// grad[i, c] = scale * (softmax(z_i)[c] - [c == y_i]) for selected rows, 0 for the others; one wave per row.
__global__ void __launch_bounds__(256)
ce_bwd_kernel(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ y,
              const uint8_t* __restrict__ mask, int64_t N, int C, const float* __restrict__ scale,
              float* __restrict__ grad, int64_t ldg) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const float s = scale[0];
  for (int64_t i = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); i < N; i += (int64_t)gridDim.x * wpb) {
    int t = -1;
    if (!mask || mask[i]) {
      const int64_t ti = y[i];
      if (ti >= 0 && ti < C) t = (int)ti;
    }
    float* g = grad + i * ldg;
    if (t < 0) {
      for (int c = lane; c < C; c += 64) g[c] = 0.f;
      continue;
    }
    const float* row = z + i * ld;
    float lse;
    int arg;
    row_lse_argmax(row, C, lane, lse, arg);
    for (int c = lane; c < C; c += 64) g[c] = s * (expf(row[c] - lse) - (c == t ? 1.f : 0.f));
  }
}

// ---- C % 4 == 0, C <= 256: a row is ONE float4 per lane of a group of LPR = pow2ceil(C / 4) lanes, so a wave holds
// 64 / LPR rows in registers at once: one global read per row, group-wide shuffles for max / sum / arg-max.
__device__ __forceinline__ void group_lse_argmax(const float (&v)[4], int c, int C, int lpr, float& lse, int& arg) {
  float best = -INFINITY;
  arg = INT32_MAX;
  if (c < C) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v[k] > best) { best = v[k]; arg = c + k; }
  }
  for (int off = lpr >> 1; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oa = __shfl_xor(arg, off);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  float s = 0.f;
  if (c < C) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s += expf(v[k] - best);
  }
  for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
  lse = best + logf(s);
}

// BLK: the logits are BLOCKED (element (i, c) at z + (c / bc) * bs + i * bc + c % bc: column slices as a node-partitioned
// run's exchange delivers them, rgbx_fused_layer_t) and `bias` [C] is added on the fly — the eval forwards of the fused
// per-rank schedule whose last transform ran before the exchange read their loss statistics straight from the received
// slices: only the SELECTED rows are touched, no pass turns all rows into row-major logits first.
template <bool BLK>
__global__ void __launch_bounds__(256)
ce_fwd_vec_kernel(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ y,
                  const uint8_t* __restrict__ mask, int64_t N, int C, int lpr, double* __restrict__ partials,
                  int64_t bc, int64_t bs, const float* __restrict__ bias) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int nr = 64 / lpr;               // rows in flight per wave
  const int grp = lane / lpr, c = (lane % lpr) * 4;
  double loss = 0.0, cnt = 0.0, hit = 0.0;
  for (int64_t base = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * 64; base < N;
       base += (int64_t)gridDim.x * wpb * 64) {
    const int64_t i = base + lane;
    int t = -1;
    if (i < N && (!mask || mask[i])) {
      const int64_t ti = y[i];
      if (ti >= 0 && ti < C) t = (int)ti;
    }
    unsigned long long sel = __ballot(t >= 0);
    while (sel) {
      // the next `nr` selected rows of this 64-row window, one per lane group
      int myrow = -1;
      for (int k = 0; k < nr && sel; ++k) {
        const int r = __builtin_ctzll(sel);
        sel &= sel - 1;
        if (k == grp) myrow = r;
      }
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int64_t ri = base + (myrow >= 0 ? myrow : 0);
      const float* row = z + ri * ld;
      if (myrow >= 0 && c < C) {
        if constexpr (BLK) {
          load_vec<4>(v, z + (int64_t)(c / (int)bc) * bs + ri * bc + (c % (int)bc));
          if (bias) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += bias[c + k];
          }
        } else {
          load_vec<4>(v, row + c);
        }
      }
      float lse;
      int arg;
      group_lse_argmax(v, c, C, lpr, lse, arg);
      const int tr = __shfl(t, myrow >= 0 ? myrow : 0);
      if (myrow >= 0 && c == 0) {
        float zt;
        if constexpr (BLK) zt = z[(int64_t)(tr / (int)bc) * bs + ri * bc + (tr % (int)bc)] + (bias ? bias[tr] : 0.f);
        else zt = row[tr];
        loss += (double)(lse - zt);
        cnt += 1.0;
        hit += arg == tr ? 1.0 : 0.0;
      }
    }
  }
  loss = block_sum(loss, sh);
  cnt = block_sum(cnt, sh);
  hit = block_sum(hit, sh);
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x + 0] = loss;
    partials[3 * blockIdx.x + 1] = cnt;
    partials[3 * blockIdx.x + 2] = hit;
  }
}

__global__ void __launch_bounds__(256)
ce_bwd_vec_kernel(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ y,
                  const uint8_t* __restrict__ mask, int64_t N, int C, int lpr, const float* __restrict__ scale,
                  float* __restrict__ grad, int64_t ldg) {
  constexpr int R = 4;  // rows per lane group and iteration: all R row loads are issued before the first reduction
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int nr = 64 / lpr;
  const int grp = lane / lpr, c = (lane % lpr) * 4;
  const float s = scale[0];
  for (int64_t i0 = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * nr * R; i0 < N;
       i0 += (int64_t)gridDim.x * wpb * nr * R) {
    int t[R];
    float v[R][4];
    // mask and label of all R rows first (independent loads), then the R row loads: no mask -> label -> row chain
    uint8_t mk[R];
    int64_t lab[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * nr + grp;
      mk[r] = (i < N) ? (mask ? mask[i] : (uint8_t)1) : (uint8_t)0;
      lab[r] = (i < N) ? y[i] : -1;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * nr + grp;
      t[r] = (mk[r] && lab[r] >= 0 && lab[r] < C) ? (int)lab[r] : -1;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[r][k] = 0.f;
      if (t[r] >= 0 && c < C) load_vec<4>(v[r], z + i * ld + c);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * nr + grp;
      float lse;
      int arg;
      group_lse_argmax(v[r], c, C, lpr, lse, arg);  // every lane takes part in the shuffles
      if (i < N && c < C) {
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = t[r] >= 0 ? s * (expf(v[r][k] - lse) - (c + k == t[r] ? 1.f : 0.f)) : 0.f;
        store_vec<4>(grad + i * ldg + c, g);
      }
    }
  }
}

// stats[k] = sum over blocks of partials[b, k], in block order (one block; reproducible, no atomics)
__global__ void __launch_bounds__(256)
nll_finish_kernel(const double* __restrict__ partials, int n_blocks, double* __restrict__ stats) {
  __shared__ double sh[4];
  for (int k = 0; k < 3; ++k) {
    double v = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) v += partials[3 * b + k];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) stats[k] = v;
  }
}

// grad[i, :] = 0 except grad[i, y_i] = -scale for selected rows; scale read from device memory
template <int VEC>
__global__ void __launch_bounds__(256)
nll_bwd_kernel(const int64_t* __restrict__ y, const uint8_t* __restrict__ mask, int64_t N, int C,
               const float* __restrict__ scale, float* __restrict__ grad, int64_t ldg) {
  const float s = -scale[0];
  const int per_row = (C + VEC - 1) / VEC;
  const int64_t total = N * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / per_row;
    const int c = (int)(idx % per_row) * VEC;
    int64_t t = -1;
    if (!mask || mask[i]) t = y[i];
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = (c + k == t) ? s : 0.f;
    if constexpr (VEC == 1) {
      grad[i * ldg + c] = v[0];
    } else {
      store_vec<VEC>(grad + i * ldg + c, v);
    }
  }
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_masked_nll_scratch_doubles(int64_t N, int want_accuracy, int64_t* count) {
  if (!count || N < 0) return fail(RGBX_E_ARG, "masked_nll_scratch_doubles: bad argument");
  int64_t b = want_accuracy ? cdiv(N, 4 * 64) : cdiv(N, 256);
  const int64_t cap = want_accuracy ? kMaxGrid : 2048;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  *count = 3 * b;
  return RGBX_OK;
}

extern "C" int rgbx_masked_nll_fwd_f32(const float* logp, int64_t ld, const int64_t* y, const uint8_t* mask,
                                       int64_t N, int64_t C, double* stats, double* scratch,
                                       int64_t scratch_doubles, int want_accuracy, rgbx_stream_t stream) {
  if (N < 0 || C <= 0 || !stats) return fail(RGBX_E_ARG, "masked_nll_fwd: bad argument");
  if (C >= INT32_MAX) return fail(RGBX_E_RANGE, "masked_nll_fwd: C exceeds int32");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    RGBX_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(double), s));
    return RGBX_OK;
  }
  if (!logp || !y || ld < C) return fail(RGBX_E_ARG, "masked_nll_fwd: null pointer or ld < C");
  int64_t need = 0;
  if (int rc = rgbx_masked_nll_scratch_doubles(N, want_accuracy, &need)) return rc;
  if (!scratch || scratch_doubles < need)
    return fail(RGBX_E_WS, "masked_nll_fwd: scratch %lld < %lld doubles", (long long)scratch_doubles, (long long)need);
  const int grid = (int)(need / 3);
  if (want_accuracy)
    nll_acc_kernel<<<grid, 256, 0, s>>>(logp, ld, y, mask, N, (int)C, scratch);
  else
    nll_sum_kernel<<<grid, 256, 0, s>>>(logp, ld, y, mask, N, (int)C, scratch);
  RGBX_CHECK_LAUNCH("masked_nll_fwd");
  nll_finish_kernel<<<1, 256, 0, s>>>(scratch, grid, stats);
  RGBX_CHECK_LAUNCH("nll_finish_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_masked_ce_fwd_f32(const float* logits, int64_t ld, const int64_t* y, const uint8_t* mask,
                                      int64_t N, int64_t C, double* stats, double* scratch, int64_t scratch_doubles,
                                      rgbx_stream_t stream) {
  if (N < 0 || C <= 0 || !stats) return fail(RGBX_E_ARG, "masked_ce_fwd: bad argument");
  if (C >= INT32_MAX) return fail(RGBX_E_RANGE, "masked_ce_fwd: C exceeds int32");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    RGBX_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(double), s));
    return RGBX_OK;
  }
  if (!logits || !y || ld < C) return fail(RGBX_E_ARG, "masked_ce_fwd: null pointer or ld < C");
  int64_t need = 0;
  if (int rc = rgbx_masked_nll_scratch_doubles(N, 1, &need)) return rc;
  if (!scratch || scratch_doubles < need)
    return fail(RGBX_E_WS, "masked_ce_fwd: scratch %lld < %lld doubles", (long long)scratch_doubles, (long long)need);
  const int grid = (int)(need / 3);
  if (C % 4 == 0 && C <= 256 && ld % 4 == 0 && aligned16(logits)) {
    int lpr = 1;
    while (lpr * 4 < C) lpr *= 2;
    ce_fwd_vec_kernel<false><<<grid, 256, 0, s>>>(logits, ld, y, mask, N, (int)C, lpr, scratch, 0, 0, nullptr);
  } else {
    ce_fwd_kernel<<<grid, 256, 0, s>>>(logits, ld, y, mask, N, (int)C, scratch);
  }
  RGBX_CHECK_LAUNCH("ce_fwd_kernel");
  nll_finish_kernel<<<1, 256, 0, s>>>(scratch, grid, stats);
  RGBX_CHECK_LAUNCH("nll_finish_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_masked_ce_fwd_blocked_f32(const float* logits, int64_t blk_cols, int64_t blk_stride,
                                              const float* bias, const int64_t* y, const uint8_t* mask, int64_t N,
                                              int64_t C, double* stats, double* scratch, int64_t scratch_doubles,
                                              rgbx_stream_t stream) {
  if (N < 0 || C <= 0 || !stats) return fail(RGBX_E_ARG, "masked_ce_fwd_blocked: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    RGBX_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(double), s));
    return RGBX_OK;
  }
  if (!logits || !y) return fail(RGBX_E_ARG, "masked_ce_fwd_blocked: null pointer");
  if (C % 4 || C > 256 || blk_cols <= 0 || blk_cols % 4 || C % blk_cols || blk_stride % 4 || !aligned16(logits))
    return fail(RGBX_E_SHAPE, "masked_ce_fwd_blocked: needs C %% 4 == 0, C <= 256, block width %% 4 == 0 dividing C, "
                              "16-byte aligned blocks (got C=%lld, blk_cols=%lld)", (long long)C, (long long)blk_cols);
  int64_t need = 0;
  if (int rc = rgbx_masked_nll_scratch_doubles(N, 1, &need)) return rc;
  if (!scratch || scratch_doubles < need)
    return fail(RGBX_E_WS, "masked_ce_fwd_blocked: scratch %lld < %lld doubles", (long long)scratch_doubles, (long long)need);
  const int grid = (int)(need / 3);
  int lpr = 1;
  while (lpr * 4 < C) lpr *= 2;
  ce_fwd_vec_kernel<true><<<grid, 256, 0, s>>>(logits, 0, y, mask, N, (int)C, lpr, scratch, blk_cols, blk_stride, bias);
  RGBX_CHECK_LAUNCH("ce_fwd_vec_kernel (blocked)");
  nll_finish_kernel<<<1, 256, 0, s>>>(scratch, grid, stats);
  RGBX_CHECK_LAUNCH("nll_finish_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_masked_ce_bwd_f32(const float* logits, int64_t ld, const int64_t* y, const uint8_t* mask,
                                      int64_t N, int64_t C, const float* scale, float* grad, int64_t ldg,
                                      rgbx_stream_t stream) {
  if (N < 0 || C <= 0) return fail(RGBX_E_ARG, "masked_ce_bwd: bad size");
  if (N == 0) return RGBX_OK;
  if (!logits || !y || !scale || !grad || ld < C || ldg < C)
    return fail(RGBX_E_ARG, "masked_ce_bwd: null pointer or ld < C");
  if (C >= INT32_MAX) return fail(RGBX_E_RANGE, "masked_ce_bwd: C exceeds int32");
  if (C % 4 == 0 && C <= 256 && ld % 4 == 0 && ldg % 4 == 0 && aligned16(logits) && aligned16(grad)) {
    int lpr = 1;
    while (lpr * 4 < C) lpr *= 2;
    int64_t b = cdiv(N, 4 * (64 / lpr) * 4);  // 4 waves x (64 / lpr) lane groups x R = 4 rows per iteration
    const int grid = (int)(b < 4 * kMaxGrid ? b : 4 * kMaxGrid);
    ce_bwd_vec_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(logits, ld, y, mask, N, (int)C, lpr, scale, grad, ldg);
  } else {
    int64_t b = cdiv(N, 4);
    const int grid = (int)(b < 4 * kMaxGrid ? b : 4 * kMaxGrid);
    ce_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(logits, ld, y, mask, N, (int)C, scale, grad, ldg);
  }
  RGBX_CHECK_LAUNCH("ce_bwd_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_masked_nll_bwd_f32(const int64_t* y, const uint8_t* mask, int64_t N, int64_t C,
                                       const float* scale, float* grad, int64_t ldg, rgbx_stream_t stream) {
  if (N < 0 || C <= 0) return fail(RGBX_E_ARG, "masked_nll_bwd: bad size");
  if (N == 0) return RGBX_OK;
  if (!y || !scale || !grad || ldg < C) return fail(RGBX_E_ARG, "masked_nll_bwd: null pointer or ld < C");
  if (C >= INT32_MAX) return fail(RGBX_E_RANGE, "masked_nll_bwd: C exceeds int32");
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = C % 4 == 0 && ldg % 4 == 0 && aligned16(grad);
  const int64_t total = N * (v4 ? C / 4 : C);
  int64_t b = cdiv(total, 256);
  const int grid = (int)(b < kMaxGrid ? b : kMaxGrid);
  if (v4)
    nll_bwd_kernel<4><<<grid, 256, 0, s>>>(y, mask, N, (int)C, scale, grad, ldg);
  else
    nll_bwd_kernel<1><<<grid, 256, 0, s>>>(y, mask, N, (int)C, scale, grad, ldg);
  RGBX_CHECK_LAUNCH("nll_bwd_kernel");
  return RGBX_OK;
}
