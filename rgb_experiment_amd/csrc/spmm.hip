// CSR row-gather SpMM for gfx950: out[i,:] = a*rs[i]*sum_p w[p]*x[col[p],:] + b*y[i,:] + bias[:].
// Replaces MessagePassing.propagate(aggr='add'|'mean', message = norm * x_j): reference
// models/dagnn.py:34-36,46,57-59 (in-repo twin of what GCNConv / APPNP run, models/gcn.py:27,
// models/appnp_stack.py:29) and models/graphsage.py:58 (mean). No [E', d] temporary exists:
// one wave owns one destination row, gathers neighbour rows straight into registers and stores
// the row once with plain stores (float atomics top out at ~1.3 TB/s on this chip).
//
// Lane layout: a wave is split into NG = 64/G groups of G lanes. A group reads ONE neighbour
// row per step, each lane VEC contiguous floats (d = 128: G = 32, VEC = 4 -> 16 B per lane,
// 1 KiB per wave-instruction, two neighbours per instruction). U steps are issued back to back
// so 8 neighbour rows are in flight per wave. The 64 column indices (and weights) of a row
// chunk are read once, coalesced, and handed to the groups with ds_bpermute.
//
// Degree skew: a row owned by one wave runs as long as its slot list, so hub rows (thousands of
// in-edges against a median of tens) would set the kernel time. With a row-split plan
// (rgbx_row_split_t) the main kernel skips rows longer than the threshold; their slot lists are cut
// into fixed chunks summed by separate waves into a partial buffer, and a last pass adds each
// row's partials in chunk order (bitwise reproducible) and applies the epilogue.
#include "rgbx_common.h"
#include "spmm_internal.h"

namespace rgbx {
namespace {

struct SpmmArgs {
  const int* rowptr;
  const int* col;
  const float* w;
  const float* rs;
  const float* x;
  const float* y;
  const float* bias;
  float* out;
  int64_t ldx, ldy, ldo;
  int N, d;
  float a, b;
  int skip_longer;  // > 0: rows with more slots than this are left to the split-row kernels
};

// Weighted sum of the gathered rows of slots [start, end) into acc (this lane's VEC columns, base
// pointer xc). On return the NG groups' partial sums are folded: every lane of a column holds the total.
template <int G, int VEC, bool HAS_W>
__device__ __forceinline__ void accumulate_slots(const SpmmArgs& A, int start, int end, const float* xc,
                                                 bool active, int lane, int g, float (&acc)[VEC]) {
  constexpr int NG = kWave / G;
  constexpr int U = 4;
  for (int base = start; base < end; base += kWave) {
    const int n = min(kWave, end - base);
    int mycol = 0;
    float myw = 0.f;
    if (lane < n) {
      mycol = A.col[base + lane];
      if constexpr (HAS_W) myw = A.w[base + lane];
    }
    for (int k = 0; k < n; k += NG * U) {
      float v[U][VEC];
      float ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = k + u * NG + g;
        const int src = __shfl(mycol, idx & 63);
        if constexpr (HAS_W) ww[u] = __shfl(myw, idx & 63);
        const bool ok = active && idx < n;
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
        if (ok) load_vec<VEC>(v[u], xc + (int64_t)src * A.ldx);
        if constexpr (HAS_W) { if (!ok) ww[u] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if constexpr (HAS_W) acc[i] = fmaf(ww[u], v[u][i], acc[i]);
          else acc[i] += v[u][i];
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off);
  }
}

template <int VEC>
__device__ __forceinline__ void spmm_epilogue(const SpmmArgs& A, int row, int c, const float (&acc)[VEC]) {
  const float scale = A.rs ? A.a * A.rs[row] : A.a;
  float r[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) r[i] = scale * acc[i];
  if (A.y) {
    float yv[VEC];
    load_vec<VEC>(yv, A.y + (int64_t)row * A.ldy + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = fmaf(A.b, yv[i], r[i]);
  }
  if (A.bias) {
    float bv[VEC];
    load_vec<VEC>(bv, A.bias + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] += bv[i];
  }
  // non-temporal for the 16-byte path: the output row is read by nobody in this launch and competes with the gathered table
  // for L2 / Infinity Cache (plain SpMM at L 4.88 -> 4.80 ms; the APPNP K-loop, whose next step gathers what this one
  // wrote, still gains: 48.2 -> 47.9 ms for K = 10; tools/ab_lib.py, profiles/r04_ab_nt_stores.txt)
  if constexpr (VEC == 4) {
    using f4v = __attribute__((ext_vector_type(4))) float;
    f4v t = {r[0], r[1], r[2], r[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(A.out + (int64_t)row * A.ldo + c));
  } else {
    store_vec<VEC>(A.out + (int64_t)row * A.ldo + c, r);
  }
}

// G lanes per neighbour row, VEC floats per lane; columns beyond G*VEC are covered by an outer
// chunk loop (only taken for d > G*VEC, i.e. d > 256 on the float4 path).
template <int G, int VEC, bool HAS_W>
__global__ void __launch_bounds__(256) spmm_csr_kernel(const SpmmArgs A) {
  const int lane = threadIdx.x & 63;
  const int g = lane / G;
  const int t = lane % G;
  const int wpb = blockDim.x >> 6;
  const int wave0 = blockIdx.x * wpb + (threadIdx.x >> 6);
  const int wstride = gridDim.x * wpb;

  for (int row = wave0; row < A.N; row += wstride) {
    const int start = __builtin_amdgcn_readfirstlane(A.rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
    if (A.skip_longer > 0 && end - start > A.skip_longer) continue;  // split-row kernels own it
    for (int cbase = 0; cbase < A.d; cbase += G * VEC) {
      const int c = cbase + t * VEC;
      const bool active = c < A.d;
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      accumulate_slots<G, VEC, HAS_W>(A, start, end, A.x + c, active, lane, g, acc);
      if (g == 0 && active) spmm_epilogue<VEC>(A, row, c, acc);
    }
  }
}

// One wave per chunk of a long row: raw weighted sums (no epilogue) into partial[chunk, d].
template <int G, int VEC, bool HAS_W>
__global__ void __launch_bounds__(256)
spmm_chunk_kernel(const SpmmArgs A, int n_chunks, const int* __restrict__ chunk_begin,
                  const int* __restrict__ chunk_end, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int g = lane / G;
  const int t = lane % G;
  const int wpb = blockDim.x >> 6;
  for (int ch = blockIdx.x * wpb + (threadIdx.x >> 6); ch < n_chunks; ch += gridDim.x * wpb) {
    const int start = __builtin_amdgcn_readfirstlane(chunk_begin[ch]);
    const int end = __builtin_amdgcn_readfirstlane(chunk_end[ch]);
    for (int cbase = 0; cbase < A.d; cbase += G * VEC) {
      const int c = cbase + t * VEC;
      const bool active = c < A.d;
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      accumulate_slots<G, VEC, HAS_W>(A, start, end, A.x + c, active, lane, g, acc);
      if (g == 0 && active) store_vec<VEC>(partial + (int64_t)ch * A.d + c, acc);
    }
  }
}

// One wave per long row: partials added in chunk order, then the normal epilogue.
template <int VEC>
__global__ void __launch_bounds__(256)
spmm_combine_kernel(const SpmmArgs A, int n_long, const int* __restrict__ long_row,
                    const int* __restrict__ long_chunk_ptr, const float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_long; r += gridDim.x * wpb) {
    const int row = long_row[r];
    const int c0 = long_chunk_ptr[r], c1 = long_chunk_ptr[r + 1];
    for (int c = lane * VEC; c < A.d; c += kWave * VEC) {
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      for (int ch = c0; ch < c1; ++ch) {
        float p[VEC];
        load_vec<VEC>(p, partial + (int64_t)ch * A.d + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += p[i];
      }
      spmm_epilogue<VEC>(A, row, c, acc);
    }
  }
}

// One wave per long row, compact output: zlong[r, :] = rs[row] * (partials added in chunk order).
template <int VEC>
__global__ void __launch_bounds__(256)
spmm_combine_compact_kernel(const float* __restrict__ rs, int d, int n_long, const int* __restrict__ long_row,
                            const int* __restrict__ long_chunk_ptr, const float* __restrict__ partial,
                            float* __restrict__ zlong) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_long; r += gridDim.x * wpb) {
    const float scale = rs ? rs[long_row[r]] : 1.0f;
    const int c0 = long_chunk_ptr[r], c1 = long_chunk_ptr[r + 1];
    for (int c = lane * VEC; c < d; c += kWave * VEC) {
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      for (int ch = c0; ch < c1; ++ch) {
        float p[VEC];
        load_vec<VEC>(p, partial + (int64_t)ch * d + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += p[i];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] *= scale;
      store_vec<VEC>(zlong + (int64_t)r * d + c, acc);
    }
  }
}

template <int G>
int launch_long_compact(const SpmmArgs& A, const rgbx_row_split_t* sp, float* zlong, hipStream_t s) {
  int64_t cb = cdiv(sp->n_chunks, 4);
  if (cb > kMaxGrid) cb = kMaxGrid;
  if (A.w)
    spmm_chunk_kernel<G, 4, true><<<(int)cb, 256, 0, s>>>(A, sp->n_chunks, sp->chunk_begin, sp->chunk_end, sp->partial);
  else
    spmm_chunk_kernel<G, 4, false><<<(int)cb, 256, 0, s>>>(A, sp->n_chunks, sp->chunk_begin, sp->chunk_end, sp->partial);
  RGBX_CHECK_LAUNCH("spmm_chunk_kernel");
  int64_t lb = cdiv(sp->n_long, 4);
  if (lb > kMaxGrid) lb = kMaxGrid;
  spmm_combine_compact_kernel<4><<<(int)lb, 256, 0, s>>>(A.rs, A.d, sp->n_long, sp->long_row, sp->long_chunk_ptr,
                                                         sp->partial, zlong);
  RGBX_CHECK_LAUNCH("spmm_combine_compact_kernel");
  return RGBX_OK;
}

template <int G, int VEC>
int launch(const SpmmArgs& A, const rgbx_row_split_t* sp, hipStream_t s) {
  constexpr int kWavesPerBlock = 4;
  // One row per wave, one launch-time block per 4 rows, NO grid cap: rows differ in length, and letting the
  // dispatcher hand out fresh blocks balances them better than a fixed grid-stride assignment (measured at
  // |V|=2M, |E|=60M: 4.94 ms uncapped against 5.38 ms with the grid capped at 8192 blocks).
  const int64_t blocks = cdiv(A.N, kWavesPerBlock);
  if (A.w)
    spmm_csr_kernel<G, VEC, true><<<(int)blocks, 256, 0, s>>>(A);
  else
    spmm_csr_kernel<G, VEC, false><<<(int)blocks, 256, 0, s>>>(A);
  RGBX_CHECK_LAUNCH("spmm_csr_kernel");
  if (sp) {
    int64_t cb = cdiv(sp->n_chunks, kWavesPerBlock);
    if (cb > kMaxGrid) cb = kMaxGrid;
    if (A.w)
      spmm_chunk_kernel<G, VEC, true><<<(int)cb, 256, 0, s>>>(A, sp->n_chunks, sp->chunk_begin, sp->chunk_end,
                                                             sp->partial);
    else
      spmm_chunk_kernel<G, VEC, false><<<(int)cb, 256, 0, s>>>(A, sp->n_chunks, sp->chunk_begin, sp->chunk_end,
                                                              sp->partial);
    RGBX_CHECK_LAUNCH("spmm_chunk_kernel");
    int64_t lb = cdiv(sp->n_long, kWavesPerBlock);
    if (lb > kMaxGrid) lb = kMaxGrid;
    spmm_combine_kernel<VEC><<<(int)lb, 256, 0, s>>>(A, sp->n_long, sp->long_row, sp->long_chunk_ptr, sp->partial);
    RGBX_CHECK_LAUNCH("spmm_combine_kernel");
  }
  return RGBX_OK;
}

// ---- the same row gather with an epilogue over the finished rows (rgbx_spmm_csr_epilogue_f32) --------------------
// Layers that transform BEFORE they aggregate (in > out: every configuration the reference ships, initial_params.py:25-29 —
// F = 1433 -> 64 -> C = 7) end in this kernel, not in the fused aggregate+transform one; what follows the layer in the
// reference is then a pass over its [N, d] output: BatchNorm's column statistics (models/gcn.py:28) or log_softmax +
// NLLLoss + arg-max (gcn.py:31, itexperiments.py:429,434,624-626). A lane group of the gather holds a target's COMPLETE
// output row (d <= 256), so both come out of the registers here:
//  - column sums: a wave adds its 8 rows per column, the 4 waves of a 32-row tile meet in LDS, one [2, d] fp32 record per
//    tile; the records are added in fp64 in a fixed order (reduce_tile_stats) — as the fused kernel's MFMA-tile statistics;
//  - masked cross-entropy: max / first arg-max / sum-exp go through the group's lanes with shuffles; per tile one record
//    (nll sum, rows, hits) x mask groups; `out` receives the loss gradient, or nothing at all (statistics only).
// Rows are assigned statically (wave w of a tile owns rows 8 w .. 8 w + 7) so that every sum has a fixed order.
struct EpiArgs {
  float* stats_part;  // [tiles][2 * d] or NULL
  const int64_t* ce_y;
  const uint8_t* ce_mask;
  const float* ce_scale;
  double* ce_part;    // [tiles][3 * ce_groups] or NULL
  int ce_groups;
  int C;              // columns [C, d) are padding: no part in the loss, gradient 0
  // hub rows: RAW weighted sums finished by the split-row kernels, compact in hub order
  const int* long_row;
  const float* zlong;
  int threshold, n_long;
};

template <int G, bool HAS_W, bool CE>
__global__ void __launch_bounds__(256) spmm_tile_epilogue_kernel(const SpmmArgs A, const EpiArgs E) {
  constexpr int VEC = 4;
  constexpr int RPW = kTileRows / 4;  // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / G, t = lane % G;
  const int c = t * VEC;
  const bool active = c < A.d;
  const int row_base = blockIdx.x * kTileRows;
  // column statistics: sums of (x - cw) and (x - cw)^2 with cw = the wave's first row (a sample of the column: no
  // cancellation against the column's mean, see the tile records below)
  float s1[VEC] = {0.f, 0.f, 0.f, 0.f}, s2[VEC] = {0.f, 0.f, 0.f, 0.f}, cw[VEC] = {0.f, 0.f, 0.f, 0.f};
  double nll = 0.0, nll2 = 0.0;
  int cnt = 0, hit = 0, cnt2 = 0, hit2 = 0;
  const float sc = (CE && E.ce_scale) ? E.ce_scale[0] : 0.f;
  float bv[VEC] = {0.f, 0.f, 0.f, 0.f};
  if (A.bias && active) load_vec<VEC>(bv, A.bias + c);

  // The wave's RPW rows' slot ranges, mask bits and labels in ONE round of loads (lane l holds row l's): read row by row
  // they are three dependent global loads in front of every row — also of the rows no mask selects, which are most rows of an
  // eval forward (statistics under one mask, d = 8 at L: 0.55 -> see profiles/r05_epilogue_preload.txt).
  const int row0 = row_base + wave * RPW;
  int my_ptr = 0, my_bits = 1, my_tgt = -1;
  if (lane <= RPW && row0 + lane <= A.N) my_ptr = A.rowptr[row0 + lane];
  if constexpr (CE) {
    if (lane < RPW && row0 + lane < A.N) {
      my_bits = E.ce_mask ? (int)E.ce_mask[row0 + lane] : 1;
      my_bits = E.ce_groups == 2 ? (my_bits & 3) : (my_bits ? 1 : 0);
      if (my_bits) {
        const int64_t ti = E.ce_y[row0 + lane];
        if (ti >= 0 && ti < E.C) my_tgt = (int)ti;
      }
    }
  }

  // (unrolled for the loss forms only: 50-61 registers instead of 59-71 there, 70 instead of 64 for the statistics forms)
#pragma unroll CE ? RPW : 1
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr;
    if (row >= A.N) break;  // wave-uniform
    const int start = __builtin_amdgcn_readlane(my_ptr, rr);
    const int end = __builtin_amdgcn_readlane(my_ptr, rr + 1);
    int tgt = -1, bits = 1;
    if constexpr (CE) {
      bits = __builtin_amdgcn_readlane(my_bits, rr);
      tgt = __builtin_amdgcn_readlane(my_tgt, rr);
      if (tgt < 0 && !E.ce_scale) continue;  // not selected and nothing to store: the row is not even gathered
    }
    float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
    if (E.threshold > 0 && end - start > E.threshold) {  // hub row (wave-uniform): finished by the split-row kernels
      int lo = 0, hi = E.n_long - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (E.long_row[mid] < row) lo = mid + 1;
        else hi = mid;
      }
      if (active) load_vec<VEC>(acc, E.zlong + (int64_t)lo * A.d + c);
    } else if (!CE || tgt >= 0) {
      accumulate_slots<G, VEC, HAS_W>(A, start, end, A.x + c, active, lane, g, acc);
    }
    // every lane group now holds the row: the epilogue of spmm_epilogue, same operation order (same bits)
    const float scale = A.rs ? A.a * A.rs[row] : A.a;
    float r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = scale * acc[i];
    if (A.y && active) {
      float yv[VEC];
      load_vec<VEC>(yv, A.y + (int64_t)row * A.ldy + c);
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = fmaf(A.b, yv[i], r[i]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] += bv[i];
    if constexpr (!CE) {
      if (g == 0 && active) {
        if (A.out) {
          using f4v = __attribute__((ext_vector_type(4))) float;
          f4v o = {r[0], r[1], r[2], r[3]};
          __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(A.out + (int64_t)row * A.ldo + c));
        }
        if (rr == 0) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) cw[i] = r[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float dv = r[i] - cw[i];
          s1[i] += dv;
          s2[i] = fmaf(dv, dv, s2[i]);
        }
      }
    } else {
      float e[VEC] = {0.f, 0.f, 0.f, 0.f};
      float lse = 0.f;
      if (tgt >= 0) {
        // NLLLoss(log_softmax(z))_i = lse_i - z[i, y_i]; arg-max = the first maximal column (rgbx_masked_ce_fwd_f32)
        float best = -INFINITY;
        int arg = INT32_MAX;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const bool valid = active && c + i < E.C;
          if (valid && r[i] > best) { best = r[i]; arg = c + i; }
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
          const float ob = __shfl_xor(best, off);
          const int oa = __shfl_xor(arg, off);
          if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        float se = 0.f, tv = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const bool valid = active && c + i < E.C;
          e[i] = valid ? expf(r[i] - best) : 0.f;
          se += e[i];
          if (valid && c + i == tgt) tv = r[i];
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
          se += __shfl_xor(se, off);
          tv += __shfl_xor(tv, off);  // exactly one lane of the group holds the target column
        }
        lse = best + logf(se);
        const double term = (double)(lse - tv);
        const int h = arg == tgt ? 1 : 0;
        if (bits & 1) { nll += term; cnt += 1; hit += h; }
        if (bits & 2) { nll2 += term; cnt2 += 1; hit2 += h; }
      }
      if (E.ce_scale && g == 0 && active) {  // grad_scale * (softmax - onehot) on selected rows, 0 elsewhere (rgbx_masked_ce_bwd_f32)
        float gr[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const bool valid = c + i < E.C;
          gr[i] = (tgt >= 0 && valid) ? sc * (expf(r[i] - lse) - (c + i == tgt ? 1.f : 0.f)) : 0.f;
        }
        using f4v = __attribute__((ext_vector_type(4))) float;
        f4v o = {gr[0], gr[1], gr[2], gr[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(A.out + (int64_t)row * A.ldo + c));
      }
    }
  }
  if constexpr (!CE) {
    if (!E.stats_part) return;
    __shared__ float sh[4][768];
    if (g == 0 && active) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sh[wave][c + i] = s1[i];
        sh[wave][A.d + c + i] = s2[i];
        sh[wave][2 * A.d + c + i] = cw[i];
      }
    }
    __syncthreads();
    // The tile's record per column: (S = sum x, M2 = sum (x - S/n)^2), as the fused kernel's MFMA tiles write it
    // (spmm_linear.hip store_tile; the reducers form sum x^2 = M2 + S^2 / n in fp64). From the waves' shifted sums in fp64:
    // wave w with n_w rows has mean_w = cw + s1 / n_w and M2_w = s2 - s1^2 / n_w, the tile adds the spread of the wave means.
    const int n_t = A.N - row_base < kTileRows ? A.N - row_base : kTileRows;
    for (int k = threadIdx.x; k < A.d; k += 256) {
      double mean_w[4], m2 = 0.0, S = 0.0;
      int n_w[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int left = n_t - w * RPW;
        n_w[w] = left <= 0 ? 0 : (left < RPW ? left : RPW);
        mean_w[w] = 0.0;
        if (n_w[w]) {
          const double a = (double)sh[w][k] / n_w[w];
          mean_w[w] = (double)sh[w][2 * A.d + k] + a;
          const double within = (double)sh[w][A.d + k] - (double)sh[w][k] * a;
          m2 += within > 0.0 ? within : 0.0;
          S += n_w[w] * mean_w[w];
        }
      }
      const double mean_t = S / n_t;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (n_w[w]) m2 += n_w[w] * (mean_w[w] - mean_t) * (mean_w[w] - mean_t);
      E.stats_part[(int64_t)blockIdx.x * 2 * A.d + k] = (float)S;
      E.stats_part[(int64_t)blockIdx.x * 2 * A.d + A.d + k] = (float)m2;
    }
  } else {
    __shared__ double cew[4][6];
    if (lane == 0) {
      cew[wave][0] = nll;
      cew[wave][1] = (double)cnt;
      cew[wave][2] = (double)hit;
      cew[wave][3] = nll2;
      cew[wave][4] = (double)cnt2;
      cew[wave][5] = (double)hit2;
    }
    __syncthreads();
    const int W = 3 * E.ce_groups;
    if ((int)threadIdx.x < W) {
      const int k = threadIdx.x;
      E.ce_part[(int64_t)blockIdx.x * W + k] = (cew[0][k] + cew[1][k]) + (cew[2][k] + cew[3][k]);
    }
  }
}

template <int G>
int launch_epilogue(const SpmmArgs& A, const EpiArgs& E, hipStream_t s) {
  const int tiles = (int)cdiv(A.N, kTileRows);
  if (E.ce_part) {
    if (A.w) spmm_tile_epilogue_kernel<G, true, true><<<tiles, 256, 0, s>>>(A, E);
    else spmm_tile_epilogue_kernel<G, false, true><<<tiles, 256, 0, s>>>(A, E);
  } else {
    if (A.w) spmm_tile_epilogue_kernel<G, true, false><<<tiles, 256, 0, s>>>(A, E);
    else spmm_tile_epilogue_kernel<G, false, false><<<tiles, 256, 0, s>>>(A, E);
  }
  RGBX_CHECK_LAUNCH("spmm_tile_epilogue_kernel");
  return RGBX_OK;
}

template <int VEC>
int dispatch_groups(const SpmmArgs& A, const rgbx_row_split_t* sp, hipStream_t s) {
  const int lanes = (A.d + VEC - 1) / VEC;  // lanes needed to cover one row
  if (lanes <= 1) return launch<1, VEC>(A, sp, s);
  if (lanes <= 2) return launch<2, VEC>(A, sp, s);
  if (lanes <= 4) return launch<4, VEC>(A, sp, s);
  if (lanes <= 8) return launch<8, VEC>(A, sp, s);
  if (lanes <= 16) return launch<16, VEC>(A, sp, s);
  if (lanes <= 32) return launch<32, VEC>(A, sp, s);
  return launch<64, VEC>(A, sp, s);
}

int spmm_dispatch(SpmmArgs A, const rgbx_row_split_t* split, hipStream_t s) {
  const rgbx_row_split_t* sp = nullptr;
  if (split && split->threshold > 0 && split->n_chunks > 0) {
    if (split->n_long <= 0 || !split->chunk_begin || !split->chunk_end || !split->long_row ||
        !split->long_chunk_ptr || !split->partial)
      return fail(RGBX_E_ARG, "spmm: incomplete row-split plan");
    sp = split;
    A.skip_longer = split->threshold;
  }
  auto vec_ok = [&](int v) {
    const uintptr_t mask = (uintptr_t)v * 4 - 1;
    auto okp = [&](const void* p, int64_t ld) {
      return !p || (((reinterpret_cast<uintptr_t>(p) & mask) == 0) && (ld % v == 0));
    };
    return A.d % v == 0 && okp(A.x, A.ldx) && okp(A.y, A.ldy) && okp(A.out, A.ldo) && okp(A.bias, v) &&
           (!sp || okp(sp->partial, v));
  };
  if (vec_ok(4)) return dispatch_groups<4>(A, sp, s);
  if (vec_ok(2)) return dispatch_groups<2>(A, sp, s);
  return dispatch_groups<1>(A, sp, s);
}

}  // namespace

int spmm_long_rows_compact(const int* rowptr, const int* col, const float* w, const float* rs, const float* x,
                           int64_t ldx, int d, const rgbx_row_split_t* split, float* zlong, hipStream_t s) {
  if (!split || split->n_long <= 0 || split->n_chunks <= 0 || !split->chunk_begin || !split->chunk_end ||
      !split->long_row || !split->long_chunk_ptr || !split->partial || !zlong)
    return fail(RGBX_E_ARG, "spmm_long_rows: incomplete row-split plan");
  if (d % 4 || !aligned16(x) || ldx % 4 || !aligned16(split->partial) || !aligned16(zlong))
    return fail(RGBX_E_ALIGN, "spmm_long_rows: needs d %% 4 == 0 and 16-byte aligned x / partial / zlong");
  SpmmArgs A{rowptr, col, w, rs, x, nullptr, nullptr, zlong, ldx, 0, d, 0, d, 1.0f, 0.0f, 0};
  const int lanes = d / 4;
  if (lanes <= 1) return launch_long_compact<1>(A, split, zlong, s);
  if (lanes <= 2) return launch_long_compact<2>(A, split, zlong, s);
  if (lanes <= 4) return launch_long_compact<4>(A, split, zlong, s);
  if (lanes <= 8) return launch_long_compact<8>(A, split, zlong, s);
  if (lanes <= 16) return launch_long_compact<16>(A, split, zlong, s);
  if (lanes <= 32) return launch_long_compact<32>(A, split, zlong, s);
  return launch_long_compact<64>(A, split, zlong, s);
}

}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w,
                                 const float* rs, const float* x, int64_t ldx, const float* y,
                                 int64_t ldy, const float* bias, float* out, int64_t ldo, int64_t N,
                                 int64_t d, float a, float b, const rgbx_row_split_t* split,
                                 rgbx_stream_t stream) {
  if (N < 0 || d < 0) return fail(RGBX_E_ARG, "spmm: negative size");
  if (N == 0 || d == 0) return RGBX_OK;
  if (!rowptr || !col || !x || !out) return fail(RGBX_E_ARG, "spmm: null pointer");
  if (N >= INT32_MAX || d >= INT32_MAX) return fail(RGBX_E_RANGE, "spmm: N or d exceeds int32");
  if (ldx < d || ldo < d || (y && ldy < d)) return fail(RGBX_E_ARG, "spmm: leading dimension < d");
  if (out == x) return fail(RGBX_E_ARG, "spmm: out must not alias x");
  SpmmArgs A{rowptr, col, w, rs, x, y, bias, out, ldx, ldy, ldo, (int)N, (int)d, a, b, 0};
  return spmm_dispatch(A, split, (hipStream_t)stream);
}

extern "C" int rgbx_appnp_f32(const int32_t* rowptr, const int32_t* col, const float* w,
                              const float* h, int64_t ldh, float* out, float* tmp, int64_t ldo,
                              int64_t N, int64_t d, int K, float alpha, const rgbx_row_split_t* split,
                              rgbx_stream_t stream) {
  if (N < 0 || d < 0 || K < 0) return fail(RGBX_E_ARG, "appnp: negative size");
  if (N == 0 || d == 0) return RGBX_OK;
  if (!rowptr || !col || !h || !out || (K > 1 && !tmp)) return fail(RGBX_E_ARG, "appnp: null pointer");
  if (N >= INT32_MAX || d >= INT32_MAX) return fail(RGBX_E_RANGE, "appnp: N or d exceeds int32");
  if (ldh < d || ldo < d) return fail(RGBX_E_ARG, "appnp: leading dimension < d");
  if (out == h || tmp == h || (K > 1 && out == tmp)) return fail(RGBX_E_ARG, "appnp: buffers alias");
  hipStream_t s = (hipStream_t)stream;
  if (K == 0) {
    RGBX_HIP(hipMemcpy2DAsync(out, ldo * sizeof(float), h, ldh * sizeof(float), d * sizeof(float),
                              N, hipMemcpyDeviceToDevice, s));
    return RGBX_OK;
  }
  // Ping-pong so that step K-1 lands in `out`; step k reads z_k and the teleport term h
  // (pta.py:83: y = (1-alpha) * adj @ y + alpha * y0).
  const float* src = h;
  int64_t lds = ldh;
  for (int k = 0; k < K; ++k) {
    float* dst = ((K - 1 - k) % 2 == 0) ? out : tmp;
    SpmmArgs A{rowptr, col, w, nullptr, src, h, nullptr, dst, lds, ldh, ldo, (int)N, (int)d, 1.0f - alpha, alpha, 0};
    if (int rc = spmm_dispatch(A, split, s)) return rc;
    src = dst;
    lds = ldo;
  }
  return RGBX_OK;
}

extern "C" int rgbx_spmm_csr_epilogue_supported(int64_t d) { return d >= 4 && d % 4 == 0 && d <= 256; }

extern "C" int rgbx_spmm_csr_epilogue_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* rs,
                                          const float* x, int64_t ldx, const float* y, int64_t ldy, const float* bias,
                                          float* out, int64_t ldo, int64_t N, int64_t d, float a, float b,
                                          const rgbx_row_split_t* split, const rgbx_spmm_epilogue_t* epi,
                                          rgbx_stream_t stream) {
  if (!epi || (!epi->ce && !epi->out_colsums)) return fail(RGBX_E_ARG, "spmm_epilogue: no epilogue given (use rgbx_spmm_csr_f32)");
  if (epi->ce && epi->out_colsums) return fail(RGBX_E_ARG, "spmm_epilogue: out_colsums and the cross-entropy epilogue exclude each other");
  if (N < 0 || d <= 0) return fail(RGBX_E_ARG, "spmm_epilogue: bad size");
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "spmm_epilogue: N exceeds int32");
  if (!rgbx_spmm_csr_epilogue_supported(d))
    return fail(RGBX_E_SHAPE, "spmm_epilogue: needs d %% 4 == 0 and d <= 256 (got %lld); pad the rows", (long long)d);
  const rgbx_ce_epilogue_t* ce = epi->ce;
  const bool stats_only = ce && !ce->grad_scale;
  if (!rowptr || !col || !x || (!out && !stats_only)) return fail(RGBX_E_ARG, "spmm_epilogue: null pointer");
  if (ldx < d || (out && ldo < d) || (y && ldy < d)) return fail(RGBX_E_ARG, "spmm_epilogue: leading dimension < d");
  if (out == x) return fail(RGBX_E_ARG, "spmm_epilogue: out must not alias x");
  if (!aligned16(x) || ldx % 4 || (y && (!aligned16(y) || ldy % 4)) || (out && (!aligned16(out) || ldo % 4)) ||
      (bias && !aligned16(bias)))
    return fail(RGBX_E_ALIGN, "spmm_epilogue: x / y / out / bias must be 16-byte aligned with ld %% 4 == 0");
  int C = (int)d;
  if (ce) {
    if (!ce->y || !ce->stats || !ce->scratch) return fail(RGBX_E_ARG, "spmm_epilogue: incomplete cross-entropy epilogue");
    if (ce->mask_groups < 0 || ce->mask_groups > 2)
      return fail(RGBX_E_ARG, "spmm_epilogue: mask_groups must be 0, 1 or 2 (got %d)", (int)ce->mask_groups);
    if (ce->mask_groups == 2 && (ce->grad_scale || !ce->mask))
      return fail(RGBX_E_ARG, "spmm_epilogue: two statistics sets (mask_groups == 2) need a mask and no loss gradient");
    if (epi->n_classes < 0 || epi->n_classes > d) return fail(RGBX_E_ARG, "spmm_epilogue: n_classes must lie in [0, d]");
    if (epi->n_classes > 0) C = (int)epi->n_classes;
  }
  hipStream_t s = (hipStream_t)stream;
  const int tiles = (int)cdiv(N, kTileRows);
  if (N == 0) {
    if (ce) RGBX_HIP(hipMemsetAsync(ce->stats, 0, sizeof(double) * (ce->mask_groups == 2 ? 6 : 3), s));
    if (epi->out_colsums) RGBX_HIP(hipMemsetAsync(epi->out_colsums, 0, sizeof(double) * 2 * d, s));
    return RGBX_OK;
  }
  float* stats_part = nullptr;
  double* stats_part2 = nullptr;
  if (epi->out_colsums) {
    size_t need = 0;
    rgbx_spmm_linear_stats_workspace_bytes(N, d, &need);
    if (!epi->stats_ws || epi->stats_ws_bytes < need)
      return fail(RGBX_E_WS, "spmm_epilogue: statistics workspace %zu < %zu bytes", epi->stats_ws_bytes, need);
    if (reinterpret_cast<uintptr_t>(epi->stats_ws) % 8) return fail(RGBX_E_ALIGN, "spmm_epilogue: stats_ws must be 8-byte aligned");
    stats_part2 = static_cast<double*>(epi->stats_ws);
    stats_part = reinterpret_cast<float*>(stats_part2 + (size_t)kStatsGather * 2 * d);
  }
  SpmmArgs A{rowptr, col, w, rs, x, y, bias, out, ldx, ldy, ldo, (int)N, (int)d, a, b, 0};
  EpiArgs E{stats_part, ce ? ce->y : nullptr, ce ? ce->mask : nullptr, ce ? ce->grad_scale : nullptr,
            ce ? ce->scratch : nullptr, ce && ce->mask_groups == 2 ? 2 : 1, C, nullptr, nullptr, 0, 0};
  if (split && split->threshold > 0 && split->n_chunks > 0) {
    // hub rows first: raw chunk sums + ordered combine into the tail of the caller's scratch ([n_long, d] behind the
    // [n_chunks, d] partials); the tile kernel applies the epilogue to them like to any other row
    float* zl = split->partial ? split->partial + (size_t)split->n_chunks * d : nullptr;
    if (int rc = spmm_long_rows_compact(rowptr, col, w, nullptr, x, ldx, (int)d, split, zl, s)) return rc;
    E.long_row = split->long_row;
    E.zlong = zl;
    E.threshold = split->threshold;
    E.n_long = split->n_long;
  }
  const int lanes = (int)(d / 4);
  int rc;
  if (lanes <= 1) rc = launch_epilogue<1>(A, E, s);
  else if (lanes <= 2) rc = launch_epilogue<2>(A, E, s);
  else if (lanes <= 4) rc = launch_epilogue<4>(A, E, s);
  else if (lanes <= 8) rc = launch_epilogue<8>(A, E, s);
  else if (lanes <= 16) rc = launch_epilogue<16>(A, E, s);
  else if (lanes <= 32) rc = launch_epilogue<32>(A, E, s);
  else rc = launch_epilogue<64>(A, E, s);
  if (rc) return rc;
  if (ce) return reduce_ce_tiles(ce->scratch, tiles, ce->stats, ce->mask_groups == 2 ? 6 : 3, s);
  return reduce_tile_stats(stats_part, tiles, (int)(2 * d), stats_part2, epi->out_colsums, N, s);
}
