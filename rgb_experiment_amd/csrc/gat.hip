// GAT: fused attention score + edge-softmax + weighted aggregation, forward and backward, for
// gfx950. Replaces GATConv.forward/message + torch_geometric.utils.softmax behind reference
// models/gat.py:18-21,28,30 [PyG]: the four edge-sized temporaries ([E',H] x3, [E',H,C]) of that
// path never exist here. One wave owns one row of the (target- or source-grouped) CSR.
//
// Lane layout (host-computed `GatLayout`): a head occupies LPH = pow2ceil(C / VEC) consecutive
// lanes, each holding VEC channels; HPC heads sit side by side in a group of G lanes that reads
// one neighbour row per step; NG = 64 / G neighbours are read per wave-instruction. Heads beyond
// HPC are covered by an outer loop (softmax is per head, so head chunks are independent).
// Softmax runs online (running max / sum per lane, merged across the NG groups at the end);
// backward recomputes alpha from the saved per-(node, head) max and 1/sum.
#include "rgbx_common.h"

namespace rgbx {
namespace {

struct GatLayout {
  int H, C;
  int LPH;  // lanes per head (power of two)
  int HPC;  // heads per chunk
  int G;    // lanes per neighbour row (power of two, >= HPC * LPH)
};

constexpr float kNegBig = -1.0e30f;
constexpr int U = 4;

// Hub rows (rgbx_row_split_t): the row kernels skip rows longer than `threshold`; the same kernels,
// instantiated with CHUNK = true, walk the chunks of those rows and store per-chunk partial states
// (pacc [n_chunks, F], p0 / p1 [n_chunks, H]); a combine kernel merges them in chunk order.
struct SplitDev {
  int threshold;
  const int* chunk_row;
  const int* chunk_begin;
  const int* chunk_end;
  float* pacc;
  float* p0;
  float* p1;
  float* pacc2;  // training forward: the positive-score part of the chunk's state, [n_chunks, F]
  float* p2;     // and of its sum, [n_chunks, H]
};

template <int VEC>
__device__ __forceinline__ float dot_vec(const float (&a)[VEC], const float (&b)[VEC]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s = fmaf(a[i], b[i], s);
  return s;
}

// Sum over the LPH lanes of a head; every lane of the head ends with the total.
__device__ __forceinline__ float head_sum(float v, int LPH) {
  for (int off = LPH >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------
// scores: a_src[n,h] = <hfeat[n,h,:], att_src[h,:]>, a_dst likewise. Same lane layout as the aggregate
// kernels; the NG groups of a wave take NG consecutive rows per step.
template <int VEC>
__global__ void __launch_bounds__(256)
gat_scores_kernel(const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ att_src,
                  const float* __restrict__ att_dst, float* __restrict__ a_src,
                  float* __restrict__ a_dst, int n, const GatLayout L) {
  const int lane = threadIdx.x & 63;
  const int NG = kWave / L.G;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;
  for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
    const int head = hbase + hl;
    const bool lane_ok = hl < L.HPC && head < L.H && ch < L.C;
    const int cofs = head * L.C + ch;
    float as[VEC], ad[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) as[i] = ad[i] = 0.f;
    if (lane_ok) {
      load_vec<VEC>(as, att_src + cofs);
      load_vec<VEC>(ad, att_dst + cofs);
    }
    for (int row0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * NG; row0 < n; row0 += gridDim.x * wpb * NG) {
      const int row = row0 + g;
      const bool active = lane_ok && row < n;
      float h[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) h[i] = 0.f;
      if (active) load_vec<VEC>(h, hfeat + (int64_t)row * ldh + cofs);
      const float ss = head_sum(dot_vec<VEC>(h, as), L.LPH);
      const float sd = head_sum(dot_vec<VEC>(h, ad), L.LPH);
      if (active && ch == 0) {
        a_src[(int64_t)row * L.H + head] = ss;
        a_dst[(int64_t)row * L.H + head] = sd;
      }
    }
  }
}

// Backward of the scores, fused with the accumulation into the feature gradient:
//   g_hfeat[n,h,:] += g_a_src[n,h] * att_src[h,:] + g_a_dst[n,h] * att_dst[h,:]     (g_a_dst rows >= n_dst are 0)
//   part[block, 0, h, c] = sum over the block's rows of g_a_src[n,h] * hfeat[n,h,c];  part[block, 1, ...] with g_a_dst
// A second kernel adds the per-block partials in block order into g_att_src / g_att_dst (reproducible).
template <int VEC>
__global__ void __launch_bounds__(256)
gat_scores_bwd_kernel(const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ g_a_src,
                      const float* __restrict__ g_a_dst, int n_dst, const float* __restrict__ att_src,
                      const float* __restrict__ att_dst, float* __restrict__ g_hfeat, int64_t ldgh,
                      float* __restrict__ part, int n, const GatLayout L) {
  extern __shared__ float red[];  // [waves][2][G * VEC]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int NG = kWave / L.G;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;
  const int F = L.H * L.C;
  for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
    const int head = hbase + hl;
    const bool lane_ok = hl < L.HPC && head < L.H && ch < L.C;
    const int cofs = head * L.C + ch;
    float as[VEC], ad[VEC], ps[VEC], pd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) as[i] = ad[i] = ps[i] = pd[i] = 0.f;
    if (lane_ok) {
      load_vec<VEC>(as, att_src + cofs);
      load_vec<VEC>(ad, att_dst + cofs);
    }
    for (int row0 = (blockIdx.x * wpb + wave) * NG; row0 < n; row0 += gridDim.x * wpb * NG) {
      const int row = row0 + g;
      if (lane_ok && row < n) {
        float h[VEC];
        load_vec<VEC>(h, hfeat + (int64_t)row * ldh + cofs);
        const float gs = g_a_src[(int64_t)row * L.H + head];
        const float gd = row < n_dst ? g_a_dst[(int64_t)row * L.H + head] : 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          ps[i] = fmaf(gs, h[i], ps[i]);
          pd[i] = fmaf(gd, h[i], pd[i]);
        }
        if (g_hfeat) {  // NULL: rgbx_gat_bwd_src_f32 already folded the score terms into its store
          float gh[VEC];
          load_vec<VEC>(gh, g_hfeat + (int64_t)row * ldgh + cofs);
#pragma unroll
          for (int i = 0; i < VEC; ++i) gh[i] = fmaf(gs, as[i], fmaf(gd, ad[i], gh[i]));
          store_vec<VEC>(g_hfeat + (int64_t)row * ldgh + cofs, gh);
        }
      }
    }
    // fold the NG row groups, then the block's waves (fixed order), then one partial record per block
    for (int off = 32; off >= L.G; off >>= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        ps[i] += __shfl_xor(ps[i], off);
        pd[i] += __shfl_xor(pd[i], off);
      }
    }
    const int width = L.G * VEC;
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        red[(wave * 2 + 0) * width + t * VEC + i] = ps[i];
        red[(wave * 2 + 1) * width + t * VEC + i] = pd[i];
      }
    }
    __syncthreads();
    if (wave == 0 && g == 0 && lane_ok) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float s0 = 0.f, s1 = 0.f;
        for (int w = 0; w < wpb; ++w) {
          s0 += red[(w * 2 + 0) * width + t * VEC + i];
          s1 += red[(w * 2 + 1) * width + t * VEC + i];
        }
        part[((int64_t)blockIdx.x * 2 + 0) * F + cofs + i] = s0;
        part[((int64_t)blockIdx.x * 2 + 1) * F + cofs + i] = s1;
      }
    }
  }
}

// 32 consecutive columns per block; 8 lanes per column stride the per-block partial records, then the 8
// sub-sums are added in lane order (fixed summation order).
__global__ void __launch_bounds__(256)
gat_scores_bwd_finish_kernel(const float* __restrict__ part, int n_blocks, int F, float* __restrict__ g_att_src,
                             float* __restrict__ g_att_dst) {
  __shared__ float sh[8][32];
  const int j = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + j;  // column of the [2, F] result
  float s = 0.f;
  if (c < 2 * F) {
    const int which = c / F, col = c % F;
    for (int b = q; b < n_blocks; b += 8) s += part[((int64_t)b * 2 + which) * F + col];
  }
  sh[q][j] = s;
  __syncthreads();
  if (q == 0 && c < 2 * F) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[k][j];
    (c / F == 0 ? g_att_src : g_att_dst)[c % F] = t;
  }
}

// ------------------------------------------------------------------------------------------
// Residency: 256-thread workgroups are admitted per CU by VGPRs (<= 64 for 8 waves per SIMD) AND by SGPRs (<= 80 for 8
// workgroups; this kernel wanted 93 = 7): both are capped here — 16 scalars live in VGPR lanes instead, nothing goes
// to scratch — which is worth 2.5 % on the H=8, C=16 forward (5.16 -> 5.03 ms at L).
// TRAIN: the launch also stores, per (target, head), the part of the aggregate and of the attention mass that comes
// from edges with a POSITIVE pre-activation score s = a_src[j] + a_dst[i]:
//   out_pos[i,h,:] = sum_{p: s_p > 0} alpha_p hfeat[col[p],h,:],   a_pos[i,h] = sum_{p: s_p > 0} alpha_p.
// LeakyReLU's derivative takes two values, so the target-side score gradient of the backward,
//   g_a_dst[i,h] = sum_p alpha_p (<gout_i, h_j> - <gout_i, out_i>) lrelu'(s_p)
//                = (1 - slope) (<gout_i, out_pos_i> - <gout_i, out_i> a_pos_i)        (sum_p alpha_p = 1),
// becomes a per-node expression (rgbx_gat_bwd_prep_f32) instead of a per-edge tensor ds[E', H] written by the
// source-side pass and summed per target by another launch.
template <int VEC, bool CHUNK, bool TRAIN>
__global__ void __launch_bounds__(256, TRAIN ? 7 : 8) __attribute__((amdgpu_num_sgpr(80)))
gat_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
               const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ a_src,
               const float* __restrict__ att_src, float* __restrict__ a_dst,
               const float* __restrict__ att_dst,
               const float* __restrict__ oscale, const float* __restrict__ bias, float* __restrict__ out,
               int64_t ldo,
               float* __restrict__ m_out, float* __restrict__ rden_out, float* __restrict__ opos_out,
               float* __restrict__ apos_out, int N, float slope,
               const GatLayout L, const SplitDev sp) {
  constexpr int U = TRAIN ? 3 : 4;  // neighbour rows in flight per lane group (TRAIN: VEC + 1 more accumulators)
  const int lane = threadIdx.x & 63;
  const int NG = kWave / L.G;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;

  const int F = L.H * L.C;
  for (int item = blockIdx.x * wpb + (threadIdx.x >> 6); item < N; item += gridDim.x * wpb) {
    int row, start, end;
    if constexpr (CHUNK) {
      row = __builtin_amdgcn_readfirstlane(sp.chunk_row[item]);
      start = __builtin_amdgcn_readfirstlane(sp.chunk_begin[item]);
      end = __builtin_amdgcn_readfirstlane(sp.chunk_end[item]);
    } else {
      row = item;
      start = __builtin_amdgcn_readfirstlane(rowptr[row]);
      end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
      if (sp.threshold > 0 && end - start > sp.threshold) continue;  // the chunk + combine kernels own it
    }
    for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
      const int head = hbase + hl;
      const bool active = hl < L.HPC && head < L.H && ch < L.C;
      const int cofs = head * L.C + ch;
      // With att_dst given, the target's own score <h_i, att_dst> is formed here from its row (one more row next to
      // the ~30 gathered ones) instead of by a pass of its own over hfeat (gat_scores_kernel); a forward that
      // prepares a backward stores it for the per-target records.
      float ad = 0.f;
      if (att_dst) {
        float hi[VEC], atd[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) hi[i] = atd[i] = 0.f;
        if (active) {
          load_vec<VEC>(hi, hfeat + (int64_t)row * ldh + cofs);
          load_vec<VEC>(atd, att_dst + cofs);
        }
        ad = head_sum(dot_vec<VEC>(hi, atd), L.LPH);
        if (a_dst && active && ch == 0 && g == 0) a_dst[(int64_t)row * L.H + head] = ad;  // a_dst: output here
      } else if (active) {
        ad = a_dst[(int64_t)row * L.H + head];
      }
      // With att_src given, the source score <h_j, att_src> is formed from the gathered row itself (a few
      // cross-lane adds) instead of a fifth cache-line request per edge for a_src[j].
      float att[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) att[i] = 0.f;
      if (att_src && active) load_vec<VEC>(att, att_src + cofs);
      float m = kNegBig, l = 0.f;
      float acc[VEC];
      float lp = 0.f;
      float accp[TRAIN ? VEC : 1];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      if constexpr (TRAIN) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) accp[i] = 0.f;
      }

      for (int base = start; base < end; base += kWave) {
        const int n = min(kWave, end - base);
        const int mycol = lane < n ? col[base + lane] : 0;
        for (int k = 0; k < n; k += NG * U) {
          float v[U][VEC];
          float as[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = k + u * NG + g;
            const int src = __shfl(mycol, idx & 63);
            ok[u] = active && idx < n;
            as[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
            if (ok[u]) {
              if (!att_src) as[u] = a_src[(int64_t)src * L.H + head];
              load_vec<VEC>(v[u], hfeat + (int64_t)src * ldh + cofs);
            }
          }
          if (att_src) {
#pragma unroll
            for (int u = 0; u < U; ++u) as[u] = head_sum(dot_vec<VEC>(v[u], att), L.LPH);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float s = as[u] + ad;
            const float e = s > 0.f ? s : slope * s;
            const float mn = ok[u] ? fmaxf(m, e) : m;
            const float sc = expf(m - mn);
            const float p = ok[u] ? expf(e - mn) : 0.f;
            l = fmaf(l, sc, p);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(acc[i], sc, p * v[u][i]);
            if constexpr (TRAIN) {
              const float pp = s > 0.f ? p : 0.f;
              lp = fmaf(lp, sc, pp);
#pragma unroll
              for (int i = 0; i < VEC; ++i) accp[i] = fmaf(accp[i], sc, pp * v[u][i]);
            }
            m = mn;
          }
        }
      }
      // merge the NG online-softmax states
      for (int off = 32; off >= L.G; off >>= 1) {
        const float m2 = __shfl_xor(m, off);
        const float l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float s1 = expf(m - mn), s2 = expf(m2 - mn);
        l = l * s1 + l2 * s2;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float a2 = __shfl_xor(acc[i], off);
          acc[i] = acc[i] * s1 + a2 * s2;
        }
        if constexpr (TRAIN) {
          lp = lp * s1 + __shfl_xor(lp, off) * s2;
#pragma unroll
          for (int i = 0; i < VEC; ++i) accp[i] = accp[i] * s1 + __shfl_xor(accp[i], off) * s2;
        }
        m = mn;
      }
      if (g == 0 && active) {
        if constexpr (CHUNK) {  // un-normalised online-softmax state of this chunk
          store_vec<VEC>(sp.pacc + (int64_t)item * F + cofs, acc);
          if constexpr (TRAIN) store_vec<VEC>(sp.pacc2 + (int64_t)item * F + cofs, accp);
          if (ch == 0) {
            sp.p0[(int64_t)item * L.H + head] = m;
            sp.p1[(int64_t)item * L.H + head] = l;
            if constexpr (TRAIN) sp.p2[(int64_t)item * L.H + head] = lp;
          }
        } else {
          const float rd = l > 0.f ? 1.0f / (l + 1e-16f) : 0.f;
          float r[VEC], bv[VEC], sv[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) { bv[i] = 0.f; sv[i] = 1.f; }
          if (bias) load_vec<VEC>(bv, bias + cofs);
          if (oscale) load_vec<VEC>(sv, oscale + cofs);
#pragma unroll
          for (int i = 0; i < VEC; ++i) r[i] = acc[i] * rd * sv[i] + bv[i];
          store_vec<VEC>(out + (int64_t)row * ldo + cofs, r);
          if constexpr (TRAIN) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) accp[i] *= rd;
            store_vec<VEC>(opos_out + (int64_t)row * F + cofs, accp);
          }
          if (ch == 0) {
            m_out[(int64_t)row * L.H + head] = l > 0.f ? m : 0.f;
            rden_out[(int64_t)row * L.H + head] = rd;
            if constexpr (TRAIN) apos_out[(int64_t)row * L.H + head] = lp * rd;
          }
        }
      }
    }
  }
}

// One wave per hub row: merge the chunk states in chunk order, normalise, store.
template <int VEC>
__global__ void __launch_bounds__(256)
gat_fwd_combine_kernel(int n_long, const int* __restrict__ long_row, const int* __restrict__ long_chunk_ptr,
                       const float* __restrict__ oscale, const float* __restrict__ bias, float* __restrict__ out, int64_t ldo, float* __restrict__ m_out,
                       float* __restrict__ rden_out, float* __restrict__ opos_out, float* __restrict__ apos_out,
                       const GatLayout L, const SplitDev sp) {
  const int lane = threadIdx.x & 63;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;
  const int F = L.H * L.C;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_long; r += gridDim.x * wpb) {
    const int row = long_row[r];
    const int c0 = long_chunk_ptr[r], c1 = long_chunk_ptr[r + 1];
    for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
      const int head = hbase + hl;
      const bool active = g == 0 && hl < L.HPC && head < L.H && ch < L.C;
      if (!active) continue;
      const int cofs = head * L.C + ch;
      float m = kNegBig, l = 0.f, lp = 0.f;
      float acc[VEC], accp[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = accp[i] = 0.f;
      for (int c = c0; c < c1; ++c) {
        const float m2 = sp.p0[(int64_t)c * L.H + head];
        const float l2 = sp.p1[(int64_t)c * L.H + head];
        float a2[VEC];
        load_vec<VEC>(a2, sp.pacc + (int64_t)c * F + cofs);
        const float mn = fmaxf(m, m2);
        const float s1 = expf(m - mn), s2 = expf(m2 - mn);
        l = l * s1 + l2 * s2;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = acc[i] * s1 + a2[i] * s2;
        if (opos_out) {
          load_vec<VEC>(a2, sp.pacc2 + (int64_t)c * F + cofs);
          lp = lp * s1 + sp.p2[(int64_t)c * L.H + head] * s2;
#pragma unroll
          for (int i = 0; i < VEC; ++i) accp[i] = accp[i] * s1 + a2[i] * s2;
        }
        m = mn;
      }
      const float rd = l > 0.f ? 1.0f / (l + 1e-16f) : 0.f;
      if (opos_out) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) accp[i] *= rd;
        store_vec<VEC>(opos_out + (int64_t)row * F + cofs, accp);
        if (ch == 0) apos_out[(int64_t)row * L.H + head] = lp * rd;
      }
      float bv[VEC], sv[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { bv[i] = 0.f; sv[i] = 1.f; }
      if (bias) load_vec<VEC>(bv, bias + cofs);
      if (oscale) load_vec<VEC>(sv, oscale + cofs);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = acc[i] * rd * sv[i] + bv[i];
      store_vec<VEC>(out + (int64_t)row * ldo + cofs, acc);
      if (ch == 0) {
        m_out[(int64_t)row * L.H + head] = l > 0.f ? m : 0.f;
        rden_out[(int64_t)row * L.H + head] = rd;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward, target side: dsum[i,h] = <gout_i, out_i>, g_a_dst[i,h] = sum_p ds_p.
// alpha = exp(e - m) * rden = exp(e - (m - log(rden))): the per-(target, head) record of the backward keeps the one
// combined constant (one register per edge in flight less in the source-side kernel). Targets without in-edges
// (rden == 0) are never gathered.
__device__ __forceinline__ float softmax_shift(float m, float rden) { return rden > 0.f ? m - logf(rden) : 0.f; }

template <int VEC>
__global__ void __launch_bounds__(256)
gat_bwd_dst_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                   const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ a_src,
                   const float* __restrict__ a_dst, const float* __restrict__ m_in,
                   const float* __restrict__ rden_in, const float* __restrict__ out, int64_t ldo,
                   const float* __restrict__ gout, int64_t ldg, float4* __restrict__ nodeq_out,
                   float* __restrict__ g_a_dst, int N, float slope, const GatLayout L) {
  const int lane = threadIdx.x & 63;
  const int NG = kWave / L.G;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;

  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < N; row += gridDim.x * wpb) {
    const int start = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
    for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
      const int head = hbase + hl;
      const bool active = hl < L.HPC && head < L.H && ch < L.C;
      const int cofs = head * L.C + ch;
      float go[VEC], o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) go[i] = o[i] = 0.f;
      float ad = 0.f, mi = 0.f, rd = 0.f;
      if (active) {
        load_vec<VEC>(go, gout + (int64_t)row * ldg + cofs);
        load_vec<VEC>(o, out + (int64_t)row * ldo + cofs);
        ad = a_dst[(int64_t)row * L.H + head];
        mi = m_in[(int64_t)row * L.H + head];
        rd = rden_in[(int64_t)row * L.H + head];
      }
      const float dsum = head_sum(dot_vec<VEC>(go, o), L.LPH);
      float acc = 0.f;

      for (int base = start; base < end; base += kWave) {
        const int n = min(kWave, end - base);
        const int mycol = lane < n ? col[base + lane] : 0;
        for (int k = 0; k < n; k += NG * U) {
          float v[U][VEC];
          float as[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = k + u * NG + g;
            const int src = __shfl(mycol, idx & 63);
            ok[u] = active && idx < n;
            as[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
            if (ok[u]) {
              as[u] = a_src[(int64_t)src * L.H + head];
              load_vec<VEC>(v[u], hfeat + (int64_t)src * ldh + cofs);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float dal = head_sum(dot_vec<VEC>(go, v[u]), L.LPH);
            const float s = as[u] + ad;
            const float e = s > 0.f ? s : slope * s;
            const float alpha = ok[u] ? expf(e - mi) * rd : 0.f;
            acc = fmaf(alpha * (dal - dsum), s > 0.f ? 1.f : slope, acc);
          }
        }
      }
      for (int off = 32; off >= L.G; off >>= 1) acc += __shfl_xor(acc, off);
      if (g == 0 && active && ch == 0) {
        // everything the source-side pass needs about target (row, head), in ONE 16-byte record
        nodeq_out[(int64_t)row * L.H + head] = make_float4(ad, softmax_shift(mi, rd), dsum, 0.f);
        g_a_dst[(int64_t)row * L.H + head] = acc;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward, per-target record only (no neighbour loop): nodeq[i,h] = (a_dst, m - log(rden), <gout_i, out_i>_h, 0).
template <int VEC>
__global__ void __launch_bounds__(256)
gat_bwd_prep_kernel(const float* __restrict__ a_dst, const float* __restrict__ m_in,
                    const float* __restrict__ rden_in, const float* __restrict__ out, int64_t ldo,
                    const float* __restrict__ bias, const float* __restrict__ gout, int64_t ldg,
                    float4* __restrict__ nodeq_out, const float* __restrict__ opos, const float* __restrict__ apos,
                    float one_minus_slope, float* __restrict__ g_a_dst, int N, const GatLayout L) {
  const int lane = threadIdx.x & 63;
  const int t = lane % L.G;
  const int g = lane / L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int NG = kWave / L.G;
  const int wpb = blockDim.x >> 6;
  // NG rows per wave step: group g of the wave takes row (base + g)
  for (int row0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * NG; row0 < N; row0 += gridDim.x * wpb * NG) {
    const int row = row0 + g;
    for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
      const int head = hbase + hl;
      const bool active = row < N && hl < L.HPC && head < L.H && ch < L.C;
      const int cofs = head * L.C + ch;
      float go[VEC], o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) go[i] = o[i] = 0.f;
      if (active) {
        load_vec<VEC>(go, gout + (int64_t)row * ldg + cofs);
        load_vec<VEC>(o, out + (int64_t)row * ldo + cofs);
        if (bias) {  // `out` was stored with the bias added: the softmax Jacobian needs the bare aggregate
          float bv[VEC];
          load_vec<VEC>(bv, bias + cofs);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] -= bv[i];
        }
      }
      const float dsum = head_sum(dot_vec<VEC>(go, o), L.LPH);
      float dpos = 0.f;
      if (opos) {  // <gout_i, out_pos_i>: see gat_fwd_kernel<.., TRAIN>
        float op[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) op[i] = 0.f;
        if (active) load_vec<VEC>(op, opos + (int64_t)row * ((int64_t)L.H * L.C) + cofs);
        dpos = head_sum(dot_vec<VEC>(go, op), L.LPH);
      }
      if (active && ch == 0) {
        const int64_t q = (int64_t)row * L.H + head;
        nodeq_out[q] = make_float4(a_dst[q], softmax_shift(m_in[q], rden_in[q]), dsum, 0.f);
        if (opos) g_a_dst[q] = one_minus_slope * (dpos - dsum * apos[q]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward, source side, over the transposed CSR: row = source j, col_t[p] = target i.
// IDX = uint32_t when every element offset into gout / nodeq fits 32 bits (two VGPRs per load in flight instead of
// four): with the combined softmax constant that brings the kernel from 78 to 69 registers = 7 waves per SIMD
// instead of 6 (9.98 -> 9.58 ms for the whole backward at L); with 3 instead of 4 neighbour rows in flight per lane
// group and the SGPR cap it fits 59 VGPRs / 78 SGPRs = 8 waves (9.28 -> 9.17 ms; H=1, C=128: 8.59 -> 8.26 ms).
template <int VEC, bool CHUNK, typename IDX>
__global__ void __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(80)))
gat_bwd_src_kernel(const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                   const float* __restrict__ hfeat, int64_t ldh, const float* __restrict__ a_src,
                   const float4* __restrict__ nodeq, const float* __restrict__ gout, int64_t ldg,
                   float* __restrict__ g_hfeat,
                   int64_t ldgh, float* __restrict__ g_a_src, float* __restrict__ ds_out,
                   const float* __restrict__ att2, const float* __restrict__ g_a_dst, int N, float slope,
                   const GatLayout L, const SplitDev sp) {
  constexpr int U = 3;  // neighbour rows in flight per lane group: 3 x 8 waves per SIMD beat 4 x 7 (registers)
  const int lane = threadIdx.x & 63;
  const int NG = kWave / L.G;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;

  const int F = L.H * L.C;
  for (int item = blockIdx.x * wpb + (threadIdx.x >> 6); item < N; item += gridDim.x * wpb) {
    int row, start, end;
    if constexpr (CHUNK) {
      row = __builtin_amdgcn_readfirstlane(sp.chunk_row[item]);
      start = __builtin_amdgcn_readfirstlane(sp.chunk_begin[item]);
      end = __builtin_amdgcn_readfirstlane(sp.chunk_end[item]);
    } else {
      row = item;
      start = __builtin_amdgcn_readfirstlane(rowptr_t[row]);
      end = __builtin_amdgcn_readfirstlane(rowptr_t[row + 1]);
      if (sp.threshold > 0 && end - start > sp.threshold) continue;
    }
    for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
      const int head = hbase + hl;
      const bool active = hl < L.HPC && head < L.H && ch < L.C;
      const int cofs = head * L.C + ch;
      float hj[VEC], acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) hj[i] = acc[i] = 0.f;
      float as = 0.f;
      if (active) {
        load_vec<VEC>(hj, hfeat + (int64_t)row * ldh + cofs);
        if (a_src) as = a_src[(int64_t)row * L.H + head];
      }
      if (!a_src) {  // the source's own score from the row it holds anyway (the forward formed it the same way)
        float ats[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) ats[i] = 0.f;
        if (active) load_vec<VEC>(ats, att2 + cofs);
        as = head_sum(dot_vec<VEC>(hj, ats), L.LPH);
      }
      float acc_as = 0.f;

      for (int base = start; base < end; base += kWave) {
        const int n = min(kWave, end - base);
        const int mycol = lane < n ? col_t[base + lane] : 0;
        for (int k = 0; k < n; k += NG * U) {
          float v[U][VEC];
          float ad[U], sh[U], dsm[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = k + u * NG + g;
            const int tgt = __shfl(mycol, idx & 63);
            ok[u] = active && idx < n;
            ad[u] = sh[u] = dsm[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
            if (ok[u]) {
              const float4 q = nodeq[(IDX)tgt * (IDX)L.H + (IDX)head];  // (a_dst, max - log(1/sum), dsum, -)
              ad[u] = q.x;
              sh[u] = q.y;
              dsm[u] = q.z;
              load_vec<VEC>(v[u], gout + ((IDX)tgt * (IDX)ldg + (IDX)cofs));
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float dal = head_sum(dot_vec<VEC>(v[u], hj), L.LPH);
            const float s = as + ad[u];
            const float e = s > 0.f ? s : slope * s;
            const float alpha = ok[u] ? expf(e - sh[u]) : 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(alpha, v[u][i], acc[i]);
            const float dsv = alpha * (dal - dsm[u]) * (s > 0.f ? 1.f : slope);
            acc_as += dsv;
            // d loss / d score of this edge and head, in transposed-slot order: the target side sums it
            // per target afterwards (a width-H segment sum) instead of re-gathering every source row
            if (ds_out && ok[u] && ch == 0) ds_out[(int64_t)(base + k + u * NG + g) * L.H + head] = dsv;
          }
        }
      }
      for (int off = 32; off >= L.G; off >>= 1) {
        acc_as += __shfl_xor(acc_as, off);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off);
      }
      if (g == 0 && active) {
        if constexpr (CHUNK) {
          store_vec<VEC>(sp.pacc + (int64_t)item * F + cofs, acc);
          if (ch == 0) sp.p0[(int64_t)item * L.H + head] = acc_as;
        } else {
          if (g_a_dst) {
            // backward of the scores a_src = <h, att_src>, a_dst = <h, att_dst>, folded into this store:
            //   g_hfeat[j,h,:] += g_a_src[j,h] att_src[h,:] + g_a_dst[j,h] att_dst[h,:]     (att2 = [att_src; att_dst])
            float ats[VEC], atd[VEC];
            load_vec<VEC>(ats, att2 + cofs);
            load_vec<VEC>(atd, att2 + F + cofs);
            const float gd = g_a_dst[(int64_t)row * L.H + head];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(acc_as, ats[i], fmaf(gd, atd[i], acc[i]));
          }
          store_vec<VEC>(g_hfeat + (int64_t)row * ldgh + cofs, acc);
          if (ch == 0) g_a_src[(int64_t)row * L.H + head] = acc_as;
        }
      }
    }
  }
}

// One wave per hub source row: chunk sums added in chunk order.
template <int VEC>
__global__ void __launch_bounds__(256)
gat_bwd_src_combine_kernel(int n_long, const int* __restrict__ long_row, const int* __restrict__ long_chunk_ptr,
                           float* __restrict__ g_hfeat, int64_t ldgh, float* __restrict__ g_a_src,
                           const float* __restrict__ att2, const float* __restrict__ g_a_dst, const GatLayout L,
                           const SplitDev sp) {
  const int lane = threadIdx.x & 63;
  const int g = lane / L.G;
  const int t = lane % L.G;
  const int hl = t / L.LPH;
  const int ch = (t % L.LPH) * VEC;
  const int wpb = blockDim.x >> 6;
  const int F = L.H * L.C;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_long; r += gridDim.x * wpb) {
    const int row = long_row[r];
    const int c0 = long_chunk_ptr[r], c1 = long_chunk_ptr[r + 1];
    for (int hbase = 0; hbase < L.H; hbase += L.HPC) {
      const int head = hbase + hl;
      if (!(g == 0 && hl < L.HPC && head < L.H && ch < L.C)) continue;
      const int cofs = head * L.C + ch;
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      float as = 0.f;
      for (int c = c0; c < c1; ++c) {
        float a2[VEC];
        load_vec<VEC>(a2, sp.pacc + (int64_t)c * F + cofs);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += a2[i];
        as += sp.p0[(int64_t)c * L.H + head];
      }
      if (g_a_dst) {
        float ats[VEC], atd[VEC];
        load_vec<VEC>(ats, att2 + cofs);
        load_vec<VEC>(atd, att2 + F + cofs);
        const float gd = g_a_dst[(int64_t)row * L.H + head];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(as, ats[i], fmaf(gd, atd[i], acc[i]));
      }
      store_vec<VEC>(g_hfeat + (int64_t)row * ldgh + cofs, acc);
      if (ch == 0) g_a_src[(int64_t)row * L.H + head] = as;
    }
  }
}

// ------------------------------------------------------------------------------------------
// `train`: the caller's scratch holds 2F + 3H floats per chunk (training forward) instead of F + 2H. The two [n_chunks, F]
// arrays come FIRST, so that both start at a multiple of F floats from the (16-byte aligned) scratch: with the H-sized
// arrays in between, pacc2 sat 2 * n_chunks * H floats further — off the 16-byte grid for an odd chunk count at one head, the
// vector width fell from 4 to 2, and a head of more than 128 channels (C = 132 ... 256) no longer fitted a wave's 64 lanes
// ("needs 128 lanes per head"; whole-model fuzz, round 4: seed 283, two nodes with 4,526 edges between them).
int split_view(const rgbx_row_split_t* split, int H, int C, bool train, SplitDev* sd, const char* name) {
  *sd = SplitDev{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!split || split->threshold <= 0 || split->n_chunks <= 0) return RGBX_OK;
  if (split->n_long <= 0 || !split->chunk_row || !split->chunk_begin || !split->chunk_end || !split->long_row ||
      !split->long_chunk_ptr || !split->partial)
    return fail(RGBX_E_ARG, "%s: incomplete row-split plan", name);
  const int64_t F = (int64_t)H * C;
  sd->threshold = split->threshold;
  sd->chunk_row = split->chunk_row;
  sd->chunk_begin = split->chunk_begin;
  sd->chunk_end = split->chunk_end;
  sd->pacc = split->partial;                                                 // [n_chunks, F]
  sd->pacc2 = train ? sd->pacc + (int64_t)split->n_chunks * F : nullptr;     // [n_chunks, F]  (training forward only)
  sd->p0 = sd->pacc + (int64_t)split->n_chunks * F * (train ? 2 : 1);        // [n_chunks, H]
  sd->p1 = sd->p0 + (int64_t)split->n_chunks * H;                            // [n_chunks, H]
  sd->p2 = train ? sd->p1 + (int64_t)split->n_chunks * H : nullptr;          // [n_chunks, H]  (training forward only)
  return RGBX_OK;
}

int pow2ceil(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

// VEC must divide C so that a lane's channels stay inside one head.
int pick_vec(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
  for (int v : {4, 2}) {
    bool ok = C % v == 0;
    for (const void* p : ptrs) ok = ok && (reinterpret_cast<uintptr_t>(p) % (v * 4) == 0);
    for (int64_t ld : lds) ok = ok && (ld % v == 0);
    if (ok) return v;
  }
  return 1;
}

int make_layout(int H, int C, int vec, GatLayout* L, const char* name) {
  const int lph = pow2ceil((C + vec - 1) / vec);
  if (lph > kWave)
    return fail(RGBX_E_SHAPE, "%s: C=%d needs %d lanes per head (> 64) at vector width %d", name, C,
                lph, vec);
  L->H = H;
  L->C = C;
  L->LPH = lph;
  L->HPC = std::min(H, kWave / lph);
  L->G = pow2ceil(L->HPC * lph);
  return RGBX_OK;
}

int gat_grid(int64_t N) {  // one row per wave, no cap (see spmm.hip: uncapped grids balance ragged rows better)
  return (int)cdiv(N, 4);
}

int check_common(int64_t N, int H, int C, const char* name) {
  if (N < 0 || H <= 0 || C <= 0) return fail(RGBX_E_ARG, "%s: bad size", name);
  if (N >= INT32_MAX || (int64_t)H * C >= INT32_MAX) return fail(RGBX_E_RANGE, "%s: size exceeds int32", name);
  return RGBX_OK;
}


// ------------------------------------------------------------------------------------------
// Single-head attention coefficients as a per-edge array (rgbx_gat_edge_softmax_f32): with ONE head the coefficients
// are 4 bytes per edge next to the 4 * C bytes of the row the edge gathers, so a layer whose transform can run BEHIND
// the aggregation (out_i = (sum_j alpha_ij x_j) W^T, single head: GAT's last layer, models/gat.py:21,30) is a plain
// weighted aggregation + MFMA epilogue on rgbx_fused_layer_f32 once alpha exists. The pass is a chain of dependent
// loads (rowptr -> col -> a_src) over short rows, i.e. latency-bound: 8 lanes per target row (8 rows per wave) with up
// to 8 slots per lane in registers keep 8 independent gathers per lane in flight (32 lanes per row and 2 slots per
// lane: 0.66 ms at |E'| = 62 M; this form: see DESIGN 3.4b); rows beyond 64 slots recompute their scores from the
// cache-resident score vectors. Sums: per lane in slot order, then a butterfly over the row's lanes — a fixed order.
template <bool TRAIN>
__global__ void __launch_bounds__(256)
gat_edge_softmax_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ a_src,
                        const float* __restrict__ a_dst, float slope, float* __restrict__ alpha,
                        float* __restrict__ alpha_pos, float* __restrict__ m_out, float* __restrict__ rden_out,
                        float* __restrict__ apos_out, int N, int threshold) {
  constexpr int LPR = 8;  // lanes per row
  constexpr int R = 8;    // slots per lane kept in registers (LPR * R = 64 slots of a row)
  const int sub = threadIdx.x & (LPR - 1);
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
  if (row >= N) return;  // the LPR lanes of a row leave together: the butterflies below stay inside a row's lanes
  const int start = rowptr[row], end = rowptr[row + 1];
  if (threshold > 0 && end - start > threshold) return;  // hub row: gat_edge_softmax_long_kernel owns it
  const float ad = a_dst[row];
  int src[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = start + k * LPR + sub;
    src[k] = p < end ? col[p] : -1;
  }
  float er[R];
  float mx = kNegBig;
  unsigned posbits = 0;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    er[k] = kNegBig;
    if (src[k] >= 0) {
      const float s = a_src[src[k]] + ad;
      if (s > 0.f) posbits |= 1u << k;
      er[k] = s > 0.f ? s : slope * s;
    }
    mx = fmaxf(mx, er[k]);
  }
  for (int p = start + R * LPR + sub; p < end; p += LPR) {
    const float s = a_src[col[p]] + ad;
    mx = fmaxf(mx, s > 0.f ? s : slope * s);
  }
#pragma unroll
  for (int off = LPR >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float l = 0.f, lp = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    er[k] = src[k] >= 0 ? expf(er[k] - mx) : 0.f;
    l += er[k];
    if (TRAIN && (posbits >> k & 1u)) lp += er[k];
  }
  for (int p = start + R * LPR + sub; p < end; p += LPR) {
    const float s = a_src[col[p]] + ad;
    const float ex = expf((s > 0.f ? s : slope * s) - mx);
    l += ex;
    if (TRAIN && s > 0.f) lp += ex;
  }
#pragma unroll
  for (int off = LPR >> 1; off > 0; off >>= 1) {
    l += __shfl_xor(l, off);
    if (TRAIN) lp += __shfl_xor(lp, off);
  }
  const float rd = l > 0.f ? 1.0f / (l + 1e-16f) : 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = start + k * LPR + sub;
    if (src[k] >= 0) {
      const float a = er[k] * rd;
      alpha[p] = a;
      if (TRAIN) alpha_pos[p] = (posbits >> k & 1u) ? a : 0.f;
    }
  }
  for (int p = start + R * LPR + sub; p < end; p += LPR) {
    const float s = a_src[col[p]] + ad;
    const float a = expf((s > 0.f ? s : slope * s) - mx) * rd;
    alpha[p] = a;
    if (TRAIN) alpha_pos[p] = s > 0.f ? a : 0.f;
  }
  if (sub == 0) {
    m_out[row] = l > 0.f ? mx : 0.f;
    rden_out[row] = rd;
    if (TRAIN) apos_out[row] = lp * rd;
  }
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_gat_scores_f32(const float* hfeat, int64_t ldh, const float* att_src,
                                   const float* att_dst, float* a_src, float* a_dst, int64_t n, int H,
                                   int C, rgbx_stream_t stream) {
  if (int rc = check_common(n, H, C, "gat_scores")) return rc;
  if (n == 0) return RGBX_OK;
  if (!hfeat || !att_src || !att_dst || !a_src || !a_dst) return fail(RGBX_E_ARG, "gat_scores: null pointer");
  if (ldh < (int64_t)H * C) return fail(RGBX_E_ARG, "gat_scores: leading dimension < H*C");
  const int vec = pick_vec(C, {hfeat, att_src, att_dst}, {ldh});
  GatLayout L;
  if (int rc = make_layout(H, C, vec, &L, "gat_scores")) return rc;
  hipStream_t s = (hipStream_t)stream;
  int64_t b = cdiv(n, 4 * (kWave / L.G));
  const int grid = (int)(b < kMaxGrid ? b : kMaxGrid);
#define RGBX_GAT_SC(V) gat_scores_kernel<V><<<grid, 256, 0, s>>>(hfeat, ldh, att_src, att_dst, a_src, a_dst, (int)n, L)
  if (vec == 4) RGBX_GAT_SC(4);
  else if (vec == 2) RGBX_GAT_SC(2);
  else RGBX_GAT_SC(1);
#undef RGBX_GAT_SC
  RGBX_CHECK_LAUNCH("gat_scores_kernel");
  return RGBX_OK;
}

namespace {
int scores_bwd_grid(int64_t n, const GatLayout& L) {
  int64_t b = cdiv(n, 4 * (kWave / L.G) * 8);  // >= 8 row steps per wave so that partial records stay few
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}
}  // namespace

extern "C" int rgbx_gat_scores_bwd_scratch_floats(int64_t n, int H, int C, int64_t* count) {
  if (!count) return fail(RGBX_E_ARG, "gat_scores_bwd_scratch_floats: null pointer");
  if (int rc = check_common(n, H, C, "gat_scores_bwd")) return rc;
  *count = (int64_t)1024 * 2 * H * C;  // upper bound over every layout's grid
  return RGBX_OK;
}

extern "C" int rgbx_gat_scores_bwd_f32(const float* hfeat, int64_t ldh, const float* g_a_src,
                                       const float* g_a_dst, int64_t n_dst, const float* att_src,
                                       const float* att_dst, float* g_hfeat, int64_t ldgh, float* g_att_src,
                                       float* g_att_dst, float* scratch, int64_t scratch_floats, int64_t n, int H,
                                       int C, rgbx_stream_t stream) {
  if (int rc = check_common(n, H, C, "gat_scores_bwd")) return rc;
  if (!g_att_src || !g_att_dst) return fail(RGBX_E_ARG, "gat_scores_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int F = H * C;
  if (n == 0) {
    RGBX_HIP(hipMemsetAsync(g_att_src, 0, F * sizeof(float), s));
    RGBX_HIP(hipMemsetAsync(g_att_dst, 0, F * sizeof(float), s));
    return RGBX_OK;
  }
  if (!hfeat || !g_a_src || !g_a_dst || !att_src || !att_dst || !scratch)
    return fail(RGBX_E_ARG, "gat_scores_bwd: null pointer");
  if (n_dst < 0 || n_dst > n) return fail(RGBX_E_ARG, "gat_scores_bwd: n_dst outside [0, n]");
  if (ldh < F || (g_hfeat && ldgh < F)) return fail(RGBX_E_ARG, "gat_scores_bwd: leading dimension < H*C");
  const int vec = pick_vec(C, {hfeat, g_hfeat, att_src, att_dst}, {ldh, ldgh});
  GatLayout L;
  if (int rc = make_layout(H, C, vec, &L, "gat_scores_bwd")) return rc;
  const int grid = scores_bwd_grid(n, L);
  if (scratch_floats < (int64_t)grid * 2 * F)
    return fail(RGBX_E_WS, "gat_scores_bwd: scratch %lld < %lld floats", (long long)scratch_floats,
                (long long)grid * 2 * F);
  const size_t lds = (size_t)4 * 2 * L.G * vec * sizeof(float);
#define RGBX_GAT_SB(V)                                                                                       \
  gat_scores_bwd_kernel<V><<<grid, 256, lds, s>>>(hfeat, ldh, g_a_src, g_a_dst, (int)n_dst, att_src, att_dst, \
                                                  g_hfeat, ldgh, scratch, (int)n, L)
  if (vec == 4) RGBX_GAT_SB(4);
  else if (vec == 2) RGBX_GAT_SB(2);
  else RGBX_GAT_SB(1);
#undef RGBX_GAT_SB
  RGBX_CHECK_LAUNCH("gat_scores_bwd_kernel");
  gat_scores_bwd_finish_kernel<<<(int)cdiv(2 * F, 32), 256, 0, s>>>(scratch, grid, F, g_att_src, g_att_dst);
  RGBX_CHECK_LAUNCH("gat_scores_bwd_finish_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_gat_aggregate_fwd_f32(const int32_t* rowptr, const int32_t* col, const float* hfeat,
                                          int64_t ldh, const float* a_src, const float* att_src,
                                          float* a_dst, const float* att_dst, const float* out_scale,
                                          const float* bias,
                                          float* out, int64_t ldo, float* m, float* rden, float* out_pos,
                                          float* a_pos, int64_t N, int H, int C,
                                          float slope, const rgbx_row_split_t* split, rgbx_stream_t stream) {
  if (int rc = check_common(N, H, C, "gat_fwd")) return rc;
  if (N == 0) return RGBX_OK;
  if (!rowptr || !col || !hfeat || (!a_src && !att_src) || (!a_dst && !att_dst) || !out || !m || !rden)
    return fail(RGBX_E_ARG, "gat_fwd: null pointer");
  if ((out_pos != nullptr) != (a_pos != nullptr)) return fail(RGBX_E_ARG, "gat_fwd: out_pos and a_pos go together");
  if (ldh < (int64_t)H * C || ldo < (int64_t)H * C) return fail(RGBX_E_ARG, "gat_fwd: leading dimension < H*C");
  SplitDev sd;
  if (int rc = split_view(split, H, C, out_pos != nullptr, &sd, "gat_fwd")) return rc;
  const int vec = pick_vec(C, {hfeat, out, att_src, att_dst, bias, out_scale, sd.pacc, out_pos, sd.pacc2}, {ldh, ldo});
  GatLayout L;
  if (int rc = make_layout(H, C, vec, &L, "gat_fwd")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int grid = gat_grid(N);
#define RGBX_GAT_FWD(V, T)                                                                                      \
  do {                                                                                                          \
    gat_fwd_kernel<V, false, T><<<grid, 256, 0, s>>>(rowptr, col, hfeat, ldh, a_src, att_src, a_dst, att_dst,  \
                                                     out_scale, bias, out, ldo, m, rden, out_pos, a_pos,        \
                                                     (int)N, slope, L, sd);                                     \
    if (sd.threshold > 0) {                                                                                     \
      gat_fwd_kernel<V, true, T><<<gat_grid(split->n_chunks), 256, 0, s>>>(                                     \
          rowptr, col, hfeat, ldh, a_src, att_src, a_dst, att_dst, out_scale, bias, out, ldo, m, rden, out_pos, \
          a_pos, split->n_chunks, slope, L, sd);                                                                \
      gat_fwd_combine_kernel<V><<<gat_grid(split->n_long), 256, 0, s>>>(                                        \
          split->n_long, split->long_row, split->long_chunk_ptr, out_scale, bias, out, ldo, m, rden, out_pos,   \
          a_pos, L, sd);                                                                                        \
    }                                                                                                           \
  } while (0)
  if (out_pos) {
    if (vec == 4) RGBX_GAT_FWD(4, true);
    else if (vec == 2) RGBX_GAT_FWD(2, true);
    else RGBX_GAT_FWD(1, true);
  } else {
    if (vec == 4) RGBX_GAT_FWD(4, false);
    else if (vec == 2) RGBX_GAT_FWD(2, false);
    else RGBX_GAT_FWD(1, false);
  }
#undef RGBX_GAT_FWD
  RGBX_CHECK_LAUNCH("gat_fwd_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_gat_bwd_dst_f32(const int32_t* rowptr, const int32_t* col, const float* hfeat,
                                    int64_t ldh, const float* a_src, const float* a_dst, const float* m,
                                    const float* rden, const float* out, int64_t ldo, const float* gout,
                                    int64_t ldg, float* nodeq, float* g_a_dst, int64_t N, int H, int C,
                                    float slope, rgbx_stream_t stream) {
  if (int rc = check_common(N, H, C, "gat_bwd_dst")) return rc;
  if (N == 0) return RGBX_OK;
  if (!rowptr || !col || !hfeat || !a_src || !a_dst || !m || !rden || !out || !gout || !nodeq || !g_a_dst)
    return fail(RGBX_E_ARG, "gat_bwd_dst: null pointer");
  const int64_t F = (int64_t)H * C;
  if (ldh < F || ldo < F || ldg < F) return fail(RGBX_E_ARG, "gat_bwd_dst: leading dimension < H*C");
  if (!aligned16(nodeq)) return fail(RGBX_E_ALIGN, "gat_bwd_dst: nodeq must be 16-byte aligned");
  const int vec = pick_vec(C, {hfeat, out, gout}, {ldh, ldo, ldg});
  GatLayout L;
  if (int rc = make_layout(H, C, vec, &L, "gat_bwd_dst")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int grid = gat_grid(N);
#define RGBX_GAT_BD(V)                                                                              \
  gat_bwd_dst_kernel<V><<<grid, 256, 0, s>>>(rowptr, col, hfeat, ldh, a_src, a_dst, m, rden, out, ldo, \
                                             gout, ldg, reinterpret_cast<float4*>(nodeq), g_a_dst, (int)N, slope, L)
  if (vec == 4) RGBX_GAT_BD(4);
  else if (vec == 2) RGBX_GAT_BD(2);
  else RGBX_GAT_BD(1);
#undef RGBX_GAT_BD
  RGBX_CHECK_LAUNCH("gat_bwd_dst_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_gat_bwd_prep_f32(const float* a_dst, const float* m, const float* rden, const float* out,
                                     int64_t ldo, const float* bias, const float* gout, int64_t ldg, float* nodeq,
                                     const float* out_pos, const float* a_pos, float slope, float* g_a_dst,
                                     int64_t N, int H, int C, rgbx_stream_t stream) {
  if (int rc = check_common(N, H, C, "gat_bwd_prep")) return rc;
  if (N == 0) return RGBX_OK;
  if (!a_dst || !m || !rden || !out || !gout || !nodeq) return fail(RGBX_E_ARG, "gat_bwd_prep: null pointer");
  if ((out_pos != nullptr) != (a_pos != nullptr) || (out_pos != nullptr) != (g_a_dst != nullptr))
    return fail(RGBX_E_ARG, "gat_bwd_prep: out_pos, a_pos and g_a_dst go together");
  const int64_t F = (int64_t)H * C;
  if (ldo < F || ldg < F) return fail(RGBX_E_ARG, "gat_bwd_prep: leading dimension < H*C");
  if (!aligned16(nodeq)) return fail(RGBX_E_ALIGN, "gat_bwd_prep: nodeq must be 16-byte aligned");
  const int vec = pick_vec(C, {out, gout, bias, out_pos}, {ldo, ldg});
  GatLayout L;
  if (int rc = make_layout(H, C, vec, &L, "gat_bwd_prep")) return rc;
  hipStream_t s = (hipStream_t)stream;
  int64_t b = cdiv(N, 4 * (kWave / L.G));
  const int grid = (int)(b < kMaxGrid ? b : kMaxGrid);
#define RGBX_GAT_BP(V)                                                                                             \
  gat_bwd_prep_kernel<V><<<grid, 256, 0, s>>>(a_dst, m, rden, out, ldo, bias, gout, ldg,                           \
                                              reinterpret_cast<float4*>(nodeq), out_pos, a_pos, 1.0f - slope, g_a_dst, \
                                              (int)N, L)
  if (vec == 4) RGBX_GAT_BP(4);
  else if (vec == 2) RGBX_GAT_BP(2);
  else RGBX_GAT_BP(1);
#undef RGBX_GAT_BP
  RGBX_CHECK_LAUNCH("gat_bwd_prep_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_gat_bwd_src_f32(const int32_t* rowptr_t, const int32_t* col_t, const float* hfeat,
                                    int64_t ldh, const float* a_src, const float* nodeq, const float* gout,
                                    int64_t ldg, float* g_hfeat, int64_t ldgh, float* g_a_src, float* ds,
                                    const float* att2, const float* g_a_dst, int64_t N, int H, int C,
                                    float slope, const rgbx_row_split_t* split, rgbx_stream_t stream) {
  if (int rc = check_common(N, H, C, "gat_bwd_src")) return rc;
  if (N == 0) return RGBX_OK;
  if (!rowptr_t || !col_t || !hfeat || (!a_src && !att2) || !nodeq || !gout || !g_hfeat || !g_a_src)
    return fail(RGBX_E_ARG, "gat_bwd_src: null pointer");
  if (g_a_dst && !att2) return fail(RGBX_E_ARG, "gat_bwd_src: folding the score gradients needs att2");
  if (!aligned16(nodeq)) return fail(RGBX_E_ALIGN, "gat_bwd_src: nodeq must be 16-byte aligned");
  const int64_t F = (int64_t)H * C;
  if (ldh < F || ldg < F || ldgh < F) return fail(RGBX_E_ARG, "gat_bwd_src: leading dimension < H*C");
  SplitDev sd;
  if (int rc = split_view(split, H, C, false, &sd, "gat_bwd_src")) return rc;
  const int vec = pick_vec(C, {hfeat, gout, g_hfeat, sd.pacc, att2}, {ldh, ldg, ldgh});
  GatLayout L;
  if (int rc = make_layout(H, C, vec, &L, "gat_bwd_src")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int grid = gat_grid(N);
  // 32-bit element offsets into gout / nodeq when they cannot wrap (targets are among the N sources: col_t[] < N)
  const bool idx32 = (uint64_t)N * (uint64_t)(ldg > H ? ldg : H) < (1ull << 32);
#define RGBX_GAT_BS(V)                                                                                          \
  do {                                                                                                          \
    if (idx32)                                                                                                  \
      gat_bwd_src_kernel<V, false, uint32_t><<<grid, 256, 0, s>>>(                                              \
          rowptr_t, col_t, hfeat, ldh, a_src, reinterpret_cast<const float4*>(nodeq), gout, ldg, g_hfeat, ldgh, \
          g_a_src, ds, att2, g_a_dst, (int)N, slope, L, sd);                                                    \
    else                                                                                                        \
      gat_bwd_src_kernel<V, false, int64_t><<<grid, 256, 0, s>>>(                                               \
          rowptr_t, col_t, hfeat, ldh, a_src, reinterpret_cast<const float4*>(nodeq), gout, ldg, g_hfeat, ldgh, \
          g_a_src, ds, att2, g_a_dst, (int)N, slope, L, sd);                                                    \
    if (sd.threshold > 0) {                                                                                     \
      gat_bwd_src_kernel<V, true, int64_t><<<gat_grid(split->n_chunks), 256, 0, s>>>(                           \
          rowptr_t, col_t, hfeat, ldh, a_src, reinterpret_cast<const float4*>(nodeq), gout, ldg, g_hfeat, ldgh, \
          g_a_src, ds, att2, g_a_dst, split->n_chunks, slope, L, sd);                                           \
      gat_bwd_src_combine_kernel<V><<<gat_grid(split->n_long), 256, 0, s>>>(                                    \
          split->n_long, split->long_row, split->long_chunk_ptr, g_hfeat, ldgh, g_a_src, att2, g_a_dst, L, sd); \
    }                                                                                                           \
  } while (0)
  if (vec == 4) RGBX_GAT_BS(4);
  else if (vec == 2) RGBX_GAT_BS(2);
  else RGBX_GAT_BS(1);
#undef RGBX_GAT_BS
  RGBX_CHECK_LAUNCH("gat_bwd_src_kernel");
  return RGBX_OK;
}

namespace {
// Hub rows (more than split->threshold slots): one 256-thread workgroup per row instead of 8 lanes; three sweeps over
// the row (max, sum, store) with the scores recomputed from the cache-resident vectors; block reductions over a fixed
// tree, so the result does not depend on scheduling.
template <bool TRAIN>
__global__ void __launch_bounds__(256)
gat_edge_softmax_long_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                             const float* __restrict__ a_src, const float* __restrict__ a_dst, float slope,
                             float* __restrict__ alpha, float* __restrict__ alpha_pos, float* __restrict__ m_out,
                             float* __restrict__ rden_out, float* __restrict__ apos_out,
                             const int* __restrict__ long_row) {
  __shared__ float sh[2][256];
  const int row = long_row[blockIdx.x];
  const int start = rowptr[row], end = rowptr[row + 1];
  const float ad = a_dst[row];
  const int t = threadIdx.x;
  float mx = rgbx::kNegBig;
  for (int p = start + t; p < end; p += 256) {
    const float s = a_src[col[p]] + ad;
    mx = fmaxf(mx, s > 0.f ? s : slope * s);
  }
  sh[0][t] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) sh[0][t] = fmaxf(sh[0][t], sh[0][t + w]);
    __syncthreads();
  }
  mx = sh[0][0];
  __syncthreads();
  float l = 0.f, lp = 0.f;
  for (int p = start + t; p < end; p += 256) {
    const float s = a_src[col[p]] + ad;
    const float ex = expf((s > 0.f ? s : slope * s) - mx);
    l += ex;
    if (TRAIN && s > 0.f) lp += ex;
  }
  sh[0][t] = l;
  sh[1][t] = lp;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) {
      sh[0][t] += sh[0][t + w];
      sh[1][t] += sh[1][t + w];
    }
    __syncthreads();
  }
  l = sh[0][0];
  lp = sh[1][0];
  const float rd = l > 0.f ? 1.0f / (l + 1e-16f) : 0.f;
  for (int p = start + t; p < end; p += 256) {
    const float s = a_src[col[p]] + ad;
    const float a = expf((s > 0.f ? s : slope * s) - mx) * rd;
    alpha[p] = a;
    if (TRAIN) alpha_pos[p] = s > 0.f ? a : 0.f;
  }
  if (t == 0) {
    m_out[row] = l > 0.f ? mx : 0.f;
    rden_out[row] = rd;
    if (TRAIN) apos_out[row] = lp * rd;
  }
}
}  // namespace

extern "C" int rgbx_gat_edge_softmax_f32(const int32_t* rowptr, const int32_t* col, const float* a_src,
                                         const float* a_dst, float slope, float* alpha, float* alpha_pos, float* m,
                                         float* rden, float* a_pos, int64_t N, const rgbx_row_split_t* split,
                                         rgbx_stream_t stream) {
  if (N < 0) return fail(RGBX_E_ARG, "gat_edge_softmax: bad size");
  if (N == 0) return RGBX_OK;
  if (!rowptr || !col || !a_src || !a_dst || !alpha || !m || !rden)
    return fail(RGBX_E_ARG, "gat_edge_softmax: null pointer");
  if ((alpha_pos != nullptr) != (a_pos != nullptr))
    return fail(RGBX_E_ARG, "gat_edge_softmax: alpha_pos and a_pos go together");
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "gat_edge_softmax: N exceeds int32");
  hipStream_t s = (hipStream_t)stream;
  const int64_t blocks = cdiv(N, 32);  // 8 lanes per row: 32 rows per 256-thread workgroup
  const bool hubs = split && split->threshold > 0 && split->n_long > 0;
  const int threshold = hubs ? split->threshold : 0;
  if (hubs && !split->long_row) return fail(RGBX_E_ARG, "gat_edge_softmax: split without long_row");
  if (alpha_pos)
    gat_edge_softmax_kernel<true><<<(unsigned)blocks, 256, 0, s>>>(rowptr, col, a_src, a_dst, slope, alpha, alpha_pos, m,
                                                                   rden, a_pos, (int)N, threshold);
  else
    gat_edge_softmax_kernel<false><<<(unsigned)blocks, 256, 0, s>>>(rowptr, col, a_src, a_dst, slope, alpha, nullptr, m,
                                                                    rden, nullptr, (int)N, threshold);
  RGBX_CHECK_LAUNCH("gat_edge_softmax_kernel");
  if (hubs) {
    if (alpha_pos)
      gat_edge_softmax_long_kernel<true><<<split->n_long, 256, 0, s>>>(rowptr, col, a_src, a_dst, slope, alpha, alpha_pos,
                                                                       m, rden, a_pos, split->long_row);
    else
      gat_edge_softmax_long_kernel<false><<<split->n_long, 256, 0, s>>>(rowptr, col, a_src, a_dst, slope, alpha, nullptr,
                                                                        m, rden, nullptr, split->long_row);
    RGBX_CHECK_LAUNCH("gat_edge_softmax_long_kernel");
  }
  return RGBX_OK;
}
