// DAGNN's adaptive mix of the K+1 propagated hops, forward and backward, as streaming kernels for gfx950.
// Replaces reference models/dagnn.py:49-55 (Prop.forward after the K propagates):
//     pps    = stack([x, A_hat x, ..., A_hat^K x], dim=1)        [N, K+1, C]
//     retain = sigmoid(proj(pps))                                  [N, K+1]      proj = Linear(C, 1)
//     out    = (retain.unsqueeze(1) @ pps).squeeze(1)              [N, C]
// The PyG-free torch formulation materialises the stack (11 x N x C floats at K = 10), the projection, and in
// backward the stack's gradient plus one cat per hop. Here the hops stay where the SpMM wrote them ([K] matrices
// behind the layer input), one pass reads them once and writes `out`; the backward pass reads them once more and
// writes, per hop, the DIRECT part of that hop's gradient,
//     direct_k[i,:] = retain_ik * gout[i,:] + c_ik * s,      c_ik = <gout_i, hop_k[i]> * retain_ik (1 - retain_ik),
// which the transposed SpMM of the Horner chain G_k = A_hat^T G_{k+1} + direct_k then takes as its `y` operand;
// g_s = sum_ik c_ik hop_k[i,:] and g_b = sum_ik c_ik are reduced in the same pass (per-block partials in a fixed
// grid, added in block order: reproducible).
//
// Lane layout as in the SpMM: a row of d floats occupies G = pow2ceil(d / 4) lanes (float4 each), NG = 64 / G
// rows side by side per wave; the two dot products per (row, hop) cost log2(G) cross-lane adds each.
// HBM-bound: forward (K+1) N d 4 B read + N d 4 B written; backward (K+2) N d 4 B read + (K+1) N d 4 B written.
#include "rgbx_common.h"

namespace rgbx {
namespace {

constexpr int kGateGrid = 2048;  // blocks of the backward pass = partial sums kept per column

struct HopSet {       // hop 0 = the layer input, hops 1..K = K matrices `stride` floats apart
  const float* h0;
  const float* rest;
  int64_t ld0, stride, ld;
  __device__ __forceinline__ const float* row(int k, int64_t r) const {
    return k == 0 ? h0 + r * ld0 : rest + (int64_t)(k - 1) * stride + r * ld;
  }
};

struct HopOut {
  float* h0;
  float* rest;
  int64_t ld0, stride, ld;
  __device__ __forceinline__ float* row(int k, int64_t r) const {
    return k == 0 ? h0 + r * ld0 : rest + (int64_t)(k - 1) * stride + r * ld;
  }
};

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int G>
__global__ void __launch_bounds__(256)
gate_fwd_kernel(const HopSet H, const float* __restrict__ s, const float* __restrict__ b, float* __restrict__ out,
                int64_t ldo, int N, int d, int K1) {
  constexpr int NG = kWave / G;
  constexpr int UK = 4;  // hops in flight per lane
  const int lane = threadIdx.x & 63;
  const int g = lane / G, t = lane % G;
  const int c = t * 4;
  const bool active = c < d;
  float sv[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) load_vec<4>(sv, s + c);
  const float bias = b ? *b : 0.f;
  const int wpb = blockDim.x >> 6;
  const int64_t step = (int64_t)gridDim.x * wpb * NG;
  for (int64_t base = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * NG; base < N; base += step) {
    const int64_t r = base + g;
    const bool ok = active && r < N;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K1; k0 += UK) {
      float h[UK][4];
#pragma unroll
      for (int u = 0; u < UK; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h[u][i] = 0.f;
        if (ok && k0 + u < K1) load_vec<4>(h[u], H.row(k0 + u, r) + c);
      }
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        if (k0 + u < K1) {  // wave-uniform
          float dot = h[u][0] * sv[0] + h[u][1] * sv[1] + h[u][2] * sv[2] + h[u][3] * sv[3];
          dot = group_sum<G>(dot);
          const float keep = sigmoidf(dot + bias);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = fmaf(keep, h[u][i], acc[i]);
        }
      }
    }
    if (ok) store_vec<4>(out + r * ldo + c, acc);
  }
}

template <int G>
__global__ void __launch_bounds__(256)
gate_bwd_kernel(const HopSet H, const float* __restrict__ s, const float* __restrict__ b,
                const float* __restrict__ gout, int64_t ldg, const HopOut D, float* __restrict__ part, int N, int d,
                int K1) {
  constexpr int NG = kWave / G;
  constexpr int UK = 2;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / G, t = lane % G;
  const int c = t * 4;
  const bool active = c < d;
  float sv[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) load_vec<4>(sv, s + c);
  const float bias = b ? *b : 0.f;
  float gs[4] = {0.f, 0.f, 0.f, 0.f};
  float gb = 0.f;
  const int wpb = blockDim.x >> 6;
  const int64_t step = (int64_t)gridDim.x * wpb * NG;
  for (int64_t base = ((int64_t)blockIdx.x * wpb + wave) * NG; base < N; base += step) {
    const int64_t r = base + g;
    const bool ok = active && r < N;
    float go[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) load_vec<4>(go, gout + r * ldg + c);
    for (int k0 = 0; k0 < K1; k0 += UK) {
      float h[UK][4];
#pragma unroll
      for (int u = 0; u < UK; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h[u][i] = 0.f;
        if (ok && k0 + u < K1) load_vec<4>(h[u], H.row(k0 + u, r) + c);
      }
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        if (k0 + u < K1) {
          float ds = h[u][0] * sv[0] + h[u][1] * sv[1] + h[u][2] * sv[2] + h[u][3] * sv[3];
          float dg = h[u][0] * go[0] + h[u][1] * go[1] + h[u][2] * go[2] + h[u][3] * go[3];
          ds = group_sum<G>(ds);
          dg = group_sum<G>(dg);
          const float keep = sigmoidf(ds + bias);
          const float cc = dg * keep * (1.0f - keep);
          float dir[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dir[i] = fmaf(keep, go[i], cc * sv[i]);
            gs[i] = fmaf(cc, h[u][i], gs[i]);
          }
          if (ok) store_vec<4>(D.row(k0 + u, r) + c, dir);
          if (t == 0 && r < N) gb += cc;
        }
      }
    }
  }
  // block partials: groups of a wave folded with cross-lane adds, the 4 waves through LDS
#pragma unroll
  for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) gs[i] += __shfl_xor(gs[i], off);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) gb += __shfl_xor(gb, off);
  __shared__ float red[4][260];
  if (g == 0 && active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][c + i] = gs[i];
  }
  if (lane == 0) red[wave][256] = gb;
  __syncthreads();
  float* mine = part + (int64_t)blockIdx.x * (d + 1);
  for (int j = threadIdx.x; j <= d; j += blockDim.x) {
    const int slot = j < d ? j : 256;
    float v = 0.f;
    for (int w = 0; w < wpb; ++w) v += red[w][slot];
    mine[j] = v;
  }
}

// One block per column (columns 0..d-1: g_s; column d: g_b): partials added in block order, in double.
__global__ void __launch_bounds__(256)
gate_bwd_finish_kernel(const float* __restrict__ part, int blocks, int d, float* __restrict__ g_s,
                       float* __restrict__ g_b) {
  const int j = blockIdx.x;
  double v = 0.0;
  for (int p = threadIdx.x; p < blocks; p += blockDim.x) v += (double)part[(int64_t)p * (d + 1) + j];
  __shared__ double red[256];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (j < d) g_s[j] = (float)red[0];
    else if (g_b) g_b[0] = (float)red[0];
  }
}

int gate_blocks(int64_t N, int ng) {
  int64_t blocks = cdiv(N, 4 * ng);
  return (int)std::min<int64_t>(std::max<int64_t>(blocks, 1), kGateGrid);
}

int check_common(const char* what, const float* h0, int64_t ld0, const float* rest, int64_t stride, int64_t ldh,
                 const float* s, int64_t N, int64_t d, int K) {
  if (N < 0 || d < 0 || K < 0) return fail(RGBX_E_ARG, "%s: negative size", what);
  if (!h0 || !s || (K > 0 && !rest)) return fail(RGBX_E_ARG, "%s: null pointer", what);
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "%s: N exceeds int32", what);
  if (d % 4 || d > 256) return fail(RGBX_E_SHAPE, "%s: needs d %% 4 == 0 and d <= 256 (got %lld)", what, (long long)d);
  if (ld0 < d || (K > 0 && (ldh < d || stride < 0))) return fail(RGBX_E_ARG, "%s: leading dimension < d", what);
  if (!aligned16(h0) || !aligned16(rest) || !aligned16(s) || ld0 % 4 || ldh % 4 || stride % 4)
    return fail(RGBX_E_ALIGN, "%s: needs 16-byte aligned rows", what);
  return RGBX_OK;
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

#define RGBX_GATE_DISPATCH(lanes, CALL)          \
  do {                                           \
    if ((lanes) <= 1) { CALL(1); }               \
    else if ((lanes) <= 2) { CALL(2); }          \
    else if ((lanes) <= 4) { CALL(4); }          \
    else if ((lanes) <= 8) { CALL(8); }          \
    else if ((lanes) <= 16) { CALL(16); }        \
    else if ((lanes) <= 32) { CALL(32); }        \
    else { CALL(64); }                           \
  } while (0)

extern "C" int rgbx_dagnn_gate_fwd_f32(const float* h0, int64_t ld0, const float* hops, int64_t hop_stride,
                                       int64_t ldh, const float* s, const float* b, float* out, int64_t ldo,
                                       int64_t N, int64_t d, int K, rgbx_stream_t stream) {
  if (int rc = check_common("dagnn_gate_fwd", h0, ld0, hops, hop_stride, ldh, s, N, d, K)) return rc;
  if (N == 0 || d == 0) return RGBX_OK;
  if (!out) return fail(RGBX_E_ARG, "dagnn_gate_fwd: null pointer");
  if (ldo < d) return fail(RGBX_E_ARG, "dagnn_gate_fwd: leading dimension < d");
  if (!aligned16(out) || ldo % 4) return fail(RGBX_E_ALIGN, "dagnn_gate_fwd: needs 16-byte aligned rows");
  const HopSet H{h0, hops, ld0, hop_stride, ldh};
  hipStream_t st = (hipStream_t)stream;
  const int lanes = (int)d / 4;
#define CALL(GG)                                                                                              \
  gate_fwd_kernel<GG><<<(int)std::min<int64_t>(cdiv(N, 4 * (kWave / GG)), kMaxGrid), 256, 0, st>>>(H, s, b, out, ldo, (int)N, \
                                                                                                   (int)d, K + 1)
  RGBX_GATE_DISPATCH(lanes, CALL);
#undef CALL
  RGBX_CHECK_LAUNCH("gate_fwd_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_dagnn_gate_bwd_workspace_bytes(int64_t d, size_t* bytes) {
  if (d < 0 || !bytes) return fail(RGBX_E_ARG, "dagnn_gate_bwd_workspace_bytes: bad argument");
  *bytes = (size_t)kGateGrid * (size_t)(d + 1) * sizeof(float);
  return RGBX_OK;
}

extern "C" int rgbx_dagnn_gate_bwd_f32(const float* h0, int64_t ld0, const float* hops, int64_t hop_stride,
                                       int64_t ldh, const float* s, const float* b, const float* gout, int64_t ldg,
                                       float* d0, int64_t ldd0, float* dk, int64_t d_stride, int64_t ldd,
                                       float* g_s, float* g_b, void* ws, size_t ws_bytes, int64_t N, int64_t d,
                                       int K, rgbx_stream_t stream) {
  if (int rc = check_common("dagnn_gate_bwd", h0, ld0, hops, hop_stride, ldh, s, N, d, K)) return rc;
  if (!gout || !d0 || (K > 0 && !dk) || !g_s || !ws) return fail(RGBX_E_ARG, "dagnn_gate_bwd: null pointer");
  if (ldg < d || ldd0 < d || (K > 0 && (ldd < d || d_stride < 0)))
    return fail(RGBX_E_ARG, "dagnn_gate_bwd: leading dimension < d");
  if (!aligned16(gout) || !aligned16(d0) || !aligned16(dk) || ldg % 4 || ldd0 % 4 || ldd % 4 || d_stride % 4)
    return fail(RGBX_E_ALIGN, "dagnn_gate_bwd: needs 16-byte aligned rows");
  size_t need = 0;
  rgbx_dagnn_gate_bwd_workspace_bytes(d, &need);
  if (ws_bytes < need) return fail(RGBX_E_WS, "dagnn_gate_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  if (N == 0 || d == 0) {
    if (d > 0) RGBX_HIP(hipMemsetAsync(g_s, 0, (size_t)d * sizeof(float), st));
    if (g_b) RGBX_HIP(hipMemsetAsync(g_b, 0, sizeof(float), st));
    return RGBX_OK;
  }
  const HopSet H{h0, hops, ld0, hop_stride, ldh};
  const HopOut D{d0, dk, ldd0, d_stride, ldd};
  float* part = static_cast<float*>(ws);
  const int lanes = (int)d / 4;
  int blocks = 1;
#define CALL(GG)                                                                                      \
  blocks = gate_blocks(N, kWave / GG);                                                                \
  gate_bwd_kernel<GG><<<blocks, 256, 0, st>>>(H, s, b, gout, ldg, D, part, (int)N, (int)d, K + 1)
  RGBX_GATE_DISPATCH(lanes, CALL);
#undef CALL
  RGBX_CHECK_LAUNCH("gate_bwd_kernel");
  gate_bwd_finish_kernel<<<(int)d + 1, 256, 0, st>>>(part, blocks, (int)d, g_s, g_b);
  RGBX_CHECK_LAUNCH("gate_bwd_finish_kernel");
  return RGBX_OK;
}
