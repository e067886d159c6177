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
#include "rgbx_common.h"

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
};

// G lanes per neighbour row, VEC floats per lane; columns beyond G*VEC are covered by an outer
// chunk loop (only taken for d > G*VEC, i.e. d > 256 on the float4 path).
template <int G, int VEC, bool HAS_W>
__global__ void __launch_bounds__(256) spmm_csr_kernel(const SpmmArgs A) {
  constexpr int NG = kWave / G;
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int g = lane / G;
  const int t = lane % G;
  const int wpb = blockDim.x >> 6;
  const int wave0 = blockIdx.x * wpb + (threadIdx.x >> 6);
  const int wstride = gridDim.x * wpb;

  for (int row = wave0; row < A.N; row += wstride) {
    const int start = __builtin_amdgcn_readfirstlane(A.rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
    for (int cbase = 0; cbase < A.d; cbase += G * VEC) {
      const int c = cbase + t * VEC;
      const bool active = c < A.d;
      const float* xc = A.x + c;
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

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
      // fold the NG neighbour groups together
#pragma unroll
      for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off);
      }
      if (g == 0 && active) {
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
        store_vec<VEC>(A.out + (int64_t)row * A.ldo + c, r);
      }
    }
  }
}

template <int G, int VEC>
int launch(const SpmmArgs& A, hipStream_t s) {
  constexpr int kWavesPerBlock = 4;
  int64_t blocks = cdiv(A.N, kWavesPerBlock);
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  if (A.w)
    spmm_csr_kernel<G, VEC, true><<<(int)blocks, 256, 0, s>>>(A);
  else
    spmm_csr_kernel<G, VEC, false><<<(int)blocks, 256, 0, s>>>(A);
  RGBX_CHECK_LAUNCH("spmm_csr_kernel");
  return RGBX_OK;
}

template <int VEC>
int dispatch_groups(const SpmmArgs& A, hipStream_t s) {
  const int lanes = (A.d + VEC - 1) / VEC;  // lanes needed to cover one row
  if (lanes <= 1) return launch<1, VEC>(A, s);
  if (lanes <= 2) return launch<2, VEC>(A, s);
  if (lanes <= 4) return launch<4, VEC>(A, s);
  if (lanes <= 8) return launch<8, VEC>(A, s);
  if (lanes <= 16) return launch<16, VEC>(A, s);
  if (lanes <= 32) return launch<32, VEC>(A, s);
  return launch<64, VEC>(A, s);
}

}  // namespace

int spmm_dispatch(const SpmmArgs& A, hipStream_t s) {
  auto vec_ok = [&](int v) {
    const uintptr_t mask = (uintptr_t)v * 4 - 1;
    auto okp = [&](const void* p, int64_t ld) {
      return !p || (((reinterpret_cast<uintptr_t>(p) & mask) == 0) && (ld % v == 0));
    };
    return A.d % v == 0 && okp(A.x, A.ldx) && okp(A.y, A.ldy) && okp(A.out, A.ldo) && okp(A.bias, v);
  };
  if (vec_ok(4)) return dispatch_groups<4>(A, s);
  if (vec_ok(2)) return dispatch_groups<2>(A, s);
  return dispatch_groups<1>(A, s);
}

}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w,
                                 const float* rs, const float* x, int64_t ldx, const float* y,
                                 int64_t ldy, const float* bias, float* out, int64_t ldo, int64_t N,
                                 int64_t d, float a, float b, rgbx_stream_t stream) {
  if (N < 0 || d < 0) return fail(RGBX_E_ARG, "spmm: negative size");
  if (N == 0 || d == 0) return RGBX_OK;
  if (!rowptr || !col || !x || !out) return fail(RGBX_E_ARG, "spmm: null pointer");
  if (N >= INT32_MAX || d >= INT32_MAX) return fail(RGBX_E_RANGE, "spmm: N or d exceeds int32");
  if (ldx < d || ldo < d || (y && ldy < d)) return fail(RGBX_E_ARG, "spmm: leading dimension < d");
  if (out == x) return fail(RGBX_E_ARG, "spmm: out must not alias x");
  SpmmArgs A{rowptr, col, w, rs, x, y, bias, out, ldx, ldy, ldo, (int)N, (int)d, a, b};
  return spmm_dispatch(A, (hipStream_t)stream);
}

extern "C" int rgbx_appnp_f32(const int32_t* rowptr, const int32_t* col, const float* w,
                              const float* h, int64_t ldh, float* out, float* tmp, int64_t ldo,
                              int64_t N, int64_t d, int K, float alpha, rgbx_stream_t stream) {
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
    SpmmArgs A{rowptr, col, w, nullptr, src, h, nullptr, dst, lds, ldh, ldo, (int)N, (int)d, 1.0f - alpha, alpha};
    if (int rc = spmm_dispatch(A, s)) return rc;
    src = dst;
    lds = ldo;
  }
  return RGBX_OK;
}
