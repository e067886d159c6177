// Fused aggregate-then-transform: out = (rs ⊙ Â x) Wᵀ + b in ONE kernel, for the conv layers whose
// propagate commutes with their Linear (GCNConv: Â(xWᵀ) = (Âx)Wᵀ, reference models/gcn.py:27 [PyG];
// my_SAGEConv / SAGEConv mean branch, models/graphsage.py:49-58, graphsage2.py:29). The [N, K] aggregate
// never makes a round trip through HBM before the GEMM, and the GEMM's flops (fp32 MFMA) hide under the
// gather, which is bound by random cache-line requests, not by arithmetic.
//
// A workgroup (4 waves) owns a tile of 32 destination rows. Phase 1: every wave aggregates 8 of them exactly
// like spmm_csr_kernel (G lanes x float4 per neighbour row, 8 neighbour rows in flight) and parks the sums in
// an LDS tile zt[32][K + 4] (the +4 keeps 16-byte row alignment and makes the MFMA A-fragment reads 4-way
// instead of 32-way conflicted). Phase 2: the tile times Wᵀ on v_mfma_f32_32x32x2_f32 (exact fp32), the 32-column
// output tiles going round the waves; B fragments come straight from wt = Wᵀ [K, Nout] (L2-resident, two
// contiguous 128-byte segments per wave-instruction). Optionally the aggregate is also written out
// (z_out) because the weight gradient of the training pass is dyᵀ (Âx).
#include "rgbx_common.h"

namespace rgbx {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct FusedArgs {
  const int* rowptr;
  const int* col;
  const float* w;
  const float* rs;
  const float* x;
  const float* wt;
  const float* bias;
  float* out;
  float* z_out;
  int64_t ldx, ldo, ldz;
  int N, K, Nout;
};

constexpr int TM = 32;

// KC = K when it is one of the common widths (the MFMA loop then unrolls fully), 0 = any supported K.
template <int G, bool HAS_W, int KC>
__global__ void __launch_bounds__(256) spmm_linear_kernel(const FusedArgs A) {
  constexpr int NG = kWave / G;
  constexpr int U = 4;
  extern __shared__ float zt[];  // [TM][K + 4]
  const int K = KC ? KC : A.K;
  const int ldz = K + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / G, t = lane % G;
  const int c = t * 4;
  const bool active = c < K;
  const int row_base = blockIdx.x * TM;

  // ---- phase 1: 8 rows per wave into the LDS tile
  for (int rr = 0; rr < TM / 4; ++rr) {
    const int lr = wave * (TM / 4) + rr;
    const int row = row_base + lr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < A.N) {
      const int start = __builtin_amdgcn_readfirstlane(A.rowptr[row]);
      const int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
      const float* xc = A.x + c;
      for (int base = start; base < end; base += kWave) {
        const int n = min(kWave, end - base);
        int mycol = 0;
        float myw = 0.f;
        if (lane < n) {
          mycol = A.col[base + lane];
          if constexpr (HAS_W) myw = A.w[base + lane];
        }
        for (int k = 0; k < n; k += NG * U) {
          float v[U][4];
          float ww[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = k + u * NG + g;
            const int src = __shfl(mycol, idx & 63);
            if constexpr (HAS_W) ww[u] = __shfl(myw, idx & 63);
            const bool ok = active && idx < n;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[u][i] = 0.f;
            if (ok) load_vec<4>(v[u], xc + (int64_t)src * A.ldx);
            if constexpr (HAS_W) { if (!ok) ww[u] = 0.f; }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if constexpr (HAS_W) acc[i] = fmaf(ww[u], v[u][i], acc[i]);
              else acc[i] += v[u][i];
            }
          }
        }
      }
#pragma unroll
      for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off);
      }
      if (A.rs) {
        const float s = A.rs[row];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] *= s;
      }
    }
    if (g == 0 && active) {
      store_vec<4>(&zt[lr * ldz + c], acc);
      if (A.z_out && row < A.N) store_vec<4>(A.z_out + (int64_t)row * A.ldz + c, acc);
    }
  }
  __syncthreads();

  // ---- phase 2: out[32, Nout] = zt[32, K] * wt[K, Nout] + bias; lane l holds A[row l&31][k + (l>>5)] and
  // B[k + (l>>5)][col l&31]; C/D: column l&31, row (r&3) + 8*(r>>2) + 4*(l>>5)
  const int kr = lane >> 5, cc = lane & 31;
  for (int n0 = wave * 32; n0 < A.Nout; n0 += 4 * 32) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (KC > 0) {
#pragma unroll 16
      for (int ks = 0; ks < KC; ks += 2) {
        const float a = zt[cc * ldz + ks + kr];
        const float b = A.wt[(int64_t)(ks + kr) * A.Nout + n0 + cc];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    } else {
      for (int ks = 0; ks < K; ks += 4) {  // K % 4 == 0: two MFMA steps per trip
        const float a0 = zt[cc * ldz + ks + kr], a1 = zt[cc * ldz + ks + 2 + kr];
        const float b0 = A.wt[(int64_t)(ks + kr) * A.Nout + n0 + cc];
        const float b1 = A.wt[(int64_t)(ks + 2 + kr) * A.Nout + n0 + cc];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
      }
    }
    const float bb = A.bias ? A.bias[n0 + cc] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * kr;
      if (row < A.N) A.out[(int64_t)row * A.ldo + n0 + cc] = acc[r] + bb;
    }
  }
}

template <int G, int KC>
int launch(const FusedArgs& A, hipStream_t s) {
  const int64_t blocks = cdiv(A.N, TM);
  const size_t lds = (size_t)TM * (A.K + 4) * sizeof(float);
  if (A.w)
    spmm_linear_kernel<G, true, KC><<<(int)blocks, 256, lds, s>>>(A);
  else
    spmm_linear_kernel<G, false, KC><<<(int)blocks, 256, lds, s>>>(A);
  RGBX_CHECK_LAUNCH("spmm_linear_kernel");
  return RGBX_OK;
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_spmm_linear_supported(int64_t K, int64_t Nout) {
  return K >= 4 && K % 4 == 0 && K <= 256 && Nout >= 32 && Nout % 32 == 0;
}

extern "C" int rgbx_spmm_linear_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* rs,
                                    const float* x, int64_t ldx, const float* wt, const float* bias, float* out,
                                    int64_t ldo, float* z_out, int64_t ldz, int64_t N, int64_t K, int64_t Nout,
                                    rgbx_stream_t stream) {
  if (N < 0 || K <= 0 || Nout <= 0) return fail(RGBX_E_ARG, "spmm_linear: bad size");
  if (N == 0) return RGBX_OK;
  if (!rowptr || !col || !x || !wt || !out) return fail(RGBX_E_ARG, "spmm_linear: null pointer");
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "spmm_linear: N exceeds int32");
  if (!rgbx_spmm_linear_supported(K, Nout))
    return fail(RGBX_E_SHAPE, "spmm_linear: needs K %% 4 == 0, K <= 256, Nout %% 32 == 0 (got K=%lld, Nout=%lld)",
                (long long)K, (long long)Nout);
  if (ldx < K || ldo < Nout || (z_out && ldz < K)) return fail(RGBX_E_ARG, "spmm_linear: leading dimension too small");
  if (!aligned16(x) || ldx % 4 || (z_out && (!aligned16(z_out) || ldz % 4)))
    return fail(RGBX_E_ALIGN, "spmm_linear: x / z_out must be 16-byte aligned with ld %% 4 == 0");
  FusedArgs A{rowptr, col, w, rs, x, wt, bias, out, z_out, ldx, ldo, ldz, (int)N, (int)K, (int)Nout};
  hipStream_t s = (hipStream_t)stream;
  if (K == 128) return launch<32, 128>(A, s);
  if (K == 64) return launch<16, 64>(A, s);
  if (K == 256) return launch<64, 256>(A, s);
  const int lanes = (int)(K / 4);
  if (lanes <= 1) return launch<1, 0>(A, s);
  if (lanes <= 2) return launch<2, 0>(A, s);
  if (lanes <= 4) return launch<4, 0>(A, s);
  if (lanes <= 8) return launch<8, 0>(A, s);
  if (lanes <= 16) return launch<16, 0>(A, s);
  if (lanes <= 32) return launch<32, 0>(A, s);
  return launch<64, 0>(A, s);
}
