// Row gather for SHORT rows over a SMALL table: out[i,:] = sum_p w[p] * x[col[p],:] (+ bias), one lane GROUP per target row.
//
// Where it runs: the product of a bag-of-words feature matrix with a Linear's weight over the features' non-zeros
// (ops.SparseRows: x W^T for reference models' first layer, itexperiments.py:296 row-normalised counts, Cora: 18 non-zeros of
// F = 1433 per row) — a CSR whose rows have tens of slots and whose gathered table is W^T [F, out], a few hundred KB that
// live in every XCD's L2. rgbx_spmm_csr_f32 gives such a row a whole wave: its 64 / G lane groups split 18 slots, fold
// their partial sums with shuffles, and 2 M waves are launched for 2 M rows — with the table in L2 the launch is bound by
// wave issue, not by bytes (0.81 ms at N = 2 M, 36 M non-zeros, d = 64). Here a wave takes 64 / G rows, every group walks
// its own row's slots in order (the group's lanes read the same col / w entry: one broadcast load) and stores it: no
// cross-lane traffic, a quarter of the waves at d = 64. The sum of a row is taken in slot order (deterministic; a different
// order than rgbx_spmm_csr_f32's group-interleaved one, so the two agree to rounding, not bit for bit).
// Not for the graph aggregation itself: with a table in HBM the gather is bound by the bytes in flight per CU, where
// one row per wave wins (multi-row forms lost at d <= 32 in round 2).
#include "rgbx_common.h"

namespace rgbx {
namespace {

struct ShortArgs {
  const int* rowptr;
  const int* col;
  const float* w;
  const float* x;
  const float* bias;
  float* out;
  int64_t ldx, ldo;
  int N, d;
};

template <int G, bool HAS_W>
__global__ void __launch_bounds__(256) spmm_short_rows_kernel(const ShortArgs A) {
  constexpr int VEC = 4;
  constexpr int NG = kWave / G;
  constexpr int U = 4;  // slots in flight per group
  const int lane = threadIdx.x & 63;
  const int g = lane / G, t = lane % G;
  const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NG + g;
  const int c = t * VEC;
  if (row >= A.N || c >= A.d) return;
  const int start = A.rowptr[row], end = A.rowptr[row + 1];
  float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
  const float* xc = A.x + c;
  for (int p = start; p < end; p += U) {
    float v[U][VEC];
    float ww[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = p + u < end;
      const int src = ok ? A.col[p + u] : 0;
      ww[u] = ok ? (HAS_W ? A.w[p + u] : 1.f) : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
      if (ok) load_vec<VEC>(v[u], xc + (int64_t)src * A.ldx);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(ww[u], v[u][i], acc[i]);
    }
  }
  if (A.bias) {
    float bv[VEC];
    load_vec<VEC>(bv, A.bias + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += bv[i];
  }
  using f4v = __attribute__((ext_vector_type(4))) float;
  f4v o = {acc[0], acc[1], acc[2], acc[3]};
  __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(A.out + (int64_t)row * A.ldo + c));
}

template <int G>
int launch_short(const ShortArgs& A, hipStream_t s) {
  constexpr int rows_per_block = 4 * (kWave / G);
  const int64_t blocks = cdiv(A.N, rows_per_block);
  if (A.w) spmm_short_rows_kernel<G, true><<<(int)blocks, 256, 0, s>>>(A);
  else spmm_short_rows_kernel<G, false><<<(int)blocks, 256, 0, s>>>(A);
  RGBX_CHECK_LAUNCH("spmm_short_rows_kernel");
  return RGBX_OK;
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_spmm_csr_short_rows_supported(int64_t d) { return d >= 4 && d % 4 == 0 && d <= 256; }

extern "C" int rgbx_spmm_csr_short_rows_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* x,
                                            int64_t ldx, const float* bias, float* out, int64_t ldo, int64_t N, int64_t d,
                                            rgbx_stream_t stream) {
  if (N < 0 || d < 0) return fail(RGBX_E_ARG, "spmm_short_rows: negative size");
  if (N == 0 || d == 0) return RGBX_OK;
  if (!rowptr || !col || !x || !out) return fail(RGBX_E_ARG, "spmm_short_rows: null pointer");
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "spmm_short_rows: N exceeds int32");
  if (!rgbx_spmm_csr_short_rows_supported(d)) return fail(RGBX_E_ARG, "spmm_short_rows: d must be a multiple of 4, <= 256");
  if (ldx < d || ldo < d || ldx % 4 || ldo % 4) return fail(RGBX_E_ARG, "spmm_short_rows: leading dimensions (>= d, multiples of 4)");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) % 16)
    return fail(RGBX_E_ALIGN, "spmm_short_rows: x, out, bias must be 16-byte aligned");
  if (out == x) return fail(RGBX_E_ARG, "spmm_short_rows: out must not alias x");
  ShortArgs A{rowptr, col, w, x, bias, out, ldx, ldo, (int)N, (int)d};
  hipStream_t s = (hipStream_t)stream;
  if (d <= 4) return launch_short<1>(A, s);
  if (d <= 8) return launch_short<2>(A, s);
  if (d <= 16) return launch_short<4>(A, s);
  if (d <= 32) return launch_short<8>(A, s);
  if (d <= 64) return launch_short<16>(A, s);
  if (d <= 128) return launch_short<32>(A, s);
  return launch_short<64>(A, s);
}
