// Tall-skinny fp32 GEMM on the MFMA units: C[M,N] = A^T B with A [K,M], B [K,N] row-major and
// K (= number of nodes) in the millions, M, N (= feature widths) in the hundreds. This is the weight
// gradient dW = dY^T X of every dense layer on the path (reference: the nn.Linear / GCNConv.lin /
// GATConv.lin_src inside models/gcn.py:18-21, graphsage.py:46-47, gat.py:18-21, appnp_stack.py:19-20,
// differentiated by loss.backward(), itexperiments.py:439). A generic GEMM library tiles M x N only,
// which leaves 16 workgroups for a 128 x 128 output; here K is split across the whole chip.
//
// Kernel 1: grid (S, ceil(M/128), ceil(N/128)); a workgroup (4 waves) owns one 128x128 output tile over
// one K-slab, stages [32 x 128] tiles of A and B through LDS with 16-byte loads, and every wave
// accumulates a 64x64 quadrant with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain). Both operands are
// read along their contiguous dimension: lane l of a fragment holds row k0 + (l >> 5), column
// c0 + (l & 31) — no transposes anywhere. Partial tiles go to a workspace with plain stores.
// Kernel 2: sums the S partials in slab order (bitwise reproducible, no float atomics).
// Optionally the column sums of A (= the bias gradient when A = dY) are taken from the staged A tiles by the
// workgroups of the first N-tile and reduced the same way: dY is then read once for dW and db together.
//
// Round 5 — the reference's default shapes (initial_params.py:25-29: F = 1433 -> hidden 64 -> C = 7), where dW1 = dH^T X
// is [64, 1433] over K = |V| rows and was 16.9 of the 42 ms epoch at |V| = 2 M:
//  (1) the 16-byte path no longer needs M % 4 == 0 / N % 4 == 0, only 16-byte aligned ROWS (lda % 4 == 0, ldb % 4 == 0):
//      a row's last float4 may reach into its padding (columns [N, ld)), whose products land in output columns that are
//      never stored. A feature matrix with F = 1433 is kept with a row stride of 1436 floats (ops.align_rows);
//  (2) M <= 64 runs a 64 x 128 tile (one 32-row MFMA block per wave instead of two): half the MFMAs and half the A-side
//      LDS traffic of the 128 x 128 tile, which multiplied 64 rows of zeros.
#include "rgbx_common.h"

namespace rgbx {
namespace {

constexpr int BM = 128, BN = 128, KT = 32;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Scalar staging (rows that are not 16-byte aligned): [KT x W] tile, row-major in LDS; rows >= k_end and columns >= ncols
// are zero-filled
template <int W>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ src,
                                           int64_t ld, int64_t k0, int64_t k_end, int c0, int ncols) {
  const int tid = threadIdx.x;
#pragma unroll 4
  for (int p = 0; p < KT * W / 256; ++p) {
    const int idx = p * 256 + tid;
    const int r = idx / W, c = idx % W;
    const int64_t k = k0 + r;
    lds[r * W + c] = (k < k_end && c0 + c < ncols) ? src[k * ld + c0 + c] : 0.f;
  }
}

// Register-staged variant (16-byte path): fetch_tile issues the global loads of one [KT x W] tile
// into registers, put_tile writes them to LDS. Splitting the two lets the loads of tile i+1 fly while the MFMAs
// of tile i run (the plain loop waits for HBM once per 32 rows of K with nothing else to do).
// W = tile width in floats (128, or 64 for the A side of the 64-row output tile); W / 4 float4 per tile row
template <int W>
__device__ __forceinline__ void fetch_tile(float4 (&regs)[KT * (W / 4) / 256], const float* __restrict__ src, int64_t ld,
                                           int64_t k0, int64_t k_end, int c0, int ncols) {
#pragma unroll
  for (int p = 0; p < KT * (W / 4) / 256; ++p) {
    const int idx = p * 256 + threadIdx.x;
    const int r = idx / (W / 4), c = (idx % (W / 4)) * 4;
    const int64_t k = k0 + r;
    regs[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    // c0 + c < ncols <= ld: the float4 may end in the row's padding (columns [ncols, ld)), never beyond the row
    if (k < k_end && c0 + c < ncols) regs[p] = *reinterpret_cast<const float4*>(src + k * ld + c0 + c);
  }
}

template <int W>
__device__ __forceinline__ void put_tile(float* __restrict__ lds, const float4 (&regs)[KT * (W / 4) / 256]) {
#pragma unroll
  for (int p = 0; p < KT * (W / 4) / 256; ++p) {
    const int idx = p * 256 + threadIdx.x;
    const int r = idx / (W / 4), c = (idx % (W / 4)) * 4;
    *reinterpret_cast<float4*>(lds + r * W + c) = regs[p];
  }
}

// MT = rows of the output tile a workgroup owns: 128 (each wave a 64 x 64 quadrant = 2 x 2 MFMA blocks) or 64 (each wave
// 32 x 64 = 1 x 2 blocks; M <= 64)
template <bool VEC4, int MT>
__global__ void __launch_bounds__(256)
gemm_tn_partial_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                       float* __restrict__ part, float* __restrict__ colpart, int64_t K, int M, int N,
                       int64_t slab) {
  constexpr int MI = MT / 64;  // 32-row MFMA blocks per wave along M
  __shared__ float lds[KT * (MT + 128)];
  float* la = lds;
  float* lb = lds + KT * MT;
  const int split = blockIdx.x;
  const int m0 = blockIdx.y * MT, n0 = blockIdx.z * BN;
  const int64_t k_begin = (int64_t)split * slab;
  const int64_t k_end = k_begin + slab < K ? k_begin + slab : K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * (MT / 2), wn = (wave & 1) * 64;
  const int kr = lane >> 5, cc = lane & 31;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool sums = colpart != nullptr && blockIdx.z == 0 && threadIdx.x < MT;
  float csum = 0.f;
  auto consume = [&]() {  // one staged pair of tiles: column sums of A (optional) + 16 MFMA steps
    if (sums) {  // thread t owns column m0 + t of A (rows past k_end were zero-filled)
#pragma unroll 8
      for (int r = 0; r < KT; ++r) csum += la[r * MT + threadIdx.x];
    }
#pragma unroll 4
    for (int kk = 0; kk < KT; kk += 2) {
      float a[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = la[(kk + kr) * MT + wm + i * 32 + cc];
      const float b0 = lb[(kk + kr) * 128 + wn + cc];
      const float b1 = lb[(kk + kr) * 128 + wn + 32 + cc];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1, acc[i][1], 0, 0, 0);
      }
    }
  };
  if constexpr (VEC4) {
    float4 ra[KT * (MT / 4) / 256], rb[KT * 32 / 256];
    fetch_tile<MT>(ra, A, lda, k_begin, k_end, m0, M);
    fetch_tile<128>(rb, B, ldb, k_begin, k_end, n0, N);
    for (int64_t k0 = k_begin; k0 < k_end; k0 += KT) {
      __syncthreads();  // the previous tile pair is fully consumed
      put_tile<MT>(la, ra);
      put_tile<128>(lb, rb);
      __syncthreads();
      if (k0 + KT < k_end) {  // next pair in flight while this one is multiplied
        fetch_tile<MT>(ra, A, lda, k0 + KT, k_end, m0, M);
        fetch_tile<128>(rb, B, ldb, k0 + KT, k_end, n0, N);
      }
      consume();
    }
  } else {
    for (int64_t k0 = k_begin; k0 < k_end; k0 += KT) {
      __syncthreads();
      stage_tile<MT>(la, A, lda, k0, k_end, m0, M);
      stage_tile<128>(lb, B, ldb, k0, k_end, n0, N);
      __syncthreads();
      consume();
    }
  }

  if (sums && m0 + (int)threadIdx.x < M) colpart[(int64_t)split * M + m0 + threadIdx.x] = csum;
  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* out = part + (int64_t)split * M * N;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kr;
        const int n = n0 + wn + j * 32 + cc;
        if (m < M && n < N) out[(int64_t)m * N + n] = acc[i][j][r];
      }
}

// 32 consecutive output elements per block; 8 split-lanes per element stride the S partials, then the
// 8 sub-sums are added in lane order: the summation order is fixed, so results are reproducible.
__global__ void __launch_bounds__(256)
gemm_tn_reduce_kernel(const float* __restrict__ part, int S, int64_t MN, float* __restrict__ C, int N,
                      int64_t ldc, float alpha) {
  __shared__ float sh[8][32];
  const int j = threadIdx.x & 31, q = threadIdx.x >> 5;
  for (int64_t base = (int64_t)blockIdx.x * 32; base < MN; base += (int64_t)gridDim.x * 32) {
    const int64_t i = base + j;
    float s = 0.f;
    if (i < MN)
      for (int p = q; p < S; p += 8) s += part[(int64_t)p * MN + i];
    __syncthreads();
    sh[q][j] = s;
    __syncthreads();
    if (q == 0 && i < MN) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += sh[k][j];
      C[(i / N) * ldc + (i % N)] = alpha * t;
    }
  }
}

int tile_rows(int M) { return M <= 64 ? 64 : BM; }

int choose_splits(int64_t K, int M, int N) {
  const int64_t tiles = cdiv(M, tile_rows(M)) * cdiv(N, BN);
  // ONE round of workgroups over the chip: the kernel's registers (66 VGPRs + 64 accumulators) allow 3 workgroups per
  // CU, so 256 x 3 K-slabs; 1024 (round 3) left a second round of 256 workgroups on an otherwise idle chip:
  // 2 M x 128 x 128: 0.710 -> 0.656 ms (tools/dense_bench.py, profiles/r04_gemm_tn_variants.txt; 4 waves per SIMD by
  // __launch_bounds__ spills 7 registers and gains less)
  int64_t s = cdiv(tile_rows(M) == 64 ? 1024 : 768, tiles);  // the 64-row tile (32 accumulators) fits 4 workgroups per CU
  const int64_t max_s = cdiv(K, 16 * KT);   // at least 16 staged tiles per slab: fewer, longer slabs for small K
                                            // (K = 200 k: 391 partial tiles to reduce instead of 1024)
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_gemm_tn_workspace_bytes(int64_t K, int64_t M, int64_t N, size_t* bytes) {
  if (!bytes || K < 0 || M < 0 || N < 0) return fail(RGBX_E_ARG, "gemm_tn_workspace_bytes: bad argument");
  if (M >= INT32_MAX || N >= INT32_MAX) return fail(RGBX_E_RANGE, "gemm_tn: M or N exceeds int32");
  // S partial tiles of C, then S partial column-sum vectors of A
  *bytes = (size_t)choose_splits(K, (int)M, (int)N) * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
  return RGBX_OK;
}

extern "C" int rgbx_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                                int64_t ldc, float* a_colsum, int64_t K, int64_t M, int64_t N, float alpha,
                                void* workspace, size_t workspace_bytes, rgbx_stream_t stream) {
  if (K < 0 || M < 0 || N < 0) return fail(RGBX_E_ARG, "gemm_tn: negative size");
  if (M == 0 || N == 0) return RGBX_OK;
  if (!C || (K > 0 && (!A || !B))) return fail(RGBX_E_ARG, "gemm_tn: null pointer");
  if (M >= INT32_MAX || N >= INT32_MAX) return fail(RGBX_E_RANGE, "gemm_tn: M or N exceeds int32");
  if (lda < M || ldb < N || ldc < N) return fail(RGBX_E_ARG, "gemm_tn: leading dimension too small");
  size_t need = 0;
  if (int rc = rgbx_gemm_tn_workspace_bytes(K, M, N, &need)) return rc;
  if (!workspace || workspace_bytes < need)
    return fail(RGBX_E_WS, "gemm_tn: workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const int S = choose_splits(K, (int)M, (int)N);
  const int64_t slab = cdiv(cdiv(K > 0 ? K : 1, S), KT) * KT;
  float* part = static_cast<float*>(workspace);
  float* colpart = a_colsum ? part + (size_t)S * M * N : nullptr;
  const int mt = tile_rows((int)M);
  dim3 grid(S, (unsigned)cdiv(M, mt), (unsigned)cdiv(N, BN));
  // 16-byte path: aligned ROWS are enough — a row's last float4 may end in its padding (header: "rows are read in whole
  // 16-byte groups")
  const bool v4 = aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0;
#define RGBX_GEMM_TN(V4, MT) \
  gemm_tn_partial_kernel<V4, MT><<<grid, 256, 0, s>>>(A, lda, B, ldb, part, colpart, K, (int)M, (int)N, slab)
  if (v4) { if (mt == 64) RGBX_GEMM_TN(true, 64); else RGBX_GEMM_TN(true, 128); }
  else { if (mt == 64) RGBX_GEMM_TN(false, 64); else RGBX_GEMM_TN(false, 128); }
#undef RGBX_GEMM_TN
  RGBX_CHECK_LAUNCH("gemm_tn_partial_kernel");
  const int64_t MN = M * N;
  int64_t rb = cdiv(MN, 32);
  if (rb > kMaxGrid) rb = kMaxGrid;
  gemm_tn_reduce_kernel<<<(int)rb, 256, 0, s>>>(part, S, MN, C, (int)N, ldc, alpha);
  RGBX_CHECK_LAUNCH("gemm_tn_reduce_kernel");
  if (a_colsum) {  // the same slab-ordered reduction over the [S, M] column-sum partials (unscaled)
    gemm_tn_reduce_kernel<<<(int)cdiv(M, 32), 256, 0, s>>>(colpart, S, M, a_colsum, (int)M, M, 1.0f);
    RGBX_CHECK_LAUNCH("gemm_tn_reduce_kernel");
  }
  return RGBX_OK;
}
