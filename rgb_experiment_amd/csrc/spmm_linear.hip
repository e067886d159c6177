// Fused aggregate-then-transform: out = (rs ⊙ Â x) Wᵀ + b in ONE kernel, for the conv layers whose
// propagate commutes with their Linear (GCNConv: Â(xWᵀ) = (Âx)Wᵀ, reference models/gcn.py:27 [PyG];
// my_SAGEConv / SAGEConv mean branch, models/graphsage.py:49-58, graphsage2.py:29). The [N, K] aggregate
// never makes a round trip through HBM before the GEMM, and the GEMM's flops (fp32 MFMA) hide under the
// gather, which is bound by random cache-line requests, not by arithmetic.
//
// A workgroup (4 waves) owns a tile of 32 destination rows. Phase 1: the waves aggregate them one row at a time exactly
// like spmm_csr_kernel (G lanes x float4 per neighbour row, 8 neighbour rows in flight) and parks the sums in
// an LDS tile zt[32][K + 4] (the +4 keeps 16-byte row alignment and makes the MFMA A-fragment reads 4-way
// instead of 32-way conflicted). Phase 2: the tile times Wᵀ on v_mfma_f32_32x32x2_f32 (exact fp32), the 32-column
// output tiles going round the waves; B fragments come straight from wt = Wᵀ [K, Nout] (L2-resident, two
// contiguous 128-byte segments per wave-instruction). Optionally the aggregate is also written out
// (z_out) because the weight gradient of the training pass is dyᵀ (Âx).
// Root term (SAGEConv / my_SAGEConv: lin_l(mean_j x_j) + lin_r(x_i)): after the first product the workgroup
// reloads the SAME LDS tile with its 32 nodes' own rows (coalesced 16-byte loads) and accumulates
// x_root · Wrᵀ into the same MFMA accumulators — a second tile would halve the workgroups per CU, which costs
// the gather 5 % (measured), two more barriers cost nothing measurable.
#include "rgbx_common.h"
#include "spmm_internal.h"

namespace rgbx {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Output stores of the aggregating kernel are NON-TEMPORAL (global_store ... nt): the rows it writes (out, the stored
// aggregate, the loss gradient: 1-2 GB per launch at L) are read by nobody in this launch and compete in L2 / Infinity
// Cache with the gathered table, a quarter of which is served from those caches. Measured on one box, all forms gathering
// the same matrix (tools/ab_fused_forms.py, profiles/r04_ab_nt_stores.txt): plain - 0.9 %, z + stats - 2.2 %,
// pre + z + ce_grad - 1.3 %; at S (everything cache-resident) + - 0.5 %.
using f4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void nt_store4(float* p, const float (&v)[4]) {
  f4v t = {v[0], v[1], v[2], v[3]};
  __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
}
__device__ __forceinline__ void nt_store1(float* p, float v) { __builtin_nontemporal_store(v, p); }

struct FusedArgs {
  const int* rowptr;
  const int* col;
  const float* w;
  const float* rs;
  const float* x;
  const float* wt;
  const float* xr;   // root rows [N, K] (ldr) or NULL
  const float* wtr;  // Wr^T [K, Nout]
  const float* bias;
  float* out;
  float* z_out;
  int64_t ldx, ldo, ldz, ldr;
  // hub rows (more than `threshold` slots): aggregates precomputed by the split-row kernels, compact in hub order
  const int* long_row;
  const float* zlong;
  int threshold, n_long;
  int N, K, Nout;
  // optional per-column affine map of the GATHERED matrix, x' = x * pre_scale + pre_shift (a training-mode
  // BatchNorm in front of this layer, never materialised): by linearity the aggregate of x' is
  // pre_scale * aggregate(x) + pre_shift * pre_rowsum[row], pre_rowsum = the row's sum of aggregation weights
  const float* pre_scale;
  const float* pre_shift;
  const float* pre_rowsum;
  // optional: per-workgroup column sums of the stored output tile, part[block][0][c] = sum_rows out, [1][c] = sum out^2
  // (the statistics of the BatchNorm that follows the layer, taken from the MFMA accumulators instead of a pass
  // over out)
  float* stats_part;
  // optional cross-entropy epilogue (the layer is the model's last: models/gcn.py:29-31 followed by the loss of
  // itexperiments.py:429 / the metrics of :624-626): per 32-row tile the masked NLL sum, row count and arg-max hits of
  // softmax(out) go to ce_part[block][3]; `out` receives the loss gradient ce_scale * (softmax - onehot) when ce_scale
  // is set, and is not written at all otherwise. Nout <= 128 only.
  const int64_t* ce_y;
  const uint8_t* ce_mask;
  const float* ce_scale;
  double* ce_part;
  int ce_groups;  // 1: ce_mask is a boolean; 2: bit 0 / bit 1 of ce_mask[row] select the row for statistics set 0 / 1
                  // (ce_part records are then 6 wide: one forward, the statistics of two masks — val and test)
  // node-partitioned runs (rgbx_fused_layer_f32). Blocked layout: element (i, c) at base + (c / cols) * stride +
  // i * cols + c % cols — column slices stored one after the other, the form the exchange sends and receives.
  // x_bc / x_bs: layout of x in DENSE mode (0 = row-major ldx); xr_bc / xr_bs: of the root rows; out_blk: a blocked
  // copy of the output (next to `out`, or instead of it when out == NULL).
  int64_t x_bc, x_bs, xr_bc, xr_bs;
  float* out_blk;
  int64_t ob_c, ob_s;
  // POS instantiations (rgbx_fused_layer_t.w_pos): a second aggregate of the same gathered rows under a second weight
  // vector, stored next to z_out (same ldz) — never transformed
  const float* w2;
  float* zpos_out;
  const float* zlong_pos;
};

__device__ __forceinline__ const float* blocked_at(const float* base, int64_t bc, int64_t bs, int64_t ld, int row,
                                                   int c) {
  return bc ? base + (int64_t)(c / (int)bc) * bs + (int64_t)row * bc + (c % (int)bc) : base + (int64_t)row * ld + c;
}

constexpr int TM = kTileRows;  // destination rows per workgroup = one MFMA row tile; 4 waves aggregate 8 rows each
                            // (64- and 128-row tiles, one B fragment feeding 2-4 MFMAs, measured slower: DESIGN 3.2a)
constexpr int NT_ROOT = 2;  // 32-column output tiles a wave may own when a root term is present (Nout <= 256)
// Registers: the row-per-wave gather needs ~47 VGPRs and is bound by how many waves keep loads in flight, so this
// kernel must not fall below the plain SpMM's 8 waves per SIMD: VGPRs + AGPRs <= 64 (`__launch_bounds__(.., 8)`);
// the first build (84 registers, 5 waves per SIMD) lost 8 % to occupancy alone.

// acc += zt[32, K] * wt[K, n0 : n0 + 32]. Lane l holds A[row l&31][k + (l>>5)] and B[k + (l>>5)][col l&31].
constexpr int KB = 16;  // MFMA steps (2 k each) whose B values are fetched as one batch of independent loads

// B values of steps [s0, s0 + KB) of the column tile n0: b[i] = wt[2 (s0 + i) + kr][n0 + cc]
__device__ __forceinline__ void load_b_batch(float (&b)[KB], const float* __restrict__ wt, int Nout, int n0,
                                             int s0, int kr, int cc) {
#pragma unroll
  for (int i = 0; i < KB; ++i) b[i] = wt[(int64_t)(2 * (s0 + i) + kr) * Nout + n0 + cc];
}

// (The 4-byte A reads at a row stride of K + 4 floats are 4-way bank-conflicted. Round 4 tried the permuted tile of
// dense_stream_kernel here — one ds_read_b128 per four MFMAs: the float4 in flight next to 2 x KB B values pushed the
// 64-VGPR instantiations into 3-8 spills and the launches got SLOWER, plain + 2.4 %, loss statistics + 12 %
// (profiles/r04_ab_fused_forms_permuted_tile.txt): this kernel's MFMA phase hides under the other workgroups' gathers,
// its register budget does not.)
__device__ __forceinline__ void mfma_batch(f32x16& acc, const float (&b)[KB], const float* __restrict__ zt, int ldz,
                                           int s0, int kr, int cc) {
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    const float a = zt[cc * ldz + 2 * (s0 + i) + kr];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i], acc, 0, 0, 0);
  }
}

// KC > 0 (K % 64 == 0): the B values come in batches of KB independent loads, so the L2 latency is paid once per
// batch instead of once per MFMA (the compiler's own schedule waited on almost every load); `first` may hold
// batch 0, fetched before the barrier that guards zt.
template <int KC>
__device__ __forceinline__ void tile_times_wt(f32x16& acc, const float* __restrict__ zt, int ldz,
                                              const float* __restrict__ wt, int K, int Nout, int n0, int kr,
                                              int cc, const float (*first)[KB] = nullptr) {
  if constexpr (KC > 0) {
    static_assert(KC % (2 * KB) == 0, "KC must be a multiple of 2 * KB");
    int s0 = 0;
    if (first) {
      mfma_batch(acc, *first, zt, ldz, 0, kr, cc);
      s0 = KB;
    }
#pragma unroll 1
    for (; s0 < KC / 2; s0 += KB) {  // one batch in registers at a time (unrolled, the scheduler hoists them all)
      float b[KB];
      load_b_batch(b, wt, Nout, n0, s0, kr, cc);
      mfma_batch(acc, b, zt, ldz, s0, kr, cc);
    }
  } else {
    for (int ks = 0; ks < K; ks += 4) {  // K % 4 == 0: two MFMA steps per trip
      const float b0 = wt[(int64_t)(ks + kr) * Nout + n0 + cc];
      const float b1 = wt[(int64_t)(ks + 2 + kr) * Nout + n0 + cc];
      const float a0 = zt[cc * ldz + ks + kr], a1 = zt[cc * ldz + ks + 2 + kr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
    }
  }
}

// C/D layout of the 32x32 MFMA: column l&31, row (r&3) + 8*(r>>2) + 4*(l>>5)
template <bool BLK>
__device__ __forceinline__ void store_tile(const f32x16& acc, const float* __restrict__ bias,
                                           float* __restrict__ out, int64_t ldo, int row_base, int N, int n0,
                                           int kr, int cc, float* __restrict__ stats_part = nullptr, int Nout = 0,
                                           float* __restrict__ out_blk = nullptr, int64_t ob_c = 0, int64_t ob_s = 0,
                                           int tile = -1) {
  if (tile < 0) tile = blockIdx.x;  // one tile per workgroup unless the caller walks several
  const float bb = bias ? bias[n0 + cc] : 0.f;
  float s1 = 0.f;
  // blocked copy: this lane's column n0 + cc sits in block (n0 + cc) / ob_c at offset (n0 + cc) % ob_c
  float* ob = BLK && out_blk ? out_blk + (int64_t)((n0 + cc) / (int)ob_c) * ob_s + ((n0 + cc) % (int)ob_c) : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * kr;
    if (row < N) {
      const float v = acc[r] + bb;
      if (!BLK || out) nt_store1(&out[(int64_t)row * ldo + n0 + cc], v);
      if constexpr (BLK) { if (ob) ob[(int64_t)row * ob_c] = v; }
      s1 += v;
    }
  }
  if (stats_part) {
    // One record per tile and column: (sum, sum of squared deviations from the TILE's mean) — the tile's values are all
    // in registers, so the second moment is taken around their own mean. A record of (sum x, sum x^2) in fp32 loses the
    // variance of a nearly constant column (|mean| >> std: x^2 rounds at 6e-8 mean^2, BatchNorm divides by
    // sqrt(var + 1e-5)); the reducers restore sum x^2 = M2 + S^2 / n per tile in fp64 (tile_stats_gather_kernel).
    // Lanes l and l + 32 hold the two row halves of column cc.
    s1 += __shfl_xor(s1, 32);
    const int n_t = N - row_base < kTileRows ? N - row_base : kTileRows;
    const float mean = s1 / (float)n_t;
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * kr;
      if (row < N) {
        const float dv = (acc[r] + bb) - mean;
        m2 = fmaf(dv, dv, m2);
      }
    }
    m2 += __shfl_xor(m2, 32);
    if (kr == 0) {
      float* rec = stats_part + (int64_t)tile * 2 * Nout;
      rec[n0 + cc] = s1;
      rec[Nout + n0 + cc] = m2;
    }
  }
}

// Second stage of the output statistics: workgroup g adds the per-tile records g, g + G, ... in fp64 (thread = column of
// the [2, Nout] record), a third stage adds the G sums in order: fixed summation order, reproducible. A tile's record is
// (S = sum x, M2 = sum (x - S/n)^2) over its n rows (kTileRows, the last tile the rest of n_rows); what is added up is
// (sum x, sum x^2) with sum x^2 = M2 + S^2 / n formed in fp64 from the SAME rounded S — so the variance the consumer takes
// from the totals is the within-tile part exactly plus the spread of the tile means, whatever the column's mean.
__device__ __forceinline__ double tile_stat(const float* __restrict__ part, int b, int width2, int c, int n_rows) {
  const int half = width2 >> 1;
  const double v = (double)part[(int64_t)b * width2 + c];
  if (c < half) return v;
  const double S = (double)part[(int64_t)b * width2 + c - half];
  const int left = n_rows - b * kTileRows;
  return v + S * S / (double)(left < kTileRows ? left : kTileRows);
}

__global__ void __launch_bounds__(256)
tile_stats_gather_kernel(const float* __restrict__ part, int n_tiles, int width2, double* __restrict__ part2, int n_rows) {
  const int G = gridDim.x;
  for (int c = threadIdx.x; c < width2; c += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains: the loads of a trip are all in flight
    int b = blockIdx.x;
    for (; b + 3 * G < n_tiles; b += 4 * G) {
      s0 += tile_stat(part, b, width2, c, n_rows);
      s1 += tile_stat(part, b + G, width2, c, n_rows);
      s2 += tile_stat(part, b + 2 * G, width2, c, n_rows);
      s3 += tile_stat(part, b + 3 * G, width2, c, n_rows);
    }
    for (; b < n_tiles; b += G) s0 += tile_stat(part, b, width2, c, n_rows);
    part2[(int64_t)blockIdx.x * width2 + c] = (s0 + s1) + (s2 + s3);
  }
}

__global__ void __launch_bounds__(256)
tile_stats_finish_kernel(const double* __restrict__ part2, int G, int width2, double* __restrict__ sums) {
  __shared__ double sh[32][8];
  const int j = threadIdx.x & 7, q = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + j;
  double s = 0.0;
  if (c < width2)
    for (int b = q; b < G; b += 32) s += part2[(int64_t)b * width2 + c];
  sh[q][j] = s;
  __syncthreads();
  if (q == 0 && c < width2) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][j];
    sums[c] = t;
  }
}


// Cross-entropy of the finished 32 x Nout tile (Nout <= 128), see FusedArgs::ce_part. The four waves park their 32 x 32
// blocks in LDS (the aggregate tile is dead by now), then each wave takes 8 rows: a lane holds columns lane and
// lane + 64, max / first arg-max / sum-exp go through the wave with shuffles. NLLLoss(log_softmax(z))_i = lse_i - z[i, y_i]
// as rgbx_masked_ce_fwd_f32 computes it; the gradient as rgbx_masked_ce_bwd_f32.
__device__ __forceinline__ void ce_epilogue(const FusedArgs& A, const f32x16& acc, float* __restrict__ ot, int row_base,
                                            int wave, int lane, int tile = -1) {
  if (tile < 0) tile = blockIdx.x;
  const int kr = lane >> 5, cc = lane & 31;
  const int ldq = A.Nout + 4;
  const int n0 = wave * 32;
  __syncthreads();  // every wave is done reading the aggregate (or root) tile
  if (n0 < A.Nout) {
    const float bb = A.bias ? A.bias[n0 + cc] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2) + 4 * kr) * ldq + n0 + cc] = acc[r] + bb;
  }
  __syncthreads();
  double nll = 0.0, nll2 = 0.0;  // nll2 / cnt2 / hit2: the second statistics set (ce_groups == 2)
  int cnt = 0, hit = 0, cnt2 = 0, hit2 = 0;  // row counts as integers: one register each, exact
  const int W = 3 * A.ce_groups;
  const float sc = A.ce_scale ? A.ce_scale[0] : 0.f;
  const bool c0 = lane < A.Nout, c1 = lane + 64 < A.Nout;
  for (int rr = 0; rr < TM / 4; ++rr) {
    const int rl = wave * (TM / 4) + rr;
    const int row = row_base + rl;
    if (row >= A.N) break;  // wave-uniform
    int t = -1;
    int bits = A.ce_mask ? (int)A.ce_mask[row] : 1;
    bits = A.ce_groups == 2 ? (bits & 3) : (bits ? 1 : 0);
    if (bits) {
      const int64_t ti = A.ce_y[row];
      if (ti >= 0 && ti < A.Nout) t = (int)ti;
    }
    if (t < 0 && !A.ce_scale) continue;  // not selected, nothing to store: wave-uniform
    const float v0 = c0 ? ot[rl * ldq + lane] : -INFINITY;
    const float v1 = c1 ? ot[rl * ldq + lane + 64] : -INFINITY;
    float best = v0;
    int arg = c0 ? lane : INT32_MAX;
    if (v1 > best) { best = v1; arg = lane + 64; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oa = __shfl_xor(arg, off);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    float se = (c0 ? expf(v0 - best) : 0.f) + (c1 ? expf(v1 - best) : 0.f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off);
    const float lse = best + logf(se);
    if (t >= 0) {
      const double term = (double)(lse - ot[rl * ldq + t]);
      const int h = arg == t ? 1 : 0;
      if (bits & 1) {
        nll += term;
        cnt += 1;
        hit += h;
      }
      if (bits & 2) {
        nll2 += term;
        cnt2 += 1;
        hit2 += h;
      }
    }
    if (A.ce_scale) {
      float* orow = A.out + (int64_t)row * A.ldo;
      if (c0) nt_store1(&orow[lane], t >= 0 ? sc * (expf(v0 - lse) - (lane == t ? 1.f : 0.f)) : 0.f);
      if (c1) nt_store1(&orow[lane + 64], t >= 0 ? sc * (expf(v1 - lse) - (lane + 64 == t ? 1.f : 0.f)) : 0.f);
    }
  }
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
  if ((int)threadIdx.x < W) {
    const int k = threadIdx.x;
    A.ce_part[(int64_t)tile * W + k] = (cew[0][k] + cew[1][k]) + (cew[2][k] + cew[3][k]);
  }
}

// stats[k] = sum over the tiles' records in a fixed order, two stages: kCeGather workgroups add a contiguous slice of
// the records each (thread-strided, then a block tree), one workgroup adds their sums.

__device__ __forceinline__ double block_tree_sum(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256)
ce_tiles_gather_kernel(const double* __restrict__ part, int n_tiles, double* __restrict__ part2, int W) {
  __shared__ double sh[256];
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int b0 = blockIdx.x * per, b1 = min(n_tiles, b0 + per);
  for (int k = 0; k < W; ++k) {
    double v = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += 256) v += part[(int64_t)b * W + k];
    v = block_tree_sum(v, sh);
    if (threadIdx.x == 0) part2[blockIdx.x * W + k] = v;
  }
}

__global__ void __launch_bounds__(256)
ce_tiles_finish_kernel(const double* __restrict__ part2, int n, double* __restrict__ stats, int W) {
  __shared__ double sh[256];
  for (int k = 0; k < W; ++k) {
    const double v = block_tree_sum((int)threadIdx.x < n ? part2[threadIdx.x * W + k] : 0.0, sh);
    if (threadIdx.x == 0) stats[k] = v;
  }
}

// KC = K when it is one of the common widths (the MFMA loop then unrolls fully), 0 = any supported K.
// CE: instantiated with the cross-entropy epilogue (NT == 1 only); a separate instantiation so that the epilogue's
// registers are not the plain kernel's problem (as a run-time branch it cost the hot kernel a spill).
// DENSE: no aggregation — the tile's rows are rows of A.x (plain or blocked), mapped by the pre-affine if given: the
// return stage of the partitioned run's exchange, or a plain x * wt product, with the same phase 2 and epilogues.
// BLK: the blocked layouts of FusedArgs are honoured (partitioned runs); the single-GPU instantiations are compiled
// without them, so their register budget (64 VGPRs, no SGPR spills in the gather loop) is what it was.
// POS: every gathered row is also accumulated under the second weight vector A.w2 into A.zpos_out (single-head GAT's
// training forward: the part of the aggregate carried by edges with a positive score, see gat.hip). One more
// accumulator per lane: 7 waves per SIMD instead of 8.
template <int G, bool HAS_W, int KC, int NT, bool CE = false, bool DENSE = false, bool BLK = false, bool POS = false>
__global__ void __launch_bounds__(256, POS ? 7 : (NT == 2 ? 5 : 8)) spmm_linear_kernel(const FusedArgs A) {
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

  // ---- phase 1: the 32 rows of the tile, handed to the waves one at a time from an LDS counter — a fixed 8 rows per
  // wave leaves three waves waiting at the barrier for the one with the longest rows (5.10 -> 5.07 ms at L)
  __shared__ int next_row;
  if constexpr (DENSE) {
    const int k4 = K >> 2;
    for (int idx = threadIdx.x; idx < TM * k4; idx += 256) {
      const int r = idx / k4, c4 = (idx - r * k4) * 4;
      const int row = row_base + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (row < A.N) {
        load_vec<4>(v, blocked_at(A.x, BLK ? A.x_bc : 0, A.x_bs, A.ldx, row, c4));
        if (A.pre_scale) {
          float ps[4], pt[4];
          load_vec<4>(ps, A.pre_scale + c4);
          load_vec<4>(pt, A.pre_shift + c4);
          const float rsum = A.pre_rowsum[row];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], ps[i], pt[i] * rsum);
        }
        if (A.z_out) store_vec<4>(A.z_out + (int64_t)row * A.ldz + c4, v);
      }
      store_vec<4>(&zt[r * ldz + c4], v);
    }
  }
  if (!DENSE && threadIdx.x == 0) next_row = 0;
  if constexpr (!DENSE) __syncthreads();
  while (!DENSE) {
    int lr = 0;
    if (lane == 0) lr = atomicAdd(&next_row, 1);
    lr = __builtin_amdgcn_readfirstlane(lr);
    if (lr >= TM) break;
    const int row = row_base + lr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float accp[4] = {0.f, 0.f, 0.f, 0.f};  // POS only
    if (row < A.N) {
      const int start = __builtin_amdgcn_readfirstlane(A.rowptr[row]);
      const int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
      if (A.threshold > 0 && end - start > A.threshold) {
        // hub row (wave-uniform branch): its aggregate was finished by the split-row kernels; find the row in
        // the sorted hub list and copy
        int lo = 0, hi = A.n_long - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (A.long_row[mid] < row) lo = mid + 1;
          else hi = mid;
        }
        if (g == 0 && active) {
          load_vec<4>(acc, A.zlong + (int64_t)lo * K + c);
          if constexpr (POS) load_vec<4>(accp, A.zlong_pos + (int64_t)lo * K + c);
        }
      } else {
        const float* xc = A.x + c;
        for (int base = start; base < end; base += kWave) {
          const int n = min(kWave, end - base);
          int mycol = 0;
          float myw = 0.f, mywp = 0.f;
          if (lane < n) {
            mycol = A.col[base + lane];
            if constexpr (HAS_W) myw = A.w[base + lane];
            if constexpr (POS) mywp = A.w2[base + lane];
          }
          for (int k = 0; k < n; k += NG * U) {
            float v[U][4];
            float ww[U];
            float wp[POS ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int idx = k + u * NG + g;
              const int src = __shfl(mycol, idx & 63);
              if constexpr (HAS_W) ww[u] = __shfl(myw, idx & 63);
              if constexpr (POS) wp[u] = __shfl(mywp, idx & 63);
              const bool ok = active && idx < n;
#pragma unroll
              for (int i = 0; i < 4; ++i) v[u][i] = 0.f;
              if (ok) load_vec<4>(v[u], xc + (int64_t)src * A.ldx);
              if constexpr (HAS_W) { if (!ok) ww[u] = 0.f; }
              if constexpr (POS) { if (!ok) wp[u] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if constexpr (HAS_W) acc[i] = fmaf(ww[u], v[u][i], acc[i]);
                else acc[i] += v[u][i];
                if constexpr (POS) accp[i] = fmaf(wp[u], v[u][i], accp[i]);
              }
            }
          }
        }
#pragma unroll
        for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off);
          if constexpr (POS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accp[i] += __shfl_xor(accp[i], off);
          }
        }
        if (A.rs) {
          const float s = A.rs[row];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] *= s;
        }
      }
    }
    if (g == 0 && active) {
      if (A.pre_scale && row < A.N) {
        float ps[4], pt[4];
        load_vec<4>(ps, A.pre_scale + c);
        load_vec<4>(pt, A.pre_shift + c);
        const float rsum = A.pre_rowsum[row];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(acc[i], ps[i], pt[i] * rsum);
      }
      store_vec<4>(&zt[lr * ldz + c], acc);
      if (A.z_out && row < A.N) nt_store4(A.z_out + (int64_t)row * A.ldz + c, acc);
      if constexpr (POS) {
        if (row < A.N) store_vec<4>(A.zpos_out + (int64_t)row * A.ldz + c, accp);
      }
    }
  }
  // the first batch of W^T values of this wave's first column tile does not depend on the tile: fetch it now, so its
  // L2 latency passes while the workgroup's slower waves finish their rows
  const int kr = lane >> 5, cc = lane & 31;
  float bpre[KB];
  const bool pre = KC > 0 && NT > 0 && wave < 4 && wave * 32 < A.Nout;
  if constexpr (KC > 0) {
    if (pre) load_b_batch(bpre, A.wt, A.Nout, wave * 32, 0, kr, cc);
  }
  // likewise the workgroup's own rows for the root term (K <= 128: at most 4 float4 per thread): their HBM latency
  // passes under the barrier and the first product instead of between two barriers
  constexpr bool kRootRegs = KC > 0 && KC <= 128 && NT > 0;
  constexpr int kRootVecs = kRootRegs ? (TM * (KC / 4)) / 256 : 1;
  float rootv[kRootVecs][4];
  if constexpr (kRootRegs) {
    if (A.xr) {
#pragma unroll
      for (int j = 0; j < kRootVecs; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int r = idx / (KC / 4), c4 = (idx - r * (KC / 4)) * 4;
        const int row = row_base + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) rootv[j][i] = 0.f;
        if (row < A.N) {
          load_vec<4>(rootv[j], blocked_at(A.xr, BLK ? A.xr_bc : 0, A.xr_bs, A.ldr, row, c4));
          if (A.pre_scale) {  // the root rows are rows of the same affinely mapped matrix
            float ps[4], pt[4];
            load_vec<4>(ps, A.pre_scale + c4);
            load_vec<4>(pt, A.pre_shift + c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) rootv[j][i] = fmaf(rootv[j][i], ps[i], pt[i]);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: out[TM, Nout] = zt[TM, K] * wt[K, Nout] (+ xroot[TM, K] * wtr[K, Nout]) + bias. Waves 0..3
  // own the 32-column tiles
  if (NT == 0) {  // Nout > 256 (no root term): column tiles one after the other
    if (wave >= 4) return;
    for (int n0 = wave * 32; n0 < A.Nout; n0 += 4 * 32) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      tile_times_wt<KC>(acc, zt, ldz, A.wt, K, A.Nout, n0, kr, cc);
      store_tile<BLK>(acc, A.bias, A.out, A.ldo, row_base, A.N, n0, kr, cc, A.stats_part, A.Nout, A.out_blk, A.ob_c, A.ob_s);
    }
    return;
  }
  constexpr int NTT = NT > 0 ? NT : 1;
  f32x16 acc[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
    const int n0 = wave * 32 + tt * 128;
    if (wave < 4 && n0 < A.Nout)
      tile_times_wt<KC>(acc[tt], zt, ldz, A.wt, K, A.Nout, n0, kr, cc, pre && tt == 0 ? &bpre : nullptr);
  }
  if (!A.xr) {  // no root term (uniform): store and leave
    if constexpr (CE) {
      ce_epilogue(A, acc[0], zt, row_base, wave, lane);
      return;
    }
    if (wave >= 4) return;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int n0 = wave * 32 + tt * 128;
      if (n0 < A.Nout) store_tile<BLK>(acc[tt], A.bias, A.out, A.ldo, row_base, A.N, n0, kr, cc, A.stats_part, A.Nout, A.out_blk,
                                  A.ob_c, A.ob_s);
    }
    return;
  }
  __syncthreads();  // every wave is done reading the aggregate tile
  if constexpr (kRootRegs) {
#pragma unroll
    for (int j = 0; j < kRootVecs; ++j) {
      const int idx = threadIdx.x + j * 256;
      const int r = idx / (KC / 4), c4 = (idx - r * (KC / 4)) * 4;
      store_vec<4>(&zt[r * ldz + c4], rootv[j]);
    }
  } else {
    const int k4 = K >> 2;
    for (int idx = threadIdx.x; idx < TM * k4; idx += 256) {
      const int r = idx / k4, c4 = (idx - r * k4) * 4;
      const int row = row_base + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (row < A.N) {
        load_vec<4>(v, blocked_at(A.xr, BLK ? A.xr_bc : 0, A.xr_bs, A.ldr, row, c4));
        if (A.pre_scale) {
          float ps[4], pt[4];
          load_vec<4>(ps, A.pre_scale + c4);
          load_vec<4>(pt, A.pre_shift + c4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], ps[i], pt[i]);
        }
      }
      store_vec<4>(&zt[r * ldz + c4], v);
    }
  }
  __syncthreads();
  if constexpr (CE) {
    if (wave * 32 < A.Nout) tile_times_wt<KC>(acc[0], zt, ldz, A.wtr, K, A.Nout, wave * 32, kr, cc);
    ce_epilogue(A, acc[0], zt, row_base, wave, lane);
    return;
  }
  if (wave >= 4) return;
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int n0 = wave * 32 + tt * 128;
    if (n0 < A.Nout) {
      tile_times_wt<KC>(acc[tt], zt, ldz, A.wtr, K, A.Nout, n0, kr, cc);
      store_tile<BLK>(acc[tt], A.bias, A.out, A.ldo, row_base, A.N, n0, kr, cc, A.stats_part, A.Nout, A.out_blk,
                                  A.ob_c, A.ob_s);
    }
  }
}

// LDS tile layout of dense_stream_kernel (round 4). A lane's A values of MFMA steps 4 j .. 4 j + 3 are
// k = 8 j + kr + {0, 2, 4, 6}: every OTHER float of its tile row. Read one by one that is a 4-byte ds_read per MFMA at a
// row stride of K + 4 floats, 4-way bank-conflicted (round 3: SQ_LDS_BANK_CONFLICT 12.0 M of 17.0 M LDS cycles). The
// tile is therefore stored with the columns of every group of 8 permuted to [k0 k2 k4 k6 | k1 k3 k5 k7]: the four values
// are then four CONSECUTIVE floats at 8 j + 4 kr — one conflict-free ds_read_b128 per four MFMAs (16 lanes x 4 banks
// cover the 64 banks once per lane group). Writers hold four consecutive k (a float4 of a row) and store them as two
// float2 (park4). K order, B operands and therefore every result bit are those of spmm_linear_kernel<.., DENSE>.
// v = columns c4 .. c4 + 3 (c4 % 4 == 0) of a tile row -> their permuted places
// (Which of the two 8-byte stores goes first alternates with bit 5 of c4: the 16 lanes of a ds_write_b64 lane group hold
// c4 = 0, 4, .., 60; written in the same order, lanes 8..15 would hit the banks of lanes 0..7 again — the 2-way
// conflicts that were 8.0 M of this kernel's 32.5 M LDS cycles per 2 M-row launch.)
__device__ __forceinline__ void park4(float* __restrict__ row, int c4, const float (&v)[4]) {
  float* p = row + (c4 & ~7) + ((c4 & 4) >> 1);
  const bool swap = (c4 & 32) != 0;
  const float2 even = make_float2(v[0], v[2]), odd = make_float2(v[1], v[3]);
  *reinterpret_cast<float2*>(swap ? p + 4 : p) = swap ? odd : even;
  *reinterpret_cast<float2*>(swap ? p : p + 4) = swap ? even : odd;
}

// The DENSE form for the widths a partitioned run lives on (K = 64 / 128), as a STREAMING kernel: a workgroup walks
// many 32-row tiles and keeps its waves' W^T fragments in registers (K / 2 values per lane and 32-column tile; the
// root term's Wr^T likewise), so W is read once per workgroup instead of once per tile — the per-tile 4-byte L2
// fetches are what held the one-tile-per-workgroup form at 0.12-0.20 ms for 250 k rows against 0.05 ms of streaming.
// Same tile loads, epilogues (blocked store, column sums, loss), K order and numbers as
// spmm_linear_kernel<.., DENSE>.
//
// Round 4: (1) A fragments by ds_read_b128 from the permuted tile (one conflict-free 16-byte read per four MFMAs; the
// 4-byte reads at stride K + 4 were 4-way conflicted and one per MFMA); (2) the plain forms (no root term, no loss
// epilogue) are software-pipelined over TWO LDS tiles: the next tile's rows are fetched into registers before the
// MFMA loop of the current one and written to the other tile after it, one barrier per tile — the load of tile t + 1
// and the stores of tile t - 1 travel under tile t's 64 MFMAs instead of between two barriers. Measured (tools/dense_bench.py,
// profiles/r04_dense_bench.txt): K = Nout = 128, 2 M rows 0.94 -> 0.70 ms (94 TF = 0.60 of the fp32 MFMA peak; hipBLASLt
// 0.63), 250 k rows 0.102 -> 0.092; K = 64, 2 M rows 0.53 -> 0.34 ms. K = 128 runs 3 workgroups per CU (146 VGPRs: 64 W^T
// values + 16 for the tile in flight; at 128 VGPRs it spilled 16 and was slower than round 3), K = 64 four.
template <int KC>
__device__ __forceinline__ void stream_tile_mfma(f32x16& acc, const float* __restrict__ zt, const float (&breg)[KC / 2],
                                                 int kr, int cc) {
  constexpr int ldz = KC + 4;
#pragma unroll
  for (int j = 0; j < KC / 8; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(&zt[cc * ldz + 8 * j + 4 * kr]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[4 * j], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[4 * j + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[4 * j + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[4 * j + 3], acc, 0, 0, 0);
  }
}

template <int KC, int NT, bool CE, bool ROOT>
__global__ void __launch_bounds__(256, ROOT || NT > 1 ? 2 : (CE || KC > 64 ? 3 : 4)) dense_stream_kernel(const FusedArgs A, int tiles) {
  extern __shared__ float zt[];  // [TM][KC + 4] (the loss epilogue re-uses it as [TM][Nout + 4]); pipelined forms: two
  constexpr int ldz = KC + 4, k4 = KC / 4, S = KC / 2;
  constexpr bool PIPE = !CE && NT == 1;  // NT = 2 keeps 128 W^T values per lane: no room for a tile in flight
  constexpr int VPT = (TM * k4) / 256;  // float4 per thread and tile (KC = 128: 4, KC = 64: 2)
  static_assert((TM * k4) % 256 == 0, "tile loads are spread evenly over the workgroup");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kr = lane >> 5, cc = lane & 31;
  float breg[NT][S];
  float rreg[ROOT ? NT : 1][ROOT ? S : 1];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const int n0 = wave * 32 + tt * 128;
    if (n0 < A.Nout) {
#pragma unroll
      for (int i = 0; i < S; ++i) {
        breg[tt][i] = A.wt[(int64_t)(2 * i + kr) * A.Nout + n0 + cc];
        if constexpr (ROOT) rreg[tt][i] = A.wtr[(int64_t)(2 * i + kr) * A.Nout + n0 + cc];
      }
    }
  }
  if constexpr (PIPE) {
    // this thread's VPT float4 of a tile: element idx = threadIdx.x + u * 256 -> row idx / k4, columns 4 * (idx % k4)
    // (ROOT: the same for the workgroup's own rows x_root, parked into a second pair of tiles behind the first pair)
    float pv[VPT][4];
    float pr[ROOT ? VPT : 1][4];
    constexpr int kRootTiles = 2 * TM * ldz;  // offset of the root rows' tiles
    auto fetch = [&](int tile) {
      const int row_base = tile * TM;
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        const int idx = threadIdx.x + u * 256;
        const int r = idx / k4, c4 = (idx - r * k4) * 4;
        const int row = row_base + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[u][i] = 0.f;
        if (row < A.N) load_vec<4>(pv[u], blocked_at(A.x, A.x_bc, A.x_bs, A.ldx, row, c4));
        if constexpr (ROOT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pr[u][i] = 0.f;
          if (row < A.N) load_vec<4>(pr[u], blocked_at(A.xr, A.xr_bc, A.xr_bs, A.ldr, row, c4));
        }
      }
    };
    auto park = [&](int tile, float* buf) {  // pre-affine map, optional z_out, then into the LDS tile
      const int row_base = tile * TM;
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        const int idx = threadIdx.x + u * 256;
        const int r = idx / k4, c4 = (idx - r * k4) * 4;
        const int row = row_base + r;
        if (row < A.N) {
          if (A.pre_scale) {
            float ps[4], pt[4];
            load_vec<4>(ps, A.pre_scale + c4);
            load_vec<4>(pt, A.pre_shift + c4);
            const float rsum = A.pre_rowsum[row];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[u][i] = fmaf(pv[u][i], ps[i], pt[i] * rsum);
            if constexpr (ROOT) {  // the root rows are rows of the same affinely mapped matrix (no row-sum factor)
#pragma unroll
              for (int i = 0; i < 4; ++i) pr[u][i] = fmaf(pr[u][i], ps[i], pt[i]);
            }
          }
          if (A.z_out) store_vec<4>(A.z_out + (int64_t)row * A.ldz + c4, pv[u]);
        }
        park4(&buf[r * ldz], c4, pv[u]);
        if constexpr (ROOT) park4(&buf[kRootTiles + r * ldz], c4, pr[u]);
      }
    };
    int cur = 0;  // the tile being multiplied lives at zt + cur * TM * ldz, the one being filled at the other half
    int tile = blockIdx.x;
    if (tile < tiles) {
      fetch(tile);
      park(tile, zt);
    }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
      const int next = tile + gridDim.x;
      if (next < tiles) fetch(next);  // in flight during this tile's MFMAs
      f32x16 acc[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
        if (wave * 32 + tt * 128 < A.Nout) {
          stream_tile_mfma<KC>(acc[tt], zt + cur * (TM * ldz), breg[tt], kr, cc);
          if constexpr (ROOT) stream_tile_mfma<KC>(acc[tt], zt + kRootTiles + cur * (TM * ldz), rreg[tt], kr, cc);
        }
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int n0 = wave * 32 + tt * 128;
        if (n0 < A.Nout)
          store_tile<true>(acc[tt], A.bias, A.out, A.ldo, tile * TM, A.N, n0, kr, cc, A.stats_part, A.Nout, A.out_blk,
                           A.ob_c, A.ob_s, tile);
      }
      if (next < tiles) park(next, zt + (cur ^ 1) * (TM * ldz));
      __syncthreads();  // the other tile is complete, and every wave is done reading this one
      cur ^= 1;
    }
    return;
  }
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row_base = tile * TM;
    for (int idx = threadIdx.x; idx < TM * k4; idx += 256) {
      const int r = idx / k4, c4 = (idx - r * k4) * 4;
      const int row = row_base + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (row < A.N) {
        load_vec<4>(v, blocked_at(A.x, A.x_bc, A.x_bs, A.ldx, row, c4));
        if (A.pre_scale) {
          float ps[4], pt[4];
          load_vec<4>(ps, A.pre_scale + c4);
          load_vec<4>(pt, A.pre_shift + c4);
          const float rsum = A.pre_rowsum[row];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], ps[i], pt[i] * rsum);
        }
        if (A.z_out) store_vec<4>(A.z_out + (int64_t)row * A.ldz + c4, v);
      }
      park4(&zt[r * ldz], c4, v);
    }
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
      if (wave * 32 + tt * 128 < A.Nout) stream_tile_mfma<KC>(acc[tt], zt, breg[tt], kr, cc);
    }
    if constexpr (ROOT) {
      __syncthreads();  // every wave is done reading the loaded tile
      for (int idx = threadIdx.x; idx < TM * k4; idx += 256) {
        const int r = idx / k4, c4 = (idx - r * k4) * 4;
        const int row = row_base + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < A.N) {
          load_vec<4>(v, blocked_at(A.xr, A.xr_bc, A.xr_bs, A.ldr, row, c4));
          if (A.pre_scale) {
            float ps[4], pt[4];
            load_vec<4>(ps, A.pre_scale + c4);
            load_vec<4>(pt, A.pre_shift + c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], ps[i], pt[i]);
          }
        }
        park4(&zt[r * ldz], c4, v);
      }
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        if (wave * 32 + tt * 128 < A.Nout) stream_tile_mfma<KC>(acc[tt], zt, rreg[tt], kr, cc);
      }
    }
    if constexpr (CE) {
      ce_epilogue(A, acc[0], zt, row_base, wave, lane, tile);
    } else {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int n0 = wave * 32 + tt * 128;
        if (n0 < A.Nout)
          store_tile<true>(acc[tt], A.bias, A.out, A.ldo, row_base, A.N, n0, kr, cc, A.stats_part, A.Nout, A.out_blk,
                           A.ob_c, A.ob_s, tile);
      }
    }
    __syncthreads();  // the tile buffer is loaded again
  }
}

// true when the streaming form took the launch
template <int KC>
bool launch_dense_stream(const FusedArgs& A, hipStream_t s) {
  const int tiles = (int)cdiv(A.N, TM);
  const bool root = A.xr != nullptr;
  const int nt = A.Nout <= 128 ? 1 : 2;
  if (A.Nout > 256 || (root && nt == 2)) return false;  // register budget: the one-tile-per-workgroup form
  // the loss epilogue (barriers, LDS-pipe reductions per row) wants the 8 workgroups per CU of the one-tile form to hide
  // behind: measured at 250 k rows, K = Nout = 128: 0.144 / 0.202 ms (eval / training) there against 0.180 / 0.269 here;
  // the plain forms gain: 0.120 -> 0.100 ms
  if (A.ce_part) return false;
  const int per_cu = (root || nt > 1) ? 2 : (KC > 64 ? 3 : 4);  // = the kernel's __launch_bounds__
  // as many workgroups as give every one of them the same number of tiles (± 1 on the last few): 7,813 tiles (250 k rows) over
  // 768 workgroups would be 10 for most and 11 for some — a tenth of the launch spent with most of the chip idle
  const int slots = 256 * per_cu;
  const int grid = tiles <= slots ? tiles : (int)cdiv(tiles, cdiv(tiles, slots));
  const size_t lds2 = 2 * (size_t)TM * (KC + 4) * sizeof(float);  // the pipelined forms keep two tiles
  if (nt == 1) {
    if (root) {  // + two tiles of root rows: 4 x 32 x (KC + 4) floats = 67.6 KB at KC = 128, beyond the 64 KB a launch
      // may ask for without saying so
      static const hipError_t allow = hipFuncSetAttribute(
          reinterpret_cast<const void*>(&dense_stream_kernel<KC, 1, false, true>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds2));
      if (allow != hipSuccess) return false;  // the one-tile-per-workgroup form takes the launch
      dense_stream_kernel<KC, 1, false, true><<<grid, 256, 2 * lds2, s>>>(A, tiles);
    }
    else dense_stream_kernel<KC, 1, false, false><<<grid, 256, lds2, s>>>(A, tiles);
  } else {
    dense_stream_kernel<KC, 2, false, false><<<grid, 256, lds2, s>>>(A, tiles);
  }
  return true;
}

template <int G, int KC, bool DENSE = false>
int launch(const FusedArgs& A, hipStream_t s) {
  const int64_t blocks = cdiv(A.N, TM);
  // the cross-entropy epilogue re-uses the tile as [TM][Nout + 4]
  const size_t lds = (size_t)TM * ((A.ce_part && A.Nout > A.K ? A.Nout : A.K) + 4) * sizeof(float);
  // NT = 32-column tiles a wave keeps accumulators for (Nout <= 128: 1, <= 256: 2); 0 = any Nout, tile by tile
  const int nt = A.Nout <= 128 ? 1 : (A.Nout <= 128 * NT_ROOT ? NT_ROOT : 0);
  // blocked layouts in play (a partitioned run): the BLK instantiations; DENSE is always one of them
  const bool blk = DENSE || A.out_blk || A.xr_bc;
#define RGBX_FUSED(HW, NTV)                                                                      \
  do {                                                                                           \
    if (blk) spmm_linear_kernel<G, HW, KC, NTV, false, DENSE, true><<<(int)blocks, 256, lds, s>>>(A); \
    else spmm_linear_kernel<G, HW, KC, NTV, false, DENSE, DENSE><<<(int)blocks, 256, lds, s>>>(A);   \
  } while (0)
  if constexpr (!DENSE && KC > 0) {
    if (A.w2) {  // the entry point checked: w set, Nout <= 128, no blocked layouts
      if (A.ce_part) spmm_linear_kernel<G, true, KC, 1, true, false, false, true><<<(int)blocks, 256, lds, s>>>(A);
      else spmm_linear_kernel<G, true, KC, 1, false, false, false, true><<<(int)blocks, 256, lds, s>>>(A);
      RGBX_CHECK_LAUNCH("spmm_linear_kernel (second aggregate)");
      return RGBX_OK;
    }
  }
  if (A.ce_part) {  // nt == 1: the entry point checked Nout <= 128; the loss epilogue has no blocked output
    if (A.w && !DENSE) spmm_linear_kernel<G, !DENSE, KC, 1, true, DENSE, DENSE><<<(int)blocks, 256, lds, s>>>(A);
    else if (blk) spmm_linear_kernel<G, false, KC, 1, true, DENSE, true><<<(int)blocks, 256, lds, s>>>(A);
    else spmm_linear_kernel<G, false, KC, 1, true, DENSE, DENSE><<<(int)blocks, 256, lds, s>>>(A);
    RGBX_CHECK_LAUNCH("spmm_linear_kernel (cross-entropy epilogue)");
    return RGBX_OK;
  }
  if (A.w && !DENSE) {
    if (nt == 0) RGBX_FUSED(!DENSE, 0);
    else if (nt == 1) RGBX_FUSED(!DENSE, 1);
    else RGBX_FUSED(!DENSE, 2);
  } else {
    if (nt == 0) RGBX_FUSED(false, 0);
    else if (nt == 1) RGBX_FUSED(false, 1);
    else RGBX_FUSED(false, 2);
  }
#undef RGBX_FUSED
  RGBX_CHECK_LAUNCH("spmm_linear_kernel");
  return RGBX_OK;
}

}  // namespace

int reduce_ce_tiles(double* scratch, int tiles, double* stats, int W, hipStream_t s) {
  double* part2 = scratch + (size_t)tiles * W;  // [kCeGather, W] behind the tile records
  ce_tiles_gather_kernel<<<kCeGather, 256, 0, s>>>(scratch, tiles, part2, W);
  RGBX_CHECK_LAUNCH("ce_tiles_gather_kernel");
  ce_tiles_finish_kernel<<<1, 256, 0, s>>>(part2, kCeGather, stats, W);
  RGBX_CHECK_LAUNCH("ce_tiles_finish_kernel");
  return RGBX_OK;
}

int reduce_tile_stats(const float* part, int tiles, int width2, double* part2, double* sums, int64_t n_rows, hipStream_t s) {
  const int G = tiles < kStatsGather ? tiles : kStatsGather;
  tile_stats_gather_kernel<<<G, 256, 0, s>>>(part, tiles, width2, part2, (int)n_rows);
  RGBX_CHECK_LAUNCH("tile_stats_gather_kernel");
  tile_stats_finish_kernel<<<(int)cdiv(width2, 8), 256, 0, s>>>(part2, G, width2, sums);
  RGBX_CHECK_LAUNCH("tile_stats_finish_kernel");
  return RGBX_OK;
}

}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_spmm_linear_supported(int64_t K, int64_t Nout, int has_root) {
  return K >= 4 && K % 4 == 0 && K <= 256 && Nout >= 32 && Nout % 32 == 0 && (!has_root || Nout <= 128 * NT_ROOT);
}

extern "C" int rgbx_spmm_linear_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* rs,
                                    const float* x, int64_t ldx, const float* wt, const float* x_root, int64_t ldr,
                                    const float* wt_root, const float* bias, float* out, int64_t ldo, float* z_out,
                                    int64_t ldz, const float* pre_scale, const float* pre_shift,
                                    const float* pre_rowsum, double* out_colsums, void* stats_ws,
                                    size_t stats_ws_bytes, const rgbx_ce_epilogue_t* ce, int64_t N, int64_t K,
                                    int64_t Nout, const rgbx_row_split_t* split, rgbx_stream_t stream) {
  if (!rowptr) return fail(RGBX_E_ARG, "spmm_linear: null pointer");
  rgbx_fused_layer_t L{};
  L.rowptr = rowptr; L.col = col; L.w = w; L.rs = rs; L.x = x; L.ldx = ldx; L.wt = wt;
  L.x_root = x_root; L.ldr = ldr; L.wt_root = wt_root; L.bias = bias; L.out = out; L.ldo = ldo;
  L.z_out = z_out; L.ldz = ldz; L.pre_scale = pre_scale; L.pre_shift = pre_shift; L.pre_rowsum = pre_rowsum;
  L.out_colsums = out_colsums; L.stats_ws = stats_ws; L.stats_ws_bytes = stats_ws_bytes; L.ce = ce;
  L.N = N; L.K = K; L.Nout = Nout; L.split = split;
  return rgbx_fused_layer_f32(&L, stream);
}

extern "C" int rgbx_fused_layer_f32(const rgbx_fused_layer_t* Lp, rgbx_stream_t stream) {
  if (!Lp) return fail(RGBX_E_ARG, "fused_layer: null pointer");
  const rgbx_fused_layer_t& L = *Lp;
  const int64_t N = L.N, K = L.K, Nout = L.Nout;
  const rgbx_ce_epilogue_t* ce = L.ce;
  const bool dense = L.rowptr == nullptr;
  if (N < 0 || K <= 0 || Nout <= 0) return fail(RGBX_E_ARG, "spmm_linear: bad size");
  if (N == 0) return RGBX_OK;
  const bool stats_only = ce && !ce->grad_scale;  // the output matrix is then not written at all
  if ((!dense && !L.col) || !L.x || !L.wt || (!L.out && !L.out_blk && !stats_only))
    return fail(RGBX_E_ARG, "spmm_linear: null pointer");
  if (ce) {
    if (!ce->y || !ce->stats || !ce->scratch) return fail(RGBX_E_ARG, "spmm_linear: incomplete cross-entropy epilogue");
    if (Nout > 128) return fail(RGBX_E_SHAPE, "spmm_linear: the cross-entropy epilogue needs Nout <= 128 (got %lld)", (long long)Nout);
    if (L.out_colsums) return fail(RGBX_E_ARG, "spmm_linear: out_colsums and the cross-entropy epilogue exclude each other");
    if (L.out_blk) return fail(RGBX_E_ARG, "spmm_linear: a blocked output and the cross-entropy epilogue exclude each other");
    if (ce->grad_scale && !L.out) return fail(RGBX_E_ARG, "spmm_linear: the loss gradient needs `out`");
    if (ce->mask_groups < 0 || ce->mask_groups > 2)
      return fail(RGBX_E_ARG, "spmm_linear: mask_groups must be 0, 1 or 2 (got %d)", (int)ce->mask_groups);
    if (ce->mask_groups == 2 && (ce->grad_scale || !ce->mask))
      return fail(RGBX_E_ARG, "spmm_linear: two statistics sets (mask_groups == 2) need a mask and no loss gradient");
  }
  if ((L.w_pos != nullptr) != (L.z_pos_out != nullptr))
    return fail(RGBX_E_ARG, "fused_layer: w_pos and z_pos_out go together");
  if (L.w_pos) {
    if (dense || !L.w || L.rs || L.pre_scale || L.out_blk || L.z_out == nullptr)
      return fail(RGBX_E_ARG, "fused_layer: w_pos needs an aggregating launch with w and z_out, without rs / pre_* / out_blk");
    if ((K != 64 && K != 128 && K != 256) || Nout > 128)
      return fail(RGBX_E_SHAPE, "fused_layer: w_pos needs K in {64, 128, 256} and Nout <= 128 (got K=%lld, Nout=%lld)",
                  (long long)K, (long long)Nout);
    if (!aligned16(L.z_pos_out)) return fail(RGBX_E_ALIGN, "fused_layer: z_pos_out must be 16-byte aligned");
  }
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "spmm_linear: N exceeds int32");
  if ((L.x_root != nullptr) != (L.wt_root != nullptr))
    return fail(RGBX_E_ARG, "spmm_linear: x_root and wt_root go together");
  if ((L.pre_scale != nullptr) != (L.pre_shift != nullptr) || (L.pre_scale != nullptr) != (L.pre_rowsum != nullptr))
    return fail(RGBX_E_ARG, "spmm_linear: pre_scale, pre_shift and pre_rowsum go together");
  if (L.pre_scale && (!aligned16(L.pre_scale) || !aligned16(L.pre_shift)))
    return fail(RGBX_E_ALIGN, "spmm_linear: pre_scale / pre_shift must be 16-byte aligned");
  if (!rgbx_spmm_linear_supported(K, Nout, L.x_root != nullptr))
    return fail(RGBX_E_SHAPE,
                "spmm_linear: needs K %% 4 == 0, K <= 256, Nout %% 32 == 0, Nout <= 256 with a root term (got K=%lld, "
                "Nout=%lld)", (long long)K, (long long)Nout);
  const bool x_blk = dense && L.x_blk_cols > 0, xr_blk = L.x_root && L.xr_blk_cols > 0;
  if (!dense && L.x_blk_cols) return fail(RGBX_E_ARG, "fused_layer: a blocked x needs DENSE mode (rowptr == NULL)");
  // an AGGREGATING launch with per-slot weights and the loss epilogue, or with a second aggregate, has no instantiation
  // that honours blocked root rows (launch(): the CE / POS branches compile without BLK): refuse instead of reading the
  // blocked buffer as row-major
  if (!dense && xr_blk && ((ce && L.w) || L.w_pos))
    return fail(RGBX_E_ARG, "fused_layer: blocked root rows (xr_blk_cols) cannot be combined with per-slot weights plus "
                            "the cross-entropy epilogue, or with w_pos, on an aggregating launch");
  if ((x_blk && (L.x_blk_cols % 4 || L.x_blk_stride % 4 || K % L.x_blk_cols)) ||
      (xr_blk && (L.xr_blk_cols % 4 || L.xr_blk_stride % 4 || K % L.xr_blk_cols)) ||
      (L.out_blk && (L.ob_cols <= 0 || Nout % L.ob_cols)))
    return fail(RGBX_E_ARG, "fused_layer: block widths must divide the matrix width (inputs: multiples of 4)");
  if ((!x_blk && L.ldx < K) || (L.out && L.ldo < Nout) || (L.z_out && L.ldz < K) || (L.x_root && !xr_blk && L.ldr < K))
    return fail(RGBX_E_ARG, "spmm_linear: leading dimension too small");
  if (!aligned16(L.x) || (!x_blk && L.ldx % 4) || (L.z_out && (!aligned16(L.z_out) || L.ldz % 4)) ||
      (L.x_root && (!aligned16(L.x_root) || (!xr_blk && L.ldr % 4))))
    return fail(RGBX_E_ALIGN, "spmm_linear: x / x_root / z_out must be 16-byte aligned with ld %% 4 == 0");
  float* stats_part = nullptr;
  double* stats_part2 = nullptr;
  if (L.out_colsums) {
    size_t need = 0;
    rgbx_spmm_linear_stats_workspace_bytes(N, Nout, &need);
    if (!L.stats_ws || L.stats_ws_bytes < need)
      return fail(RGBX_E_WS, "spmm_linear: statistics workspace %zu < %zu bytes", L.stats_ws_bytes, need);
    if (reinterpret_cast<uintptr_t>(L.stats_ws) % 8) return fail(RGBX_E_ALIGN, "spmm_linear: stats_ws must be 8-byte aligned");
    stats_part2 = static_cast<double*>(L.stats_ws);                            // [kStatsGather, 2 * Nout] doubles
    stats_part = reinterpret_cast<float*>(stats_part2 + (size_t)kStatsGather * 2 * Nout);  // [tiles, 2 * Nout] floats
  }
  hipStream_t s = (hipStream_t)stream;
  const int* long_row = nullptr;
  const float* zlong = nullptr;
  const float* zlong_pos = nullptr;
  int threshold = 0, n_long = 0;
  const rgbx_row_split_t* split = dense ? nullptr : L.split;
  if (split && split->threshold > 0 && split->n_chunks > 0) {
    // hub rows first: chunk sums + ordered combine into the tail of the caller's scratch, [n_long, K] after the
    // [n_chunks, K] partials
    float* zl = split->partial ? split->partial + (size_t)split->n_chunks * K : nullptr;
    if (int rc = spmm_long_rows_compact(L.rowptr, L.col, L.w, L.rs, L.x, L.ldx, (int)K, split, zl, s)) return rc;
    if (L.w_pos) {  // the same rows under the second weight vector: [n_long, K] more behind the first
      float* zlp = zl ? zl + (size_t)split->n_long * K : nullptr;
      if (int rc = spmm_long_rows_compact(L.rowptr, L.col, L.w_pos, L.rs, L.x, L.ldx, (int)K, split, zlp, s)) return rc;
      zlong_pos = zlp;
    }
    long_row = split->long_row;
    zlong = zl;
    threshold = split->threshold;
    n_long = split->n_long;
  }
  FusedArgs A{L.rowptr, L.col, L.w, L.rs, L.x, L.wt, L.x_root, L.wt_root, L.bias, L.out, L.z_out,
              L.ldx, L.ldo, L.ldz, L.ldr, long_row, zlong, threshold, n_long, (int)N, (int)K, (int)Nout,
              L.pre_scale, L.pre_shift, L.pre_rowsum, stats_part,
              ce ? ce->y : nullptr, ce ? ce->mask : nullptr, ce ? ce->grad_scale : nullptr, ce ? ce->scratch : nullptr,
              ce && ce->mask_groups == 2 ? 2 : 1,
              x_blk ? L.x_blk_cols : 0, L.x_blk_stride, xr_blk ? L.xr_blk_cols : 0, L.xr_blk_stride,
              L.out_blk, L.ob_cols, L.ob_stride, L.w_pos, L.z_pos_out, zlong_pos};
  const int lanes = (int)(K / 4);
  int rc;
  if (dense) {  // the lane grouping of the gather is irrelevant: one instantiation per unrolled width
    if (K == 128 && launch_dense_stream<128>(A, s)) { RGBX_CHECK_LAUNCH("dense_stream_kernel"); rc = RGBX_OK; }
    else if (K == 64 && launch_dense_stream<64>(A, s)) { RGBX_CHECK_LAUNCH("dense_stream_kernel"); rc = RGBX_OK; }
    else if (K == 128) rc = launch<32, 128, true>(A, s);
    else if (K == 64) rc = launch<16, 64, true>(A, s);
    else if (K == 256) rc = launch<64, 256, true>(A, s);
    else rc = launch<32, 0, true>(A, s);
  }
  else if (K == 128) rc = launch<32, 128>(A, s);
  else if (K == 64) rc = launch<16, 64>(A, s);
  else if (K == 256) rc = launch<64, 256>(A, s);
  else if (lanes <= 1) rc = launch<1, 0>(A, s);
  else if (lanes <= 2) rc = launch<2, 0>(A, s);
  else if (lanes <= 4) rc = launch<4, 0>(A, s);
  else if (lanes <= 8) rc = launch<8, 0>(A, s);
  else if (lanes <= 16) rc = launch<16, 0>(A, s);
  else if (lanes <= 32) rc = launch<32, 0>(A, s);
  else rc = launch<64, 0>(A, s);
  if (rc) return rc;
  if (ce) {
    if (int rc2 = reduce_ce_tiles(ce->scratch, (int)cdiv(N, TM), ce->stats, ce->mask_groups == 2 ? 6 : 3, s)) return rc2;
  }
  if (!L.out_colsums) return RGBX_OK;
  return reduce_tile_stats(stats_part, (int)cdiv(N, TM), (int)(2 * Nout), stats_part2, L.out_colsums, N, s);
}

extern "C" int rgbx_spmm_linear_stats_workspace_bytes(int64_t N, int64_t Nout, size_t* bytes) {
  if (!bytes || N < 0 || Nout <= 0) return fail(RGBX_E_ARG, "spmm_linear_stats_workspace_bytes: bad argument");
  *bytes = (size_t)kStatsGather * 2 * (size_t)Nout * sizeof(double) + (size_t)cdiv(N, TM) * 2 * (size_t)Nout * sizeof(float);
  return RGBX_OK;
}
