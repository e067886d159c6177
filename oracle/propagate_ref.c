/* CPU ORACLE (C restatement) — TEST INFRASTRUCTURE ONLY; never linked into the product.
 *
 * Restates the reference's message + aggregate step in plain C:
 *   message(x_j, norm) = norm.view(-1,1) * x_j                    (models/dagnn.py:57-59)
 *   aggr = 'add' over the target index edge_index[1]              (models/dagnn.py:36,46)
 *   aggr = 'mean': the same sums divided by max(count, 1)         (models/graphsage.py:39,58)
 *
 * oracle_propagate_coo_f32   edge by edge, in edge order, single thread: the literal PyG dataflow
 *                            (gather row, scale, scatter-add) without materialising [E, d].
 * oracle_propagate_csr_f32   the same sums grouped per target (CSR from oracle.csr_from_edges), rows
 *                            spread over OpenMP threads: the strongest plain CPU form of the same
 *                            arithmetic; bench.py times this one as the multi-core CPU baseline.
 * Built by oracle/Makefile (gcc -O3 -fopenmp) into oracle/_build/liboracle_ref.so.
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_propagate_coo_f32(const int64_t* src, const int64_t* dst, const float* w, int64_t E,
                              const float* x, int64_t ldx, float* out, int64_t ldo, int64_t N, int64_t d,
                              int mean) {
  for (int64_t i = 0; i < N; ++i) memset(out + i * ldo, 0, (size_t)d * sizeof(float));
  for (int64_t e = 0; e < E; ++e) {
    const float* xj = x + src[e] * ldx;
    float* oi = out + dst[e] * ldo;
    const float we = w ? w[e] : 1.0f;
    for (int64_t c = 0; c < d; ++c) oi[c] += we * xj[c];
  }
  if (mean) {
    int64_t* cnt = (int64_t*)__builtin_malloc((size_t)N * sizeof(int64_t));
    memset(cnt, 0, (size_t)N * sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) cnt[dst[e]]++;
    for (int64_t i = 0; i < N; ++i) {
      const float inv = 1.0f / (float)(cnt[i] > 1 ? cnt[i] : 1);
      for (int64_t c = 0; c < d; ++c) out[i * ldo + c] *= inv;
    }
    __builtin_free(cnt);
  }
}

void oracle_propagate_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* x,
                              int64_t ldx, float* out, int64_t ldo, int64_t N, int64_t d, int mean,
                              int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < N; ++i) {
    float* oi = out + i * ldo;
    for (int64_t c = 0; c < d; ++c) oi[c] = 0.0f;
    const int32_t s = rowptr[i], e = rowptr[i + 1];
    for (int32_t p = s; p < e; ++p) {
      const float* xj = x + (int64_t)col[p] * ldx;
      const float wp = w ? w[p] : 1.0f;
      for (int64_t c = 0; c < d; ++c) oi[c] += wp * xj[c];
    }
    if (mean) {
      const float inv = 1.0f / (float)(e - s > 1 ? e - s : 1);
      for (int64_t c = 0; c < d; ++c) oi[c] *= inv;
    }
  }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- GATConv's message + aggregate over a per-target CSR (PyG GATConv.forward / message + torch_geometric.utils.softmax,
 * called at reference models/gat.py:28,30; the arithmetic is restated from oracle/ref_cpu.py gat_conv / segment_softmax —
 * PARITY UNPINNED: the reference holds no statement or fixture of it). For target i, head k, in-edges p (source j = col[p]):
 *   s_p = a_src[j,k] + a_dst[i,k];  e_p = s_p > 0 ? s_p : slope * s_p;  alpha_p = exp(e_p - max_p e) / (sum_p exp(e_p - max) + 1e-16)
 *   out[i,k,:] = sum_p alpha_p h[j,k,:]
 * alpha ([nnz, H], optional) receives the coefficients in CSR slot order (what the adjoint needs). */
#include <math.h>

void oracle_gat_forward_csr_f32(const int32_t* rowptr, const int32_t* col, const float* h, const float* a_src,
                                const float* a_dst, float slope, float* out, float* alpha, int64_t N, int64_t H,
                                int64_t C, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const int64_t d = H * C;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < N; ++i) {
    float* oi = out + i * d;
    for (int64_t c = 0; c < d; ++c) oi[c] = 0.0f;
    const int32_t s = rowptr[i], e = rowptr[i + 1];
    for (int64_t k = 0; k < H; ++k) {
      float mx = -INFINITY;
      for (int32_t p = s; p < e; ++p) {
        const float sc = a_src[(int64_t)col[p] * H + k] + a_dst[i * H + k];
        const float ev = sc > 0.0f ? sc : slope * sc;
        if (ev > mx) mx = ev;
      }
      float den = 0.0f;
      for (int32_t p = s; p < e; ++p) {
        const float sc = a_src[(int64_t)col[p] * H + k] + a_dst[i * H + k];
        const float ev = sc > 0.0f ? sc : slope * sc;
        den += expf(ev - mx);
      }
      for (int32_t p = s; p < e; ++p) {
        const float sc = a_src[(int64_t)col[p] * H + k] + a_dst[i * H + k];
        const float ev = sc > 0.0f ? sc : slope * sc;
        const float al = expf(ev - mx) / (den + 1e-16f);
        if (alpha) alpha[(int64_t)p * H + k] = al;
        const float* hj = h + (int64_t)col[p] * d + k * C;
        for (int64_t c = 0; c < C; ++c) oi[k * C + c] += al * hj[c];
      }
    }
  }
}

/* Adjoint, target side: per in-edge p of target i and head k, with g_alpha_p = <gout[i,k,:], h[j,k,:]>:
 *   g_e_p = alpha_p (g_alpha_p - sum_q alpha_q g_alpha_q);  gs[p,k] = g_e_p * (s_p > 0 ? 1 : slope);  g_a_dst[i,k] = sum_p gs[p,k]
 * (softmax Jacobian incl. the 1e-16 in the denominator: d alpha_p / d e_q = [p == q] alpha_p - alpha_p alpha_q). */
void oracle_gat_backward_dst_csr_f32(const int32_t* rowptr, const int32_t* col, const float* h, const float* a_src,
                                     const float* a_dst, float slope, const float* alpha, const float* gout, float* gs,
                                     float* g_a_dst, int64_t N, int64_t H, int64_t C, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const int64_t d = H * C;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < N; ++i) {
    const int32_t s = rowptr[i], e = rowptr[i + 1];
    for (int64_t k = 0; k < H; ++k) {
      const float* gi = gout + i * d + k * C;
      double dot_sum = 0.0;
      for (int32_t p = s; p < e; ++p) {
        const float* hj = h + (int64_t)col[p] * d + k * C;
        float ga = 0.0f;
        for (int64_t c = 0; c < C; ++c) ga += gi[c] * hj[c];
        gs[(int64_t)p * H + k] = ga;  /* g_alpha for now */
        dot_sum += (double)alpha[(int64_t)p * H + k] * (double)ga;
      }
      float acc = 0.0f;
      for (int32_t p = s; p < e; ++p) {
        const float sc = a_src[(int64_t)col[p] * H + k] + a_dst[i * H + k];
        const float ge = alpha[(int64_t)p * H + k] * (gs[(int64_t)p * H + k] - (float)dot_sum);
        const float g = ge * (sc > 0.0f ? 1.0f : slope);
        gs[(int64_t)p * H + k] = g;
        acc += g;
      }
      g_a_dst[i * H + k] = acc;
    }
  }
}

/* Adjoint, source side, over the TRANSPOSED CSR (rows = sources j, col_t = targets i); fwd_slot[p_t] = the forward CSR slot of
 * the same edge:  g_h[j,k,:] = sum_{p_t} alpha[f,k] gout[i,k,:],  g_a_src[j,k] = sum_{p_t} gs[f,k],  f = fwd_slot[p_t]. */
void oracle_gat_backward_src_csr_f32(const int32_t* rowptr_t, const int32_t* col_t, const int64_t* fwd_slot,
                                     const float* alpha, const float* gs, const float* gout, float* g_h, float* g_a_src,
                                     int64_t N, int64_t H, int64_t C, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const int64_t d = H * C;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t j = 0; j < N; ++j) {
    float* gj = g_h + j * d;
    for (int64_t c = 0; c < d; ++c) gj[c] = 0.0f;
    for (int64_t k = 0; k < H; ++k) g_a_src[j * H + k] = 0.0f;
    for (int32_t p = rowptr_t[j]; p < rowptr_t[j + 1]; ++p) {
      const int64_t f = fwd_slot[p];
      const float* gi = gout + (int64_t)col_t[p] * d;
      for (int64_t k = 0; k < H; ++k) {
        const float al = alpha[f * H + k];
        for (int64_t c = 0; c < C; ++c) gj[k * C + c] += al * gi[k * C + c];
        g_a_src[j * H + k] += gs[f * H + k];
      }
    }
  }
}

/* Stable grouping of E items by key in [0, N): order[pos] = the item placed at position pos, items of one key in input
 * order — what a STABLE sort by key returns (oracle/ref_cpu.py csr_from_edges' torch.sort(stable=True): 6.6 s for 62 M keys
 * on 8 cores; this counting sort: one histogram, one prefix sum, one in-order placement pass). rowptr [N + 1] (int64). */
void oracle_stable_group_i64(const int64_t* key, int64_t E, int64_t N, int64_t* rowptr, int64_t* order) {
  for (int64_t i = 0; i <= N; ++i) rowptr[i] = 0;
  for (int64_t e = 0; e < E; ++e) rowptr[key[e] + 1]++;
  for (int64_t i = 0; i < N; ++i) rowptr[i + 1] += rowptr[i];
  int64_t* next = (int64_t*)__builtin_malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
  for (int64_t i = 0; i < N; ++i) next[i] = rowptr[i];
  for (int64_t e = 0; e < E; ++e) order[next[key[e]]++] = e;
  __builtin_free(next);
}
