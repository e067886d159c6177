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
