// Graph preparation: int64 edge list -> CSR grouped by aggregation index (stable), degree
// normalisation. Replaces the per-call index bookkeeping of PyG's propagate / gcn_norm
// (reference models/dagnn.py:12-31; call sites models/gcn.py:27, graphsage.py:53-58, gat.py:28,
// appnp_stack.py:29). Runs once per edge_index; all outputs live in caller-owned buffers.
#include "rgbx_common.h"

#include <rocprim/rocprim.hpp>

namespace rgbx {

char* err_buf() {
  static thread_local char buf[256] = "";
  return buf;
}

namespace {

// Slot e < E is input edge e; slot E + i is the self-loop added for node i (loops_mode != 0).
// A removed self-loop gets key N, which sorts behind every real row and is cut off by
// rowptr[N].
__global__ void __launch_bounds__(256)
fill_keys_kernel(const int64_t* __restrict__ agg_row, const int64_t* __restrict__ other_row,
                 int64_t E, int64_t M, int N, int loops_mode, int* __restrict__ keys,
                 int* __restrict__ ids) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M;
       e += (int64_t)gridDim.x * blockDim.x) {
    int key;
    if (e < E) {
      const int64_t t = agg_row[e];
      key = (loops_mode != RGBX_LOOPS_KEEP && other_row[e] == t) ? N : (int)t;
    } else {
      key = (int)(e - E);
    }
    keys[e] = key;
    ids[e] = (int)e;
  }
}

// After the sort: gather index per slot, and rowptr from the key boundaries. Slot p opens every
// row in (key[p-1], key[p]]; the last slot also closes the rows behind its key.
__global__ void __launch_bounds__(256)
finish_csr_kernel(const int* __restrict__ keys, const int* __restrict__ perm,
                  const int64_t* __restrict__ other_row, int64_t E, int64_t M, int N,
                  int* __restrict__ col, int* __restrict__ rowptr) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < M;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int key = keys[p];
    const int id = perm[p];
    col[p] = id < E ? (int)other_row[id] : (int)(id - E);
    const int prev = p > 0 ? keys[p - 1] : -1;
    for (int r = prev + 1; r <= key; ++r) rowptr[r] = (int)p;
    if (p == M - 1)
      for (int r = key + 1; r <= N; ++r) rowptr[r] = (int)M;
  }
}

__global__ void __launch_bounds__(256) zero_rowptr_kernel(int* rowptr, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    rowptr[i] = 0;
}

__global__ void __launch_bounds__(256)
deg_inv_sqrt_kernel(const int* __restrict__ rowptr, int N, float* __restrict__ dis) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int cnt = rowptr[i + 1] - rowptr[i];
    dis[i] = cnt > 0 ? 1.0f / sqrtf((float)cnt) : 0.0f;  // inf -> 0, dagnn.py:29-30
  }
}

__global__ void __launch_bounds__(256)
inv_degree_kernel(const int* __restrict__ rowptr, int N, float* __restrict__ inv) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int cnt = rowptr[i + 1] - rowptr[i];
    inv[i] = 1.0f / (float)(cnt > 1 ? cnt : 1);
  }
}

// One wave per row, lanes over its slots: w = (dis[src] * 1) * dis[tgt] as dagnn.py:31.
__global__ void __launch_bounds__(256)
gcn_norm_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int N,
                const float* __restrict__ dis, float* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < N; row += gridDim.x * wpb) {
    const int s = rowptr[row], e = rowptr[row + 1];
    const float di = dis[row];
    for (int p = s + lane; p < e; p += 64) w[p] = dis[col[p]] * di;
  }
}

int sort_bits(int64_t N) {
  int b = 1;
  while (((int64_t)1 << b) <= N) ++b;  // keys lie in [0, N]
  return b;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int grid_for(int64_t n, int per_block = 256) {
  int64_t g = cdiv(n > 0 ? n : 1, per_block);
  return (int)(g < kMaxGrid ? g : kMaxGrid);
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_version(void) { return RGBX_VERSION; }

extern "C" const char* rgbx_last_error_string(void) { return err_buf(); }

extern "C" int rgbx_csr_workspace_bytes(int64_t E, int64_t N, size_t* bytes) {
  if (!bytes || E < 0 || N < 0) return fail(RGBX_E_ARG, "csr_workspace_bytes: bad argument");
  const int64_t M = E + N;
  if (M >= INT32_MAX || N >= INT32_MAX) return fail(RGBX_E_RANGE, "E+N=%lld exceeds int32", (long long)M);
  size_t tmp = 0;
  int* nul = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, nul, nul, nul, nul, (size_t)(M > 0 ? M : 1),
                                           0, sort_bits(N), (hipStream_t) nullptr);
  if (e != hipSuccess) return hip_fail(e, "radix_sort_pairs(size query)");
  *bytes = 3 * align256((size_t)(M > 0 ? M : 1) * sizeof(int)) + align256(tmp);
  return RGBX_OK;
}

extern "C" int rgbx_csr_build(const int64_t* agg_row, const int64_t* other_row, int64_t E, int64_t N,
                              int loops_mode, int32_t* rowptr, int32_t* col, int32_t* perm,
                              void* workspace, size_t workspace_bytes, rgbx_stream_t stream) {
  if (E < 0 || N < 0 || !rowptr || (E > 0 && (!agg_row || !other_row)))
    return fail(RGBX_E_ARG, "csr_build: null pointer or negative size");
  if (loops_mode < RGBX_LOOPS_KEEP || loops_mode > RGBX_LOOPS_REMOVE_ADD)
    return fail(RGBX_E_ARG, "csr_build: loops_mode %d", loops_mode);
  size_t need = 0;
  if (int rc = rgbx_csr_workspace_bytes(E, N, &need)) return rc;
  const int64_t M = loops_mode == RGBX_LOOPS_KEEP ? E : E + N;
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    zero_rowptr_kernel<<<grid_for(N + 1), 256, 0, s>>>(rowptr, N + 1);
    RGBX_CHECK_LAUNCH("zero_rowptr_kernel");
    return RGBX_OK;
  }
  if (!col || !perm || !workspace) return fail(RGBX_E_ARG, "csr_build: null output/workspace");
  if (workspace_bytes < need)
    return fail(RGBX_E_WS, "csr_build: workspace %zu < %zu bytes", workspace_bytes, need);

  const size_t slab = align256((size_t)(E + N > 0 ? E + N : 1) * sizeof(int));
  char* ws = static_cast<char*>(workspace);
  int* keys_in = reinterpret_cast<int*>(ws);
  int* keys_out = reinterpret_cast<int*>(ws + slab);
  int* ids_in = reinterpret_cast<int*>(ws + 2 * slab);
  void* sort_tmp = ws + 3 * slab;
  size_t sort_bytes = workspace_bytes - 3 * slab;

  fill_keys_kernel<<<grid_for(M), 256, 0, s>>>(agg_row, other_row, E, M, (int)N, loops_mode,
                                               keys_in, ids_in);
  RGBX_CHECK_LAUNCH("fill_keys_kernel");
  // LSD radix sort is stable: slots of one row keep the rewritten list's order.
  RGBX_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys_out, ids_in, perm,
                                     (size_t)M, 0, sort_bits(N), s));
  finish_csr_kernel<<<grid_for(M), 256, 0, s>>>(keys_out, perm, other_row, E, M, (int)N, col,
                                                rowptr);
  RGBX_CHECK_LAUNCH("finish_csr_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_deg_inv_sqrt_f32(const int32_t* rowptr, int64_t N, float* dis,
                                     rgbx_stream_t stream) {
  if (N < 0 || !rowptr || (N > 0 && !dis)) return fail(RGBX_E_ARG, "deg_inv_sqrt: bad argument");
  if (N == 0) return RGBX_OK;
  deg_inv_sqrt_kernel<<<grid_for(N), 256, 0, (hipStream_t)stream>>>(rowptr, (int)N, dis);
  RGBX_CHECK_LAUNCH("deg_inv_sqrt_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_inv_degree_f32(const int32_t* rowptr, int64_t N, float* inv,
                                   rgbx_stream_t stream) {
  if (N < 0 || !rowptr || (N > 0 && !inv)) return fail(RGBX_E_ARG, "inv_degree: bad argument");
  if (N == 0) return RGBX_OK;
  inv_degree_kernel<<<grid_for(N), 256, 0, (hipStream_t)stream>>>(rowptr, (int)N, inv);
  RGBX_CHECK_LAUNCH("inv_degree_kernel");
  return RGBX_OK;
}

extern "C" int rgbx_gcn_norm_f32(const int32_t* rowptr, const int32_t* col, int64_t N,
                                 const float* dis, float* w, rgbx_stream_t stream) {
  if (N < 0 || !rowptr || !col || !dis || !w) return fail(RGBX_E_ARG, "gcn_norm: bad argument");
  if (N == 0) return RGBX_OK;
  gcn_norm_kernel<<<grid_for(N, 4), 256, 0, (hipStream_t)stream>>>(rowptr, col, (int)N, dis, w);
  RGBX_CHECK_LAUNCH("gcn_norm_kernel");
  return RGBX_OK;
}
