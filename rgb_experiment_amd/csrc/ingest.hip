// Edge-list ingest on the device: coalesce (sort by (row, col), drop duplicates) and to_undirected (mirror every
// edge, then coalesce) of the reference's int64 [2, E] edge_index — the one-shot edits the reference makes before the
// path (torch_sparse.coalesce at rd2pd.py:92-93, torch_geometric.utils.to_undirected at itexperiments.py:235-238;
// SURVEY 8(f3)). At |E| = 60 M the mirrored list is 120 M pairs: a radix-sort job.
//
// One 64-bit key per edge, key = row * N + col (N < 2^31, so the key takes 2 * bits(N) <= 62 bits); rocPRIM's LSD
// radix sort over exactly those bits (N = 2 M: 42 bits, 6 passes instead of the 8 of a full int64 sort), rocPRIM's
// unique over the sorted keys, then one pass that splits the keys back into the int64 rows of the output. Keys are
// unique per (row, col), so the result does not depend on sort stability: it IS the sorted set of pairs, bit for bit
// what the CPU statement computes.
#include "rgbx_common.h"

#include <rocprim/rocprim.hpp>

namespace rgbx {
namespace {

// Slot e < E is edge (row[e], col[e]); with `mirror`, slot E + e is its reverse (col[e], row[e]).
// Out-of-range endpoints are counted in *bad (checked by the caller after its read-back) and clamped into range so that
// nothing downstream indexes outside its buffers.
__global__ void __launch_bounds__(256)
edge_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E, int64_t M, int64_t N,
                 uint64_t* __restrict__ keys, unsigned long long* __restrict__ bad) {
  unsigned long long my_bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r, c;
    if (e < E) {
      r = row[e];
      c = col[e];
    } else {
      r = col[e - E];
      c = row[e - E];
    }
    if (r < 0 || r >= N || c < 0 || c >= N) {
      ++my_bad;
      r = r < 0 ? 0 : (r >= N ? N - 1 : r);
      c = c < 0 ? 0 : (c >= N ? N - 1 : c);
    }
    keys[e] = (uint64_t)r * (uint64_t)N + (uint64_t)c;
  }
  if (my_bad) atomicAdd(bad, my_bad);
}

// counts[0] = number of unique keys (written by rocprim::unique on the device); rows of the [2, cap] output
__global__ void __launch_bounds__(256)
split_keys_kernel(const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ count, int64_t cap,
                  int64_t N, int64_t* __restrict__ out_row, int64_t* __restrict__ out_col) {
  const int64_t n = (int64_t)*count < cap ? (int64_t)*count : cap;
  const uint64_t un = (uint64_t)N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = keys[i];
    const uint64_t r = k / un;
    out_row[i] = (int64_t)r;
    out_col[i] = (int64_t)(k - r * un);
  }
}

int key_bits(int64_t N) {  // keys lie in [0, N * N)
  int b = 1;
  while (b < 63 && ((uint64_t)1 << b) < (uint64_t)N * (uint64_t)N) ++b;
  return b;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int grid_for(int64_t n) {
  int64_t g = cdiv(n > 0 ? n : 1, 256);
  return (int)(g < kMaxGrid ? g : kMaxGrid);
}

struct Layout {
  size_t slab, sort_tmp, uniq_tmp, total;
};

int layout_for(int64_t M, int64_t N, Layout* L) {
  const size_t m = (size_t)(M > 0 ? M : 1);
  uint64_t* nk = nullptr;
  unsigned long long* nc = nullptr;
  size_t st = 0, ut = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, st, nk, nk, m, 0, key_bits(N), (hipStream_t) nullptr);
  if (e != hipSuccess) return hip_fail(e, "radix_sort_keys(size query)");
  e = rocprim::unique(nullptr, ut, nk, nk, nc, m, rocprim::equal_to<uint64_t>(), (hipStream_t) nullptr);
  if (e != hipSuccess) return hip_fail(e, "unique(size query)");
  L->slab = align256(m * sizeof(uint64_t));
  L->sort_tmp = align256(st);
  L->uniq_tmp = align256(ut);
  L->total = 2 * L->slab + (L->sort_tmp > L->uniq_tmp ? L->sort_tmp : L->uniq_tmp);
  return RGBX_OK;
}

}  // namespace
}  // namespace rgbx

using namespace rgbx;

extern "C" int rgbx_coalesce_workspace_bytes(int64_t E, int64_t N, int mirror, size_t* bytes) {
  if (!bytes || E < 0 || N < 0) return fail(RGBX_E_ARG, "coalesce_workspace_bytes: bad argument");
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "coalesce: N=%lld exceeds int32", (long long)N);
  Layout L;
  if (int rc = layout_for(mirror ? 2 * E : E, N, &L)) return rc;
  *bytes = L.total;
  return RGBX_OK;
}

extern "C" int rgbx_coalesce_keys_i64(const int64_t* row, const int64_t* col, int64_t E, int64_t N, int mirror,
                                      uint64_t* keys_out, uint64_t* counts, void* workspace, size_t workspace_bytes,
                                      rgbx_stream_t stream) {
  if (E < 0 || N < 0 || !counts || (E > 0 && (!row || !col || !keys_out || !workspace)))
    return fail(RGBX_E_ARG, "coalesce_keys: null pointer or negative size");
  if (N >= INT32_MAX) return fail(RGBX_E_RANGE, "coalesce: N=%lld exceeds int32", (long long)N);
  hipStream_t s = (hipStream_t)stream;
  RGBX_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), s));
  const int64_t M = mirror ? 2 * E : E;
  if (M == 0) return RGBX_OK;
  if (N == 0) return fail(RGBX_E_ARG, "coalesce_keys: edges over an empty node set");
  Layout L;
  if (int rc = layout_for(M, N, &L)) return rc;
  if (workspace_bytes < L.total)
    return fail(RGBX_E_WS, "coalesce_keys: workspace %zu < %zu bytes", workspace_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(ws);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(ws + L.slab);
  void* tmp = ws + 2 * L.slab;
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(counts);

  edge_keys_kernel<<<grid_for(M), 256, 0, s>>>(row, col, E, M, N, keys_a, cnt + 1);
  RGBX_CHECK_LAUNCH("edge_keys_kernel");
  size_t st = L.sort_tmp;
  RGBX_HIP(rocprim::radix_sort_keys(tmp, st, keys_a, keys_b, (size_t)M, 0, key_bits(N), s));
  size_t ut = L.uniq_tmp;
  RGBX_HIP(rocprim::unique(tmp, ut, keys_b, keys_out, cnt, (size_t)M, rocprim::equal_to<uint64_t>(), s));
  return RGBX_OK;
}

extern "C" int rgbx_split_edge_keys_i64(const uint64_t* keys, const uint64_t* counts, int64_t cap, int64_t N,
                                        int64_t* out_row, int64_t* out_col, rgbx_stream_t stream) {
  if (cap < 0 || N < 0 || !counts || (cap > 0 && (!keys || !out_row || !out_col || N == 0)))
    return fail(RGBX_E_ARG, "split_edge_keys: bad argument");
  if (cap == 0) return RGBX_OK;
  split_keys_kernel<<<grid_for(cap), 256, 0, (hipStream_t)stream>>>(
      keys, reinterpret_cast<const unsigned long long*>(counts), cap, N, out_row, out_col);
  RGBX_CHECK_LAUNCH("split_keys_kernel");
  return RGBX_OK;
}
