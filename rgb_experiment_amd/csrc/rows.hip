// Halo pack / unpack for the 1-D node partition: whole-row gather and unique-index scatter-add.
// New capability (the reference is single-device, itexperiments.py:246); the rows moved here
// are the boundary features the RCCL all-to-all ships between GPUs.
#include "rgbx_common.h"

namespace rgbx {
namespace {

// One wave moves one row at a time; lanes stride the columns VEC floats each.
template <int VEC, bool SCATTER_ADD>
__global__ void __launch_bounds__(256)
rows_kernel(const float* __restrict__ src, int64_t lds, const int* __restrict__ idx, int n, int d,
            float* __restrict__ dst, int64_t ldd) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n; r += gridDim.x * wpb) {
    const int64_t j = idx[r];
    const float* s = SCATTER_ADD ? src + (int64_t)r * lds : src + j * lds;
    float* o = SCATTER_ADD ? dst + j * ldd : dst + (int64_t)r * ldd;
    for (int c = lane * VEC; c < d; c += 64 * VEC) {
      if constexpr (VEC == 4) {
        float4 v = *reinterpret_cast<const float4*>(s + c);
        if constexpr (SCATTER_ADD) {
          const float4 a = *reinterpret_cast<const float4*>(o + c);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(o + c) = v;
      } else {
        float v = s[c];
        if constexpr (SCATTER_ADD) v += o[c];
        o[c] = v;
      }
    }
  }
}

template <bool SCATTER_ADD>
int run(const float* src, int64_t lds, const int32_t* idx, int64_t n, int64_t d, float* dst,
        int64_t ldd, hipStream_t s, const char* name) {
  if (n < 0 || d < 0) return fail(RGBX_E_ARG, "%s: negative size", name);
  if (n == 0 || d == 0) return RGBX_OK;
  if (!src || !idx || !dst) return fail(RGBX_E_ARG, "%s: null pointer", name);
  if (n >= INT32_MAX || d >= INT32_MAX) return fail(RGBX_E_RANGE, "%s: size exceeds int32", name);
  if (lds < d || ldd < d) return fail(RGBX_E_ARG, "%s: leading dimension < d", name);
  int64_t blocks = cdiv(n, 4);
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  const bool v4 = d % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst);
  if (v4)
    rows_kernel<4, SCATTER_ADD><<<(int)blocks, 256, 0, s>>>(src, lds, idx, (int)n, (int)d, dst, ldd);
  else
    rows_kernel<1, SCATTER_ADD><<<(int)blocks, 256, 0, s>>>(src, lds, idx, (int)n, (int)d, dst, ldd);
  RGBX_CHECK_LAUNCH(name);
  return RGBX_OK;
}

// dst[i, c] = src blocked [i, c]: a thread moves one float4; consecutive threads walk a row, so both sides move
// contiguous runs of blk_cols floats
__global__ void __launch_bounds__(256)
blocked_to_rows_kernel(const float* __restrict__ src, int64_t bc, int64_t bs, float* __restrict__ dst, int64_t ldd,
                       int64_t n, int d4, const float* __restrict__ bias) {
  const int64_t total = n * d4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / d4;
    const int c = (int)(i - row * d4) * 4;
    float4 v = *reinterpret_cast<const float4*>(src + (int64_t)(c / (int)bc) * bs + row * bc + (c % (int)bc));
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + c);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    *reinterpret_cast<float4*>(dst + row * ldd + c) = v;
  }
}

// Stand-in for the memory traffic of an exchange on a ONE-GPU emulation of a partitioned rank (bench.py --emulate-rank
// with --emulate-contend): `workgroups` workgroups copy n floats, eight independent 16-byte loads in flight per thread. Few
// workgroups move the bytes at a link-like rate (calibrated by the caller) while the rank's own kernels run on another
// stream, so that the HBM / L2 / CU time a real exchange takes from them is IN the measured step time.
// NT: non-temporal loads and stores — traffic that does not allocate in L2 / Infinity Cache, the optimistic end of what
// inbound DMA writes and outbound reads of a real exchange may do to the caches (the plain form is the pessimistic end).
using f4v = __attribute__((ext_vector_type(4))) float;
template <bool NT>
__global__ void __launch_bounds__(256)
paced_copy_kernel(const f4v* __restrict__ src, f4v* __restrict__ dst, int64_t n4) {
  constexpr int U = 8;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace
}  // namespace rgbx

extern "C" int rgbx_paced_copy_f32(const float* src, float* dst, int64_t n, int workgroups, int nontemporal,
                                   rgbx_stream_t stream) {
  using namespace rgbx;
  if (n < 0 || workgroups <= 0) return fail(RGBX_E_ARG, "paced_copy: bad size");
  if (n == 0) return RGBX_OK;
  if (!src || !dst) return fail(RGBX_E_ARG, "paced_copy: null pointer");
  if (n % 4 || !aligned16(src) || !aligned16(dst)) return fail(RGBX_E_ALIGN, "paced_copy: n % 4 == 0 and 16-byte alignment");
  const int grid = workgroups < kMaxGrid ? workgroups : kMaxGrid;
  if (nontemporal)
    paced_copy_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const f4v*>(src),
                                                                   reinterpret_cast<f4v*>(dst), n / 4);
  else
    paced_copy_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const f4v*>(src),
                                                                    reinterpret_cast<f4v*>(dst), n / 4);
  RGBX_CHECK_LAUNCH("paced_copy");
  return RGBX_OK;
}

extern "C" int rgbx_blocked_to_rows_f32(const float* src, int64_t blk_cols, int64_t blk_stride, float* dst, int64_t ldd,
                                        int64_t n, int64_t d, const float* bias, rgbx_stream_t stream) {
  using namespace rgbx;
  if (n < 0 || d < 0) return fail(RGBX_E_ARG, "blocked_to_rows: negative size");
  if (n == 0 || d == 0) return RGBX_OK;
  if (!src || !dst) return fail(RGBX_E_ARG, "blocked_to_rows: null pointer");
  if (blk_cols <= 0 || blk_cols % 4 || d % blk_cols || blk_stride % 4 || ldd < d || ldd % 4 || !aligned16(src) ||
      !aligned16(dst) || (bias && !aligned16(bias)))
    return fail(RGBX_E_ARG, "blocked_to_rows: block width must be a multiple of 4 that divides d; 16-byte alignment");
  if (n >= INT32_MAX || d >= INT32_MAX) return fail(RGBX_E_RANGE, "blocked_to_rows: size exceeds int32");
  int64_t blocks = cdiv(n * (d / 4), 256);
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  blocked_to_rows_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(src, blk_cols, blk_stride, dst, ldd, n, (int)(d / 4),
                                                                      bias);
  RGBX_CHECK_LAUNCH("blocked_to_rows");
  return RGBX_OK;
}

extern "C" int rgbx_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int64_t n,
                                    int64_t d, float* dst, int64_t ldd, rgbx_stream_t stream) {
  return rgbx::run<false>(src, lds, idx, n, d, dst, ldd, (hipStream_t)stream, "gather_rows");
}

extern "C" int rgbx_scatter_add_rows_f32(const float* src, int64_t lds, const int32_t* idx, int64_t n,
                                         int64_t d, float* dst, int64_t ldd, rgbx_stream_t stream) {
  return rgbx::run<true>(src, lds, idx, n, d, dst, ldd, (hipStream_t)stream, "scatter_add_rows");
}
