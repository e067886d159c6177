// Entry points shared between translation units of librgbx_hip (not part of the C ABI).
#pragma once
#include "rgbx_common.h"

namespace rgbx {

// Aggregates of the hub rows of a row-split plan, stored compactly in hub order:
//   zlong[r, 0:d] = rs[long_row[r]] * sum_{p in row long_row[r]} w[p] * x[col[p], 0:d]     (w, rs optional)
// Chunk sums go to split->partial ([n_chunks, d]) and are added in chunk order (reproducible). Used by the fused
// aggregate+transform kernel, which reads a hub row's aggregate from here instead of gathering it.
int spmm_long_rows_compact(const int* rowptr, const int* col, const float* w, const float* rs, const float* x,
                           int64_t ldx, int d, const rgbx_row_split_t* split, float* zlong, hipStream_t s);

}  // namespace rgbx
