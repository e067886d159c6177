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

// Per-tile records -> totals, in a fixed order (reproducible), shared by every kernel that leaves one record per 32-row
// tile (spmm_linear.hip defines them):
//  - loss statistics: `scratch` holds `tiles` records of W doubles (W = 3 or 6) followed by room for kCeGather * W more;
//    stats[0:W] = the sums.
//  - column sums: `part` holds `tiles` records of width2 floats, `part2` room for kStatsGather * width2 doubles;
//    sums[0:width2] = the sums in fp64.
constexpr int kTileRows = 32;
constexpr int kCeGather = 64;
constexpr int kStatsGather = 1024;
int reduce_ce_tiles(double* scratch, int tiles, double* stats, int W, hipStream_t s);
int reduce_tile_stats(const float* part, int tiles, int width2, double* part2, double* sums, int64_t n_rows, hipStream_t s);

}  // namespace rgbx
