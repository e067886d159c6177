/*
 * rgbx_hip.h — C ABI of librgbx_hip.so, the MI355X (gfx950) native library behind the
 * message-passing hot path of rgb-experiment.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo root;
 * "[PyG]" = the arithmetic lives in the un-vendored torch_geometric dependency the reference
 * imports at that line):
 *
 *   rgbx_csr_*            self-loop rewrite + grouping of edges by aggregation index that
 *                         MessagePassing.propagate performs implicitly on every call
 *                         (models/gcn.py:27,29; models/graphsage.py:53-58; models/gat.py:28,30;
 *                         models/appnp_stack.py:29), restated in-repo by models/dagnn.py:20-24.
 *   rgbx_gcn_norm_f32     gcn_norm, models/dagnn.py:12-31 (degree over the target index,
 *                         deg^-1/2 with inf -> 0, w = dis[src] * dis[tgt]).
 *   rgbx_inv_degree_f32   the 1/max(count,1) of aggr='mean' (models/graphsage.py:39,58) [PyG].
 *   rgbx_spmm_csr_f32     MessagePassing.propagate with aggr='add' / 'mean' and message
 *                         norm * x_j (models/dagnn.py:34-36,46,57-59; models/graphsage.py:58).
 *   rgbx_appnp_f32        APPNP.forward's K-step recurrence (models/appnp_stack.py:29),
 *                         restated in-repo by models/pta.py:79-84.
 *   rgbx_dagnn_gate_*     Prop.forward's sigmoid-gated mix of the K+1 hops (models/dagnn.py:49-55) and its backward.
 *   rgbx_gat_*            GATConv.forward/message + segment softmax (models/gat.py:28,30) [PyG].
 *   rgbx_gemm_tn_f32      dW = dY^T X of the nn.Linear / conv.lin layers under loss.backward()
 *                         (itexperiments.py:439; layers at models/gcn.py:18-21, appnp_stack.py:19-20).
 *   rgbx_bn_* / rgbx_affine_cols_f32
 *                         nn.BatchNorm1d over all N nodes between conv layers (models/gcn.py:23,28;
 *                         graphsage.py:24,29; gat.py:23,29; appnp_stack.py:21,27), forward and backward.
 *   rgbx_masked_nll_*     nn.NLLLoss on out[mask] and the arg-max accuracy (itexperiments.py:400,
 *                         429,434,624-626,643).
 *   rgbx_coalesce_* / rgbx_split_edge_keys_i64
 *                         the one-shot edge-list edits in front of the path: torch_sparse.coalesce
 *                         (rd2pd.py:92-93) and torch_geometric.utils.to_undirected (itexperiments.py:235-238) [PyG].
 *   rgbx_gather_rows_f32 / rgbx_scatter_add_rows_f32
 *                         halo pack / unpack for the 1-D node partition (new capability; the
 *                         reference is single-device, itexperiments.py:246).
 *
 * Conventions
 *   - Plain C types only. Every pointer except `rgbx_last_error_string`'s result is a DEVICE
 *     pointer owned by the caller; the library never allocates or frees device memory.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream); nothing inside synchronises the device.
 *   - Return value: 0 = OK; < 0 = argument/shape/alignment error (RGBX_E_*); > 0 = hipError_t.
 *     `rgbx_last_error_string()` describes the calling thread's last failure.
 *   - Feature matrices are fp32 row-major with an explicit leading dimension in ELEMENTS.
 *   - Indices inside the library are int32 (requires N < 2^31 and E' < 2^31); `edge_index`
 *     arrives as the reference's int64 [2, E].
 */
#ifndef RGBX_HIP_H
#define RGBX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBX_VERSION 501 /* major*10000 + minor*100 + patch */

#define RGBX_OK 0
#define RGBX_E_ARG (-1)    /* null pointer / negative size / bad enum */
#define RGBX_E_RANGE (-2)  /* N or E' does not fit int32 */
#define RGBX_E_ALIGN (-3)  /* pointer or leading dimension not aligned for the vector path */
#define RGBX_E_WS (-4)     /* workspace too small */
#define RGBX_E_SHAPE (-5)  /* unsupported shape (e.g. heads*channels combination) */

typedef void* rgbx_stream_t;

/* Self-loop rewrite applied while grouping edges (what the conv layers do before propagate). */
enum {
  RGBX_LOOPS_KEEP = 0,          /* edges as given (SAGEConv [PyG], models/graphsage2.py:20-23) */
  RGBX_LOOPS_ADD_REMAINING = 1, /* add_remaining_self_loops, fill 1 (models/dagnn.py:20-24) */
  RGBX_LOOPS_REMOVE_ADD = 2     /* remove_self_loops + add_self_loops (models/graphsage.py:53-56) */
};
/* For unweighted graphs modes 1 and 2 give the same edge list: the non-loop edges in their
 * original order followed by one self-loop per node, nodes ascending. */

int rgbx_version(void);
const char* rgbx_last_error_string(void);

/* ---- graph preparation: int64 edge list -> CSR grouped by `agg_row` ------------------------ */

/* Bytes of scratch `rgbx_csr_build` needs for E input edges over N nodes. */
int rgbx_csr_workspace_bytes(int64_t E, int64_t N, size_t* bytes);

/* Group edges by aggregation index with a STABLE sort.
 *   agg_row[e]   = index the edge aggregates INTO  (edge_index[1] forward, edge_index[0] for the
 *                  transposed graph used by backward)
 *   other_row[e] = index the edge gathers FROM
 * Outputs (capacity E + N entries each for col / perm; rowptr has N + 1):
 *   rowptr[i]..rowptr[i+1]  edges aggregating into node i; rowptr[N] = E'
 *   col[p]                  gather index of CSR slot p
 *   perm[p]                 id of the edge in slot p: e in [0,E) = column of edge_index,
 *                           E + i = the added self-loop of node i
 * Within a row, slots keep the order of the rewritten edge list (original edges first, in
 * input order; the added self-loop last). */
int rgbx_csr_build(const int64_t* agg_row, const int64_t* other_row, int64_t E, int64_t N,
                   int loops_mode, int32_t* rowptr, int32_t* col, int32_t* perm, void* workspace,
                   size_t workspace_bytes, rgbx_stream_t stream);

/* dis[i] = (rowptr[i+1]-rowptr[i])^-1/2, 0 where the count is 0 (models/dagnn.py:27-30). */
int rgbx_deg_inv_sqrt_f32(const int32_t* rowptr, int64_t N, float* dis, rgbx_stream_t stream);

/* w[p] = dis[col[p]] * dis[i] for every slot p of row i (models/dagnn.py:31). `dis` must come
 * from the FORWARD (target-grouped) CSR also when (rowptr, col) is the transposed one. */
int rgbx_gcn_norm_f32(const int32_t* rowptr, const int32_t* col, int64_t N, const float* dis,
                      float* w, rgbx_stream_t stream);

/* inv[i] = 1 / max(rowptr[i+1]-rowptr[i], 1) — the divisor of aggr='mean'. */
int rgbx_inv_degree_f32(const int32_t* rowptr, int64_t N, float* inv, rgbx_stream_t stream);

/* ---- aggregation ------------------------------------------------------------------------- */

/* Optional plan for rows with very many slots (hub nodes). Rows with more than `threshold` slots are
 * skipped by the row-per-wave kernel; each is cut into consecutive chunks (chunk c covers slots
 * [chunk_begin[c], chunk_end[c]) of the CSR), one wave sums one chunk into partial[c, :], and the
 * partials of long row r (chunks long_chunk_ptr[r] .. long_chunk_ptr[r+1]) are added in chunk order
 * before the epilogue — bitwise reproducible, no atomics. All arrays are device memory owned by the
 * caller; `partial` is an [n_chunks, d] fp32 scratch. NULL (or threshold 0) = no splitting. */
typedef struct rgbx_row_split {
  int32_t threshold;
  int32_t n_chunks;
  int32_t n_long;
  const int32_t* chunk_begin;    /* [n_chunks] */
  const int32_t* chunk_end;      /* [n_chunks] */
  const int32_t* chunk_row;      /* [n_chunks] row id of every chunk (used by the GAT kernels) */
  const int32_t* long_row;       /* [n_long] row ids, ascending */
  const int32_t* long_chunk_ptr; /* [n_long + 1] */
  float* partial;                /* [n_chunks, d] fp32 scratch; GAT: [n_chunks, H*C + 2*H] */
} rgbx_row_split_t;

/* out[i,:] = a * rs[i] * sum_{p in row i} w[p] * x[col[p],:]  +  b * y[i,:]  +  bias[:]
 *   w  == NULL -> every weight is 1;  rs == NULL -> every row scale is 1;
 *   y  == NULL -> no additive term (b ignored);  bias == NULL -> no per-column term (the conv
 *   layer's `out += bias`, fused into the store).  `out` may alias `y` but not `x`.
 * N rows, d columns; x has n_src rows (col[] < n_src is the caller's contract). */
int rgbx_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* rs,
                      const float* x, int64_t ldx, const float* y, int64_t ldy, const float* bias,
                      float* out, int64_t ldo, int64_t N, int64_t d, float a, float b,
                      const rgbx_row_split_t* split, rgbx_stream_t stream);

/* rgbx_spmm_csr_f32 with an epilogue over the finished output rows, for layers that transform BEFORE they aggregate
 * (in > out — every default configuration of the reference, initial_params.py:25-29: F -> 64 -> C with C = 7, 6, 3, 40 ...),
 * whose last kernel is this gather and not the fused aggregate+transform one. A lane group holds a target's complete
 * output row, so what the reference runs next as separate passes over [N, d] comes out of the registers:
 *   out_colsums ([2, d] doubles): column sums of out and out^2 — the statistics of the training-mode BatchNorm1d
 *     behind the layer (models/gcn.py:27 then :28); per 32-row tile an fp32 record (sum, squared deviations from the
 *     tile's own mean), the tiles' sum x and sum x^2 = M2 + S^2 / n added in fp64 in a fixed order (a column whose
 *     spread is far below its mean keeps its variance);
 *     stats_ws: rgbx_spmm_linear_stats_workspace_bytes(N, d) bytes, 8-byte aligned. `out` is written as usual.
 *   ce: log_softmax + NLLLoss on out[mask] + arg-max (models/gcn.py:31, itexperiments.py:400,429,434,624-626) exactly as
 *     rgbx_ce_epilogue_t describes: statistics only (out not written, may be NULL; rows no mask selects are not even
 *     gathered) or, with grad_scale, `out` = the loss gradient w.r.t. the logits. n_classes = C <= d: columns [C, d) are
 *     PADDING (rows padded to a multiple of 4 floats: C = 7 -> d = 8) — no part in max / sum-exp / arg-max, gradient 0;
 *     0 = all d columns. Labels outside [0, C) deselect a row. ce->scratch as for rgbx_spmm_linear_f32.
 * Exactly one of out_colsums / ce. d % 4 == 0, d <= 256 (rgbx_spmm_csr_epilogue_supported); x, y, out, bias 16-byte
 * aligned, leading dimensions % 4 == 0. The logits are bit-identical to rgbx_spmm_csr_f32's (same summation order).
 * `split`: as rgbx_spmm_linear_f32 — split->partial must hold (n_chunks + n_long) * d floats. */
typedef struct rgbx_spmm_epilogue {
  const struct rgbx_ce_epilogue* ce; /* or NULL */
  int64_t n_classes;
  double* out_colsums;               /* or NULL */
  void* stats_ws;
  size_t stats_ws_bytes;
} rgbx_spmm_epilogue_t;

int rgbx_spmm_csr_epilogue_supported(int64_t d);
int rgbx_spmm_csr_epilogue_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* rs,
                               const float* x, int64_t ldx, const float* y, int64_t ldy, const float* bias,
                               float* out, int64_t ldo, int64_t N, int64_t d, float a, float b,
                               const rgbx_row_split_t* split, const rgbx_spmm_epilogue_t* epi,
                               rgbx_stream_t stream);

/* The same weighted row gather for SHORT rows over a SMALL table, one lane group per target row (a wave takes 64 / G rows):
 *   out[i,:] = sum_{p in row i} w[p] * x[col[p],:] + bias      (w optional: NULL = 1; slot order, deterministic)
 * What it replaces: the first Linear of every reference model applied to bag-of-words features — x W^T with x the
 * row-normalised counts of itexperiments.py:296 (models/gcn.py:27, graphsage.py:49-50, appnp_stack.py:25, dagnn.py:73) —
 * computed over the features' non-zeros: CSR rows = nodes, col = feature ids, w = feature values, the gathered table
 * W^T [F, d] (L2-resident). rgbx_spmm_csr_f32 gives the same sums (up to summation order) with one wave per row; with the
 * table in cache that form is bound by wave issue. d % 4 == 0, d <= 256; x, out, bias 16-byte aligned, ldx, ldo % 4 == 0.
 * No row-split plan: a long row is walked by its one lane group. */
int rgbx_spmm_csr_short_rows_supported(int64_t d);
int rgbx_spmm_csr_short_rows_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* x, int64_t ldx,
                                 const float* bias, float* out, int64_t ldo, int64_t N, int64_t d, rgbx_stream_t stream);

/* Fused aggregate-then-transform for layers whose propagate commutes with their Linear
 * (GCNConv, the mean branch of SAGEConv / my_SAGEConv):
 *   z[i,:]   = rs[i] * sum_{p in row i} w[p] * x[col[p],:]          (w, rs optional as above)
 *   out[i,:] = z[i,:] * wt + x_root[i,:] * wt_root + bias,   wt = W^T as a [K, Nout] row-major matrix
 * The [32 x K] tile of z lives in LDS and is multiplied on v_mfma_f32_32x32x2_f32 (exact fp32). If z_out is
 * not NULL the aggregate is also stored (ldz), for the weight gradient dy^T z of a training pass.
 * Root term (optional; x_root and wt_root both NULL or both set): SAGEConv's lin_r(x_i)
 * (models/graphsage.py:50,60; graphsage2.py:29 [PyG]) — x_root [N, K] (ldr) are the targets' own rows
 * (usually x itself), wt_root = Wr^T [K, Nout].
 * Supported shapes: K % 4 == 0, K <= 256, Nout % 32 == 0, and Nout <= 256 with a root term
 * (rgbx_spmm_linear_supported); otherwise RGBX_E_SHAPE and the caller runs rgbx_spmm_csr_f32 + GEMMs.
 * `split` (optional): hub rows are aggregated first by the split-row kernels (chunk sums added in chunk order)
 * and the fused kernel copies their finished aggregates; `split->partial` must then hold
 * (n_chunks + n_long) * K floats.
 * `pre_scale`, `pre_shift` ([K]) and `pre_rowsum` ([N]) — all NULL or all set: the layer's input is the affinely
 * mapped matrix x' = x * pre_scale + pre_shift (per column), which is never materialised: a training-mode
 * BatchNorm1d in front of the conv layer (models/gcn.py:28 then :29). By linearity
 *   z[i,:] = pre_scale * (rs[i] * sum_p w[p] x[col[p],:]) + pre_shift * pre_rowsum[i],
 * pre_rowsum[i] = rs[i] * sum_{p in row i} w[p] (the caller's per-graph constant); the root rows get the map as
 * they are loaded; z_out receives the mapped aggregate (what dW = dy^T z needs).
 * `out_colsums` ([2, Nout] doubles, optional): out_colsums[0,c] = sum_i out[i,c], [1,c] = sum_i out[i,c]^2 — the
 * statistics of a training-mode BatchNorm1d that follows the layer (models/gcn.py:27 then :28), taken from the
 * MFMA accumulators of the 32-row tiles (per tile an fp32 record of the sum and of the squared deviations from the
 * tile's mean; sum x and sum x^2 = M2 + S^2 / n over the tiles in fp64 in a fixed order) instead of a pass
 * over `out`; needs `stats_ws` of rgbx_spmm_linear_stats_workspace_bytes(N, Nout) bytes (8-byte aligned). */
/* Optional cross-entropy epilogue of rgbx_spmm_linear_f32, for the model's LAST layer (models/gcn.py:29-31: the
 * logits go straight into log_softmax + NLLLoss on out[mask], itexperiments.py:400,429, or into the arg-max /
 * loss metrics of :624-626): the loss quantities come out of the 32 x Nout output tiles while they are in LDS.
 *   stats[0] = sum over selected rows of logsumexp(out_i) - out[i, y_i], stats[1] = selected rows,
 *   stats[2] = selected rows whose first arg-max equals y_i    (selection as rgbx_masked_ce_fwd_f32)
 * grad_scale == NULL: statistics only, `out` is NOT written (may be NULL) — an eval forward whose logits nobody
 * reads. grad_scale != NULL (device scalar): `out` receives grad_scale * (softmax(out_i) - onehot(y_i)) for selected
 * rows, 0 otherwise: the gradient of grad_scale * stats[0] w.r.t. the logits (rgbx_masked_ce_bwd_f32), instead of
 * the logits. scratch: (ceil(N / 32) + 64) * 3 doubles. Needs Nout <= 128; excludes out_colsums. */
typedef struct rgbx_ce_epilogue {
  const int64_t* y;        /* [N] labels */
  const uint8_t* mask;     /* [N] or NULL = all rows */
  const float* grad_scale; /* device scalar or NULL */
  double* stats;           /* [3]; [6] with mask_groups == 2 */
  double* scratch;         /* [(ceil(N / 32) + 64) * 3]; * 6 with mask_groups == 2 */
  /* 0 / 1: `mask` is a boolean. 2 (since 4.0.2; statistics only, mask required): bit 0 / bit 1 of mask[i] select row i
   * for statistics set 0 / set 1 — stats[0:3] and stats[3:6] — so that ONE eval forward serves the val and the test
   * metrics of an epoch (itexperiments.py:464-473 runs two identical forwards for them). */
  int32_t mask_groups;
} rgbx_ce_epilogue_t;

int rgbx_spmm_linear_supported(int64_t K, int64_t Nout, int has_root);
int rgbx_spmm_linear_stats_workspace_bytes(int64_t N, int64_t Nout, size_t* bytes);
int rgbx_spmm_linear_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* rs,
                         const float* x, int64_t ldx, const float* wt, const float* x_root, int64_t ldr,
                         const float* wt_root, const float* bias, float* out, int64_t ldo, float* z_out,
                         int64_t ldz, const float* pre_scale, const float* pre_shift, const float* pre_rowsum,
                         double* out_colsums, void* stats_ws, size_t stats_ws_bytes,
                         const rgbx_ce_epilogue_t* ce, int64_t N, int64_t K, int64_t Nout,
                         const rgbx_row_split_t* split, rgbx_stream_t stream);

/* The same layer with everything a NODE-PARTITIONED run needs of it (new capability: the reference is single-device,
 * itexperiments.py:246; the layer arithmetic is unchanged: models/gcn.py:27-29, graphsage.py:49-62), all arguments in
 * one struct. On top of rgbx_spmm_linear_f32:
 *  - DENSE mode (rowptr == NULL): the 32-row tile is not aggregated here but LOADED — row i of the tile is row i of
 *    `x` (after the optional pre-affine map, where pre_rowsum[i] multiplies pre_shift as above). This is the RETURN
 *    stage of the row-group x column-slice exchange: another rank aggregated column slices of this rank's rows and
 *    sent them back; transform, root term, statistics and the loss epilogue then run here without an unpack pass.
 *    It is also the plain product x * wt (e.g. dy * W of the backward pass) with any of the stores below.
 *  - BLOCKED layouts: a matrix whose columns are cut into blocks of `cols` columns, each block stored as its own
 *    contiguous [N, cols] matrix, `stride` elements from one block to the next: element (i, c) lives at
 *    base + (c / cols) * stride + i * cols + c % cols. cols == 0 means plain row-major with the leading dimension
 *    given beside it. Column slices of the rows are what the exchange sends and receives, so a producer writes its
 *    output blocked straight into the send buffer (out_blk; `out` row-major may be written as well, or be NULL) and a
 *    consumer reads the received slices in place (x in DENSE mode, x_root always). cols % 4 == 0.
 * Everything else (w, rs, root term, z_out, pre_*, out_colsums, ce, split) as rgbx_spmm_linear_f32. */
typedef struct rgbx_fused_layer {
  const int32_t* rowptr;   /* NULL = DENSE mode (col, w, rs, split ignored) */
  const int32_t* col;
  const float* w;
  const float* rs;
  const float* x;
  int64_t ldx;
  int64_t x_blk_cols, x_blk_stride;   /* DENSE mode only */
  const float* wt;         /* [K, Nout] */
  const float* x_root;     /* or NULL */
  int64_t ldr;
  int64_t xr_blk_cols, xr_blk_stride;
  const float* wt_root;
  const float* bias;
  float* out;              /* row-major [N, Nout] (ldo), or NULL */
  int64_t ldo;
  float* out_blk;          /* blocked copy of the output, or NULL */
  int64_t ob_cols, ob_stride;
  float* z_out;            /* [N, K] (ldz) or NULL */
  int64_t ldz;
  const float* pre_scale;
  const float* pre_shift;
  const float* pre_rowsum;
  double* out_colsums;
  void* stats_ws;
  size_t stats_ws_bytes;
  const rgbx_ce_epilogue_t* ce;
  int64_t N, K, Nout;
  const rgbx_row_split_t* split;
  /* Second aggregate (both NULL, or both set): z_pos_out[i,:] (ld = ldz) = sum_p w_pos[p] x[col[p],:] over the same
   * slots, from the rows the launch gathers anyway; never transformed. Single-head GAT's training forward: w = the
   * attention coefficients, w_pos = those of edges with a positive score (rgbx_gat_edge_softmax_f32), the pair
   * (z_out, z_pos_out) = (out, out_pos) of rgbx_gat_bwd_prep_f32. Needs w and z_out, K in {64, 128, 256}, Nout <= 128,
   * no rs / pre_* / out_blk; with `split`, split->partial must hold n_chunks * K + 2 * n_long * K floats. */
  const float* w_pos;
  float* z_pos_out;
} rgbx_fused_layer_t;

int rgbx_fused_layer_f32(const rgbx_fused_layer_t* layer, rgbx_stream_t stream);

/* dst[i, c] (row-major, ldd) = src (blocked as above: blk_cols, blk_stride)[i, c] (+ bias[c], bias optional) for
 * i < n, c < d: the unpack of received column slices where a consumer wants plain rows (BatchNorm's backward kernels; the
 * logits of an eval forward whose last transform ran before the exchange, + the layer's bias). */
int rgbx_blocked_to_rows_f32(const float* src, int64_t blk_cols, int64_t blk_stride, float* dst, int64_t ldd,
                             int64_t n, int64_t d, const float* bias, rgbx_stream_t stream);

/* z_0 = h;  z_{k+1} = (1-alpha) * A_hat z_k + alpha * h, k = 0..K-1; result in `out`.
 * `tmp` is an [N, d] scratch (ld = ldo); h, out, tmp must not alias. K >= 0. */
int rgbx_appnp_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* h,
                   int64_t ldh, float* out, float* tmp, int64_t ldo, int64_t N, int64_t d, int K,
                   float alpha, const rgbx_row_split_t* split, rgbx_stream_t stream);

/* ---- DAGNN: adaptive mix of the K+1 hops (reference models/dagnn.py:49-55) ---------------- */

/* Hop 0 is the layer input h0 [N, d] (ld0); hops 1..K are K matrices [N, d] (ldh) `hop_stride` floats apart,
 * written there by K calls of rgbx_spmm_csr_f32.  s [d] / b [1] = weight / bias of Prop.proj = Linear(C, 1),
 * both on the device.   retain[i,k] = sigmoid(<hop_k[i,:], s> + b);   out[i,:] = sum_k retain[i,k] * hop_k[i,:]
 * (the stack / proj / sigmoid / matmul of dagnn.py:51-54 in one pass over the hops).  d % 4 == 0, d <= 256,
 * 16-byte aligned rows. */
int rgbx_dagnn_gate_fwd_f32(const float* h0, int64_t ld0, const float* hops, int64_t hop_stride, int64_t ldh,
                            const float* s, const float* b, float* out, int64_t ldo, int64_t N, int64_t d, int K,
                            rgbx_stream_t stream);

/* Backward of the mix for gout = dL/dout [N, d]: per hop the DIRECT part of that hop's gradient,
 *   direct_k[i,:] = retain[i,k] * gout[i,:] + c[i,k] * s,   c[i,k] = <gout_i, hop_k[i]> retain (1 - retain),
 * written to d0 (hop 0) and dk + (k-1) * d_stride (hops 1..K) — the `y` operands of the Horner chain
 * G_k = A_hat^T G_{k+1} + direct_k run with rgbx_spmm_csr_f32 on the transposed CSR — and the gradients of the
 * projection, g_s[d] = sum_ik c[i,k] hop_k[i,:], g_b[1] = sum_ik c[i,k] (g_b may be NULL).  `ws`: scratch of
 * rgbx_dagnn_gate_bwd_workspace_bytes(d) bytes (per-block partial sums, added in block order). */
int rgbx_dagnn_gate_bwd_workspace_bytes(int64_t d, size_t* bytes);
int rgbx_dagnn_gate_bwd_f32(const float* h0, int64_t ld0, const float* hops, int64_t hop_stride, int64_t ldh,
                            const float* s, const float* b, const float* gout, int64_t ldg, float* d0,
                            int64_t ldd0, float* dk, int64_t d_stride, int64_t ldd, float* g_s, float* g_b,
                            void* ws, size_t ws_bytes, int64_t N, int64_t d, int K, rgbx_stream_t stream);

/* ---- GAT: fused score + edge-softmax + aggregate ------------------------------------------ */

/* a_src[n,h] = <hfeat[n,h,:], att_src[h,:]>, a_dst likewise; hfeat is [n, H*C] (ld = ldh). */
int rgbx_gat_scores_f32(const float* hfeat, int64_t ldh, const float* att_src,
                        const float* att_dst, float* a_src, float* a_dst, int64_t n, int H, int C,
                        rgbx_stream_t stream);

/* Floats of device scratch rgbx_gat_scores_bwd_f32 needs (per-workgroup partial sums). */
int rgbx_gat_scores_bwd_scratch_floats(int64_t n, int H, int C, int64_t* count);

/* Backward of rgbx_gat_scores_f32, fused with the accumulation into the feature gradient:
 *   g_hfeat[r,h,:] += g_a_src[r,h] * att_src[h,:] + g_a_dst[r,h] * att_dst[h,:]   (in place; skipped when
 *                                                  g_hfeat is NULL: rgbx_gat_bwd_src_f32 can fold it into its store)
 *   g_att_src[h,:]  = sum_r g_a_src[r,h] * hfeat[r,h,:],   g_att_dst likewise
 * over rows r in [0, n); g_a_dst has n_dst <= n rows (rows beyond are zero: halo rows of a partitioned
 * run have no target role). Partial sums per workgroup are added in workgroup order (reproducible). */
int rgbx_gat_scores_bwd_f32(const float* hfeat, int64_t ldh, const float* g_a_src, const float* g_a_dst,
                            int64_t n_dst, const float* att_src, const float* att_dst, float* g_hfeat,
                            int64_t ldgh, float* g_att_src, float* g_att_dst, float* scratch,
                            int64_t scratch_floats, int64_t n, int H, int C, rgbx_stream_t stream);

/* Forward over the target-grouped CSR. For every target i and head h:
 *   e_p   = leaky_relu(a_src[col[p],h] + a_dst[i,h], slope)
 *   alpha = exp(e_p - max_p e_p) / (sum_p exp(e_p - max) + 1e-16)
 *   out[i,h,:] = sum_p alpha_p * hfeat[col[p],h,:]
 * Saves m[i,h] = max and rden[i,h] = 1/(sum + 1e-16) for backward (both [N,H]).
 * Rows without edges produce 0 (m = 0, rden = 0).
 * Source scores: either `a_src` ([n_src, H], gathered per edge) or, when `att_src` ([H, C]) is not
 * NULL, recomputed inside the kernel from each gathered row as <hfeat[col[p],h,:], att_src[h,:]>
 * (saves one cache-line request per edge; `a_src` is then ignored and may be NULL).
 * Target scores: `a_dst` ([N, H]) is an INPUT when `att_dst` is NULL. With `att_dst` ([H, C]) the kernel forms
 * a_dst[i,h] = <hfeat[i,h,:], att_dst[h,:]> itself from the target's own row (targets are rows [0, N) of hfeat) and,
 * if `a_dst` is not NULL, stores it there for the backward: rgbx_gat_scores_f32 is then not needed at all.
 * `bias` ([H*C], optional): added to every stored row (GATConv's `out + bias` with concat=True, or heads=1);
 * pass the same pointer to rgbx_gat_bwd_prep_f32, which needs the bare aggregate.
 * `out_scale` ([H*C], optional, inference only): stored row = aggregate * out_scale + bias — an eval-mode
 * BatchNorm after the layer (models/gat.py:29) folded into the store (bias then = bias * scale + shift).
 * `out_pos` ([N, H*C], dense) and `a_pos` ([N, H]) — both NULL (inference) or both set (a forward whose backward
 * will be asked for): the part of the aggregate, and of the attention mass, carried by edges whose pre-activation
 * score a_src[j] + a_dst[i] is positive,
 *   out_pos[i,h,:] = sum_{p: s_p > 0} alpha_p * hfeat[col[p],h,:],   a_pos[i,h] = sum_{p: s_p > 0} alpha_p.
 * LeakyReLU's derivative is 1 on those edges and `slope` on the others, which turns the target-side score
 * gradient into a per-node expression (rgbx_gat_bwd_prep_f32): no per-edge tensor is written in the backward.
 * `split` (optional): hub targets are cut into chunks whose online-softmax states are merged in chunk
 * order; `split->partial` must hold n_chunks * (H*C + 2*H) floats, or n_chunks * (2*H*C + 3*H) with out_pos. */
int rgbx_gat_aggregate_fwd_f32(const int32_t* rowptr, const int32_t* col, const float* hfeat,
                               int64_t ldh, const float* a_src, const float* att_src,
                               float* a_dst, const float* att_dst, const float* out_scale, const float* bias,
                               float* out, int64_t ldo, float* m, float* rden, float* out_pos, float* a_pos,
                               int64_t N, int H, int C, float slope, const rgbx_row_split_t* split,
                               rgbx_stream_t stream);

/* Single head: the attention coefficients of every in-edge as a per-edge vector in CSR slot order,
 *   alpha[p] = exp(e_p - max) / (sum + 1e-16),  e_p = leaky_relu(a_src[col[p]] + a_dst[i], slope)   (p in row i),
 * plus m[i] / rden[i] as rgbx_gat_aggregate_fwd_f32 saves them. With ONE head the transform of a GATConv can run
 * behind the aggregation, out_i = (sum_j alpha_ij x_j) W^T + b with scores a = x (W^T att) (GATConv.forward with
 * heads = 1 [PyG], the last layer of reference models/gat.py:21,30), which is rgbx_fused_layer_f32 with w = alpha:
 * no h = x W^T product and no logits pass (the loss epilogue applies). `alpha_pos` ([E']) and `a_pos` ([N]): both NULL
 * (inference) or both set — alpha_pos[p] = alpha[p] where the pre-activation score is positive, else 0, a_pos[i] their
 * sum: rgbx_fused_layer_f32's w_pos / z_pos_out then yield the out_pos of rgbx_gat_bwd_prep_f32.
 * `split` (optional): rows with more than split->threshold slots (split->long_row, n_long) get a workgroup each
 * instead of 8 lanes; no scratch needed (split->partial is not used). */
int rgbx_gat_edge_softmax_f32(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                              float slope, float* alpha, float* alpha_pos, float* m, float* rden, float* a_pos,
                              int64_t N, const rgbx_row_split_t* split, rgbx_stream_t stream);

/* Backward, target side (same CSR as forward). Per target i, head h:
 *   dsum[i,h]    = <gout[i,h,:], out[i,h,:]>
 *   g_a_dst[i,h] = sum_p alpha_p * (<gout[i,h,:], hfeat[col[p],h,:]> - dsum[i,h]) * lrelu'(s_p)
 * `nodeq` ([N,H,4] floats, 16-byte aligned) is an output consumed by the source-side pass: the record
 * (a_dst, m - log(rden), dsum, 0) of every (target, head) — alpha = exp(e - m) * rden = exp(e - (m - log rden)) —
 * so that pass needs one 16-byte gather per edge and head instead of four 4-byte ones. */
int rgbx_gat_bwd_dst_f32(const int32_t* rowptr, const int32_t* col, const float* hfeat,
                         int64_t ldh, const float* a_src, const float* a_dst, const float* m,
                         const float* rden, const float* out, int64_t ldo, const float* gout,
                         int64_t ldg, float* nodeq, float* g_a_dst, int64_t N, int H, int C,
                         float slope, rgbx_stream_t stream);

/* The per-target record alone (no neighbour loop, one streaming pass over out / gout):
 *   nodeq[i,h] = (a_dst[i,h], m[i,h] - log(rden[i,h]), dsum = <gout[i,h,:], out[i,h,:] - bias[h,:]>, 0)
 * (`bias` = the pointer given to the forward, or NULL). With the forward's `out_pos` / `a_pos` (all three of
 * out_pos, a_pos, g_a_dst set, or all NULL) also the target-side score gradient, with no pass over the edges:
 *   g_a_dst[i,h] = sum_p alpha_p (<gout_i, h_j> - dsum) lrelu'(s_p) = (1 - slope) (<gout_i, out_pos_i> - dsum a_pos_i). */
int rgbx_gat_bwd_prep_f32(const float* a_dst, const float* m, const float* rden, const float* out,
                          int64_t ldo, const float* bias, const float* gout, int64_t ldg, float* nodeq,
                          const float* out_pos, const float* a_pos, float slope, float* g_a_dst,
                          int64_t N, int H, int C, rgbx_stream_t stream);

/* Backward, source side, over the TRANSPOSED CSR (rows = sources j, col = targets i):
 *   g_hfeat[j,h,:] = sum_{p: j->i} alpha_p * gout[i,h,:]
 *   ds_p[h]        = alpha_p * (<gout[i,h,:], hfeat[j,h,:]> - dsum[i,h]) * lrelu'(s_p)
 *   g_a_src[j,h]   = sum_{p: j->i} ds_p[h]
 * alpha is recomputed from a_src[j] and nodeq[i]. If `ds` is not NULL, ds_p is also stored at
 * ds[p, :] ([E', H] in transposed-slot order): g_a_dst[i,h] = sum of ds over the in-edges of i is then a
 * width-H segment sum (rgbx_spmm_csr_f32 over the forward rowptr with col = the forward-slot ->
 * transposed-slot map), which replaces the second full gather pass of rgbx_gat_bwd_dst_f32.
 * `a_src` NULL: the source's own score is formed from its row with att_src = att2[0], as the forward did.
 * `att2` ([2, H, C] = [att_src; att_dst], optional) with `g_a_dst` ([N, H]: the target-side score gradient of every
 * SOURCE row, zero for rows that are no target): the backward of the two score products is folded into the store,
 *   g_hfeat[j,h,:] += g_a_src[j,h] * att_src[h,:] + g_a_dst[j,h] * att_dst[h,:]
 * — rgbx_gat_scores_bwd_f32 then runs with g_hfeat = NULL (attention-vector gradients only). */
int rgbx_gat_bwd_src_f32(const int32_t* rowptr_t, const int32_t* col_t, const float* hfeat,
                         int64_t ldh, const float* a_src, const float* nodeq, const float* gout,
                         int64_t ldg, float* g_hfeat, int64_t ldgh, float* g_a_src, float* ds,
                         const float* att2, const float* g_a_dst, int64_t N, int H, int C, float slope,
                         const rgbx_row_split_t* split, rgbx_stream_t stream);

/* ---- dense layers on the MFMA units -------------------------------------------------------- */

/* Scratch bytes for rgbx_gemm_tn_f32 (split-K partial tiles + partial column sums). */
int rgbx_gemm_tn_workspace_bytes(int64_t K, int64_t M, int64_t N, size_t* bytes);

/* C[M,N] = alpha * A^T B, A [K,M] (lda), B [K,N] (ldb), fp32 row-major, K = node count (huge),
 * M,N = feature widths. Split-K over the whole chip on v_mfma_f32_32x32x2_f32 (exact fp32), partials
 * reduced in slab order (bitwise reproducible). This is dW = dY^T X of every Linear on the path.
 * Alignment: when A and B are 16-byte aligned with lda % 4 == 0 and ldb % 4 == 0, rows are read in whole 16-byte groups —
 * for M % 4 != 0 / N % 4 != 0 the last group of a row ends in the row's padding (columns [M, lda) / [N, ldb)), which
 * must therefore be readable memory for EVERY row, the last included (any finite or non-finite contents: those products
 * are never stored). F = 1433 feature rows kept at a stride of 1436 floats take this path; rows that are not 16-byte
 * aligned take a 4-byte path several times slower.
 * `a_colsum` ([M], optional): also receives the column sums of A (unscaled) — the bias gradient
 * db = sum_rows dY, taken from the A tiles the kernel stages anyway. */
int rgbx_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                     float* a_colsum, int64_t K, int64_t M, int64_t N, float alpha, void* workspace,
                     size_t workspace_bytes, rgbx_stream_t stream);

/* ---- BatchNorm1d over the node axis --------------------------------------------------------- */

/* Doubles of scratch the two column-reduction entry points need. */
int rgbx_bn_scratch_doubles(int64_t N, int64_t d, int64_t* count);

/* sums[0,c] = sum_r x[r,c], sums[1,c] = sum_r x[r,c]^2 (fp64 accumulation; [2,d] doubles, device).
 * The caller forms mean / biased variance (and, in a node-partitioned run, all-reduces `sums` first). */
int rgbx_bn_stats_f32(const float* x, int64_t ldx, int64_t N, int64_t d, double* sums, double* scratch,
                      int64_t scratch_doubles, rgbx_stream_t stream);

/* Between the raw sums and the apply pass, one launch. `packed` = [sum x (d), sum x^2 (d), row count (1)] in fp64,
 * after any cross-rank reduction. Produces mean, rstd = 1/sqrt(biased var + eps), the training forward's affine map
 * scale = weight * rstd, shift = bias - mean * scale, and (running_* not NULL) the nn.BatchNorm1d update
 * running = (1 - momentum) * running + momentum * (mean | unbiased var). */
int rgbx_bn_finalize_f32(const double* packed, const float* weight, const float* bias, float eps, float momentum,
                         float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                         float* shift, int64_t d, rgbx_stream_t stream);

/* Eval forwards: the BatchNorm1d BEHIND a linear layer in eval mode (running statistics; models/gcn.py:27-28 under
 * model.eval(), itexperiments.py:619-622) is a per-column affine map of that layer's output, i.e. the same layer with
 * rescaled weights. One launch makes the operands the fused kernels read:
 *   scale[n] = gamma[n] / sqrt(running_var[n] + eps),  shift[n] = beta[n] - running_mean[n] * scale[n]
 *   wt[k, n]  = W[n, k]  * scale[n]      (W [Nout, K] row-major -> W'^T [K, Nout], the [K, Nout] operand layout)
 *   wrt[k, n] = Wr[n, k] * scale[n]      (optional second weight: the root term of SAGEConv / my_SAGEConv)
 *   b_out[n]  = (bias[n] + bias2[n]) * scale[n] + shift[n]      (bias, bias2, b_out optional)
 * gamma == beta == running_mean == running_var == NULL: no BatchNorm (scale 1, shift 0), i.e. the plain transpose
 * of the model's last layer. Replaces ~8 [out]-sized vector launches and two transposes per layer and eval forward. */
int rgbx_fold_bn_linear_f32(const float* W, int64_t ldw, const float* Wr, int64_t ldwr, const float* bias,
                            const float* bias2, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* wt, float* wrt, float* b_out, int64_t Nout,
                            int64_t K, rgbx_stream_t stream);

/* Backward counterpart: ca = glob[0]/n, cb = glob[1]/n, ck = weight * rstd for rgbx_bn_bwd_apply_f32 (`glob` = the
 * [2, d] sums of rgbx_bn_bwd_reduce_f32 after any cross-rank reduction, `count` = device pointer to n), and the
 * parameter gradients of this rank, g_bias = local[0], g_weight = local[1]. */
int rgbx_bn_bwd_finalize_f32(const double* glob, const double* local, const double* count, const float* weight,
                             const float* rstd, float* ca, float* cb, float* ck, float* g_weight, float* g_bias,
                             int64_t d, rgbx_stream_t stream);

/* y[r,c] = x[r,c] * scale[c] + shift[c] — BatchNorm's normalise+affine with
 * scale = gamma * rstd, shift = beta - mean * scale (training or running statistics alike). */
int rgbx_affine_cols_f32(const float* x, int64_t ldx, const float* scale, const float* shift, float* y,
                         int64_t ldy, int64_t N, int64_t d, rgbx_stream_t stream);

/* sums[0,c] = sum_r gy[r,c], sums[1,c] = sum_r gy[r,c] * xhat[r,c], xhat = (x - mean) * rstd. */
int rgbx_bn_bwd_reduce_f32(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* mean,
                           const float* rstd, int64_t N, int64_t d, double* sums, double* scratch,
                           int64_t scratch_doubles, rgbx_stream_t stream);

/* gx[r,c] = (gy[r,c] - ca[c] - xhat[r,c] * cb[c]) * ck[c]
 * (ca = sum gy / n, cb = sum gy*xhat / n, ck = gamma * rstd). */
int rgbx_bn_bwd_apply_f32(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* mean,
                          const float* rstd, const float* ca, const float* cb, const float* ck, float* gx,
                          int64_t ldgx, int64_t N, int64_t d, rgbx_stream_t stream);

/* ---- loss / metrics on masked rows ---------------------------------------------------------- */

/* Doubles of device scratch rgbx_masked_nll_fwd_f32 needs (3 per workgroup). */
int rgbx_masked_nll_scratch_doubles(int64_t N, int want_accuracy, int64_t* count);

/* stats[0] = sum over selected rows of -logp[i, y[i]]; stats[1] = number of selected rows;
 * stats[2] = number of selected rows whose first arg-max equals y[i] (only if want_accuracy).
 * A row is selected when mask == NULL or mask[i] != 0, and 0 <= y[i] < C. `stats` = 3 doubles (device).
 * Per-workgroup sums go to `scratch` and are added in workgroup order (reproducible; no atomics).
 * logp is [N, C] (ld). */
int rgbx_masked_nll_fwd_f32(const float* logp, int64_t ld, const int64_t* y, const uint8_t* mask,
                            int64_t N, int64_t C, double* stats, double* scratch,
                            int64_t scratch_doubles, int want_accuracy, rgbx_stream_t stream);

/* grad[i,c] = -scale[0] if row i is selected and c == y[i], else 0, for every (i, c): the gradient
 * of scale * stats[0] w.r.t. logp. `scale` is a device scalar (no host sync). */
int rgbx_masked_nll_bwd_f32(const int64_t* y, const uint8_t* mask, int64_t N, int64_t C,
                            const float* scale, float* grad, int64_t ldg, rgbx_stream_t stream);

/* The same loss taken from the LOGITS (cross-entropy = NLLLoss(log_softmax(z)), models/gcn.py:31 +
 * itexperiments.py:400,429): stats as rgbx_masked_nll_fwd_f32 with -logp[i, y_i] = logsumexp(z_i) - z[i, y_i]
 * and the arg-max taken on z (same as on log_softmax(z)); log-softmax is never written out.
 * Scratch: rgbx_masked_nll_scratch_doubles(N, 1). */
int rgbx_masked_ce_fwd_f32(const float* logits, int64_t ld, const int64_t* y, const uint8_t* mask, int64_t N,
                           int64_t C, double* stats, double* scratch, int64_t scratch_doubles,
                           rgbx_stream_t stream);

/* The same statistics from BLOCKED logits (+ bias[c], optional): element (i, c) at logits + (c / blk_cols) * blk_stride
 * + i * blk_cols + c % blk_cols, the layout a node-partitioned run's exchange delivers (rgbx_fused_layer_t) — the eval
 * forward's loss straight from the received column slices (new capability: the reference is single-device,
 * itexperiments.py:246; the arithmetic is itexperiments.py:624-626 on models/gcn.py:29-31). Only the selected rows are
 * read. C % 4 == 0, C <= 256, blk_cols % 4 == 0 dividing C. */
int rgbx_masked_ce_fwd_blocked_f32(const float* logits, int64_t blk_cols, int64_t blk_stride, const float* bias,
                                   const int64_t* y, const uint8_t* mask, int64_t N, int64_t C, double* stats,
                                   double* scratch, int64_t scratch_doubles, rgbx_stream_t stream);

/* grad[i,c] = scale[0] * (softmax(z_i)[c] - [c == y[i]]) for selected rows, 0 otherwise: the gradient of
 * scale * stats[0] w.r.t. the logits, one pass. `scale` is a device scalar. */
int rgbx_masked_ce_bwd_f32(const float* logits, int64_t ld, const int64_t* y, const uint8_t* mask, int64_t N,
                           int64_t C, const float* scale, float* grad, int64_t ldg, rgbx_stream_t stream);

/* ---- edge-list ingest (before the path: rd2pd.py:92-93, itexperiments.py:235-238) ------------ */

/* Sorted set of the pairs (row[e], col[e]) — with `mirror` != 0 also of their reverses (col[e], row[e]): coalesce and
 * to_undirected. Two steps because the caller owns the output and its size is only known on the device:
 *   1. rgbx_coalesce_keys_i64 writes the distinct keys row * N + col in ascending order to keys_out (capacity
 *      M = mirror ? 2E : E) and counts[0] = their number, counts[1] = number of endpoints outside [0, N) (those are
 *      clamped; a caller that sees counts[1] != 0 must discard the result). counts: 2 x uint64 on the device.
 *   2. after reading counts[0] back and allocating int64 [2, count], rgbx_split_edge_keys_i64 writes the rows
 *      (cap = the caller's capacity; min(cap, counts[0]) pairs are written).
 * Workspace: rgbx_coalesce_workspace_bytes(E, N, mirror). Self-loops are kept (once), as both PyG helpers do. */
int rgbx_coalesce_workspace_bytes(int64_t E, int64_t N, int mirror, size_t* bytes);
int rgbx_coalesce_keys_i64(const int64_t* row, const int64_t* col, int64_t E, int64_t N, int mirror,
                           uint64_t* keys_out, uint64_t* counts, void* workspace, size_t workspace_bytes,
                           rgbx_stream_t stream);
int rgbx_split_edge_keys_i64(const uint64_t* keys, const uint64_t* counts, int64_t cap, int64_t N, int64_t* out_row,
                             int64_t* out_col, rgbx_stream_t stream);

/* HOST function (the one entry point that takes a host pointer and runs on the CPU, synchronously): perm[0..n) =
 * list(range(n)) after Python's `random.seed(seed); random.shuffle(...)`, bit for bit (MT19937 seeded by init_by_array over
 * the 32-bit words of |seed|; randbelow by rejection over getrandbits(bit_length)) — the shuffle behind the reference's split
 * masks (utils/mask.py:66-102; itexperiments.py:210-215). n < 2^32. */
int rgbx_py_random_shuffle_i64(int64_t seed, int64_t n, int64_t* perm);

/* ---- halo pack / unpack (multi-GPU node partition) ----------------------------------------- */

/* dst[r,:] = src[idx[r],:] for r in [0,n) — pack boundary rows into a send buffer. */
int rgbx_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int64_t n, int64_t d,
                         float* dst, int64_t ldd, rgbx_stream_t stream);

/* dst[idx[r],:] += src[r,:] for r in [0,n); idx entries must be unique (no atomics). */
int rgbx_scatter_add_rows_f32(const float* src, int64_t lds, const int32_t* idx, int64_t n,
                              int64_t d, float* dst, int64_t ldd, rgbx_stream_t stream);

/* Measurement aid, not on the path: dst[0:n] = src[0:n] by `workgroups` workgroups only, i.e. at a rate the caller
 * calibrates — the stand-in for an exchange's memory traffic when ONE GPU emulates a rank of a partitioned job
 * (bench.py --emulate-rank P --emulate-contend GBS). nontemporal != 0: loads and stores that do not allocate in the caches
 * (the optimistic bracket of what a real exchange's DMA traffic does to them). n % 4 == 0, 16-byte aligned pointers. */
int rgbx_paced_copy_f32(const float* src, float* dst, int64_t n, int workgroups, int nontemporal, rgbx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RGBX_HIP_H */
