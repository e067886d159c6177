"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product.

A plain-torch (CPU, fp32, edge-parallel gather -> multiply -> index_add_) restatement of the
message-passing path the reference delegates to torch_geometric. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module, and
only as the checker / the timed CPU baseline; ``rgb_experiment_amd`` never imports it.

PINNING STATUS:
  * ``gcn_norm`` / ``propagate`` / ``appnp`` are pinned against golden vectors produced by the
    reference's own PyG-free code run in the build container (tests/golden/make_golden.py):
    G1 ``normalize_adj`` (itexperiments.py:677-684), G2 ``label_propagation`` (:698-719),
    G3 ``PTA.inference`` (models/pta.py:79-84).
  * ``gat_conv`` and ``sage_conv`` restate PyG's documented GATConv / SAGEConv formulas; the
    reference holds no in-repo statement, test or fixture for them and PyG is not installable here:
    **parity unpinned** for those two (checked only against hand-derived known answers).

Every function cites the reference file:line it follows. `edge_index` is int64 [2, E] with
row 0 = source j and row 1 = target i (the aggregation index), as in PyG.
"""
import torch


# ---- self-loop rewrites ---------------------------------------------------------------------

def _select_columns(edge_index, keep):
    """edge_index[:, keep] (row by row: a boolean mask along dim 1 of a [2, 60 M] tensor takes 6 s, two 1-D selections 1 s)."""
    return torch.stack([edge_index[0][keep], edge_index[1][keep]])


def remove_self_loops(edge_index, edge_weight=None):
    """PyG remove_self_loops as called at models/graphsage.py:55."""
    keep = edge_index[0] != edge_index[1]
    return _select_columns(edge_index, keep), (None if edge_weight is None else edge_weight[keep])


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    """PyG add_self_loops as called at models/graphsage.py:56: append (i, i) for every node."""
    n = int(num_nodes)
    loops = torch.arange(n, dtype=edge_index.dtype).unsqueeze(0).repeat(2, 1)
    ei = torch.cat([edge_index, loops], dim=1)
    if edge_weight is not None:
        edge_weight = torch.cat([edge_weight, edge_weight.new_full((n,), fill_value)])
    return ei, edge_weight


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    """PyG add_remaining_self_loops as called at models/dagnn.py:21-22: non-loop edges keep their
    order; every node then gets exactly one loop whose weight is the existing loop's weight if it had
    one (the last one, if several), else `fill_value`."""
    n = int(num_nodes)
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop_w = None
    if edge_weight is not None:
        loop_w = edge_weight.new_full((n,), fill_value)
        loop_w[row[~keep]] = edge_weight[~keep]
        edge_weight = torch.cat([edge_weight[keep], loop_w])
    loops = torch.arange(n, dtype=edge_index.dtype).unsqueeze(0).repeat(2, 1)
    return torch.cat([_select_columns(edge_index, keep), loops], dim=1), edge_weight


def rewrite_edges(edge_index, num_nodes, loops_mode):
    """Edge list a conv layer aggregates over, plus for every rewritten edge the id of the input
    edge it came from (e in [0,E); E + i for the self-loop added for node i).
    loops_mode: 0 keep (SAGEConv), 1 add_remaining_self_loops (GCNConv/APPNP), 2 remove+add
    (my_SAGEConv, GATConv)."""
    E = edge_index.size(1)
    ids = torch.arange(E, dtype=torch.int64)
    if loops_mode == 0:
        return edge_index, ids
    keep = edge_index[0] != edge_index[1]
    if loops_mode == 1:
        ei, _ = add_remaining_self_loops(edge_index, None, 1.0, num_nodes)
    elif loops_mode == 2:
        ei, _ = remove_self_loops(edge_index)
        ei, _ = add_self_loops(ei, num_nodes=num_nodes)
    else:
        raise ValueError(loops_mode)
    return ei, torch.cat([ids[keep], E + torch.arange(num_nodes, dtype=torch.int64)])


def csr_from_edges(agg_index, other_index, edge_ids, num_nodes):
    """Group a (rewritten) edge list by `agg_index` with a STABLE sort: the index bookkeeping
    propagate does implicitly. Returns int32 rowptr [N+1], col [E'], perm [E'] — the bit-exact
    expectation for rgbx_csr_build."""
    if agg_index.numel() >= (1 << 20) and _have_c_lib():
        # the same stable grouping as a counting sort in C (oracle_stable_group_i64; equality with the torch.sort form:
        # tests/test_oracle_golden.py) — the BASELINE-size checkers group 62 M keys eight times
        agg = agg_index.contiguous()
        rowptr = torch.empty(num_nodes + 1, dtype=torch.int64)
        order = torch.empty(agg.numel(), dtype=torch.int64)
        _c_lib().oracle_stable_group_i64(agg.data_ptr(), agg.numel(), int(num_nodes), rowptr.data_ptr(), order.data_ptr())
    else:
        order = torch.sort(agg_index, stable=True)[1]
        counts = torch.bincount(agg_index, minlength=num_nodes)
        rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), other_index[order].to(torch.int32), edge_ids[order].to(torch.int32)


# ---- gcn_norm + propagate ---------------------------------------------------------------------

def gcn_norm(edge_index, edge_weight=None, num_nodes=None, add_loops=True, dtype=torch.float32):
    """models/dagnn.py:12-31 line by line (improved=False -> fill 1)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype)
    if add_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, 1.0, n)
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(n, dtype=edge_weight.dtype).index_add_(0, col, edge_weight)  # scatter_add over col
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0)
    return edge_index, dis[row] * edge_weight * dis[col]


def propagate(edge_index, x, num_nodes, edge_weight=None, aggr="add"):
    """MessagePassing.propagate: x_j = x[edge_index[0]]; message = norm.view(-1,1) * x_j
    (models/dagnn.py:57-59); aggregate at edge_index[1] by 'add' (dagnn.py:36) or 'mean'
    (graphsage.py:39; mean = sum / max(count, 1)). Materialises [E, d] like the PyG path."""
    src, dst = edge_index[0], edge_index[1]
    msg = x.index_select(0, src)
    if edge_weight is not None:
        msg = edge_weight.view(-1, 1) * msg
    out = torch.zeros((num_nodes, x.size(1)), dtype=x.dtype).index_add_(0, dst, msg)
    if aggr == "mean":
        cnt = torch.bincount(dst, minlength=num_nodes).clamp(min=1).to(x.dtype)
        out = out / cnt.view(-1, 1)
    elif aggr != "add":
        raise ValueError(aggr)
    return out


def gcn_dense_adj(edge_index, num_nodes):
    """Dense A_hat with A_hat[i, j] = weight of edge j -> i (so that A_hat @ x == propagate)."""
    ei, w = gcn_norm(edge_index, None, num_nodes)
    a = torch.zeros(num_nodes, num_nodes, dtype=torch.float32)
    a.index_put_((ei[1], ei[0]), w, accumulate=True)
    return a


# ---- conv layers ------------------------------------------------------------------------------

def gcn_conv(x, edge_index, weight, bias):
    """GCNConv.forward [PyG] behind models/gcn.py:27: gcn_norm, x @ W^T, propagate(add), + bias."""
    n = x.size(0)
    ei, w = gcn_norm(edge_index, None, n)
    out = propagate(ei, x @ weight.t(), n, w, "add")
    return out if bias is None else out + bias


def my_sage_conv(x, edge_index, w_l, b_l, w_r, b_r):
    """models/graphsage.py:49-62."""
    n = x.size(0)
    x_l = x @ w_l.t() + b_l
    x_r = x @ w_r.t() + b_r
    ei, _ = remove_self_loops(edge_index)
    ei, _ = add_self_loops(ei, num_nodes=n)
    return propagate(ei, x_l, n, None, "mean") + x_r


def sage_conv(x, edge_index, w_l, b_l, w_r):
    """SAGEConv.forward [PyG] behind models/graphsage2.py:29: lin_l(mean_j x_j) + lin_r(x_i).
    PARITY UNPINNED (no in-repo statement)."""
    n = x.size(0)
    return propagate(edge_index, x, n, None, "mean") @ w_l.t() + b_l + x @ w_r.t()


def segment_softmax(e, index, num_nodes):
    """torch_geometric.utils.softmax [PyG]: exp(e - max_i) / (sum_i + 1e-16) per target segment."""
    H = e.size(1)
    mx = torch.full((num_nodes, H), float("-inf"), dtype=e.dtype)
    mx = mx.scatter_reduce(0, index.view(-1, 1).expand(-1, H), e, reduce="amax", include_self=True)
    ex = (e - mx[index]).exp()
    den = torch.zeros((num_nodes, H), dtype=e.dtype).index_add_(0, index, ex)
    return ex / (den[index] + 1e-16)


def gat_conv(x, edge_index, weight, att_src, att_dst, bias, heads, concat=True, negative_slope=0.2):
    """GATConv.forward/message [PyG] behind models/gat.py:28,30. PARITY UNPINNED."""
    n = x.size(0)
    H = heads
    C = weight.size(0) // H
    h = (x @ weight.t()).view(n, H, C)
    a_s = (h * att_src.view(1, H, C)).sum(-1)
    a_d = (h * att_dst.view(1, H, C)).sum(-1)
    ei, _ = remove_self_loops(edge_index)
    ei, _ = add_self_loops(ei, num_nodes=n)
    src, dst = ei[0], ei[1]
    e = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], negative_slope)
    alpha = segment_softmax(e, dst, n)
    msg = h[src] * alpha.unsqueeze(-1)
    out = torch.zeros((n, H, C), dtype=x.dtype).index_add_(0, dst, msg)
    out = out.reshape(n, H * C) if concat else out.mean(dim=1)
    return out if bias is None else out + bias


def appnp(x, edge_index, K, alpha):
    """APPNP.forward [PyG] behind models/appnp_stack.py:29, recurrence as models/pta.py:79-84:
    gcn_norm once, then z <- (1-alpha) * A_hat z + alpha * x, K times."""
    n = x.size(0)
    ei, w = gcn_norm(edge_index, None, n)
    z = x
    for _ in range(K):
        z = (1 - alpha) * propagate(ei, z, n, w, "add") + alpha * x
    return z


# ---- model forwards from a product state_dict ---------------------------------------------------

def batch_norm(x, sd, prefix, training, eps=1e-5):
    """nn.BatchNorm1d over the node axis (models/gcn.py:23,28): batch statistics (biased variance)
    in training, running statistics in eval."""
    if training:
        mean, var = x.mean(0), x.var(0, unbiased=False)
    else:
        mean, var = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    return (x - mean) / torch.sqrt(var + eps) * sd[prefix + "weight"] + sd[prefix + "bias"]


def _finish(x):
    return {"out": torch.log_softmax(x, dim=1), "emb": x}


def _stack(sd, x, edge_index, num_layers, training, conv):
    for i in range(num_layers - 1):
        x = batch_norm(conv(i, x), sd, f"bns.{i}.", training)
    return _finish(conv(num_layers - 1, x))


def gcn_forward(sd, x, edge_index, num_layers, training=False):
    """models/gcn.py:25-31."""
    return _stack(sd, x, edge_index, num_layers, training,
                  lambda i, v: gcn_conv(v, edge_index, sd[f"convs.{i}.lin.weight"], sd[f"convs.{i}.bias"]))


def graphsage_forward(sd, x, edge_index, num_layers, training=False):
    """models/graphsage.py:26-32."""
    return _stack(sd, x, edge_index, num_layers, training,
                  lambda i, v: my_sage_conv(v, edge_index, sd[f"convs.{i}.lin_l.weight"], sd[f"convs.{i}.lin_l.bias"],
                                            sd[f"convs.{i}.lin_r.weight"], sd[f"convs.{i}.lin_r.bias"]))


def graphsage2_forward(sd, x, edge_index, num_layers, training=False):
    """models/graphsage2.py:27-33."""
    return _stack(sd, x, edge_index, num_layers, training,
                  lambda i, v: sage_conv(v, edge_index, sd[f"convs.{i}.lin_l.weight"], sd[f"convs.{i}.lin_l.bias"],
                                         sd[f"convs.{i}.lin_r.weight"]))


def gat_forward(sd, x, edge_index, num_layers, heads, training=False):
    """models/gat.py:26-32: hidden layers `heads` heads concatenated, last layer 1 head, concat=False."""
    def conv(i, v):
        last = i == num_layers - 1
        return gat_conv(v, edge_index, sd[f"convs.{i}.lin_src.weight"], sd[f"convs.{i}.att_src"],
                        sd[f"convs.{i}.att_dst"], sd[f"convs.{i}.bias"], 1 if last else heads, concat=not last)
    return _stack(sd, x, edge_index, num_layers, training, conv)


def appnp_stack_forward(sd, x, edge_index, K, alpha, training=False):
    """models/appnp_stack.py:25-31."""
    h = x @ sd["lin1.weight"].t() + sd["lin1.bias"]
    h = batch_norm(h, sd, "bn.", training)
    h = h @ sd["lin2.weight"].t() + sd["lin2.bias"]
    return _finish(appnp(h, edge_index, K, alpha))


# ---- "next" rows: models that reuse the same propagate --------------------------------

def pta_norm_adj_dense(edge_index, num_nodes):
    """Dense D^-1/2 (A + I) D^-1/2 exactly as itexperiments.py:354-356,671-684: A[src, dst] += 1 per edge
    (coo_matrix sums duplicates), + identity (an existing self-loop ends up with 2), D = row sums."""
    a = torch.zeros(num_nodes, num_nodes, dtype=torch.float64)
    a.index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.size(1), dtype=torch.float64), accumulate=True)
    a = a + torch.eye(num_nodes, dtype=torch.float64)
    r = a.sum(1).pow(-0.5)
    r[torch.isinf(r)] = 0
    return (r.view(-1, 1) * a * r.view(1, -1)).float()


def label_propagation(adj, labels, idx, K, alpha):
    """itexperiments.py:698-719 with a dense adj."""
    c = int(labels.max()) + 1
    y0 = torch.zeros(labels.size(0), c)
    y0[idx, labels[idx]] = 1.0
    onehot = torch.nn.functional.one_hot(labels.clamp(min=0), c).float()
    y = y0
    for _ in range(K):
        y = adj @ y
        y[idx] = onehot[idx]
        y = (1 - alpha) * y + alpha * y0
    return y


def pta_inference(h, adj, K, alpha):
    """models/pta.py:79-84 with a dense adj."""
    y0 = torch.softmax(h, dim=-1)
    y = y0
    for _ in range(K):
        y = (1 - alpha) * (adj @ y) + alpha * y0
    return y


def sgc_forward(sd, x, edge_index, K, add_loops=True):
    """models/sgc.py:12-14 -> SGConv [PyG]: lin(A_hat^K x), gcn_norm with or without added self-loops."""
    n = x.size(0)
    ei, w = gcn_norm(edge_index, None, n, add_loops=add_loops)
    for _ in range(K):
        x = propagate(ei, x, n, w, "add")
    return _finish(x @ sd["conv1.lin.weight"].t() + sd["conv1.lin.bias"])


def _gin_block(sd, prefix, x, training):
    h = torch.relu(x @ sd[prefix + "0.weight"].t() + sd[prefix + "0.bias"])
    h = torch.relu(h @ sd[prefix + "2.weight"].t() + sd[prefix + "2.bias"])
    return batch_norm(h, sd, prefix + "4.", training)


def gin_forward(sd, x, edge_index, num_layers, training=False):
    """models/gin.py:48-54 -> GINConv [PyG]: nn((1 + eps) x_i + sum_j x_j); dropout omitted (p = 0 / eval)."""
    n = x.size(0)
    for prefix in ["conv1."] + [f"convs.{i}." for i in range(num_layers - 1)]:
        agg = propagate(edge_index, x, n, None, "add") + (1 + sd[prefix + "eps"]) * x
        x = _gin_block(sd, prefix + "nn.", agg, training)
    x = torch.relu(x @ sd["lin1.weight"].t() + sd["lin1.bias"])
    return _finish(x @ sd["lin2.weight"].t() + sd["lin2.bias"])


def dagnn_forward(sd, x, edge_index, K):
    """models/dagnn.py:41-55,79-86 (dropout omitted: p = 0 / eval)."""
    n = x.size(0)
    x = torch.relu(x @ sd["lin1.weight"].t() + sd["lin1.bias"])
    x = x @ sd["lin2.weight"].t() + sd["lin2.bias"]
    ei, w = gcn_norm(edge_index, None, n)
    preds = [x]
    for _ in range(K):
        x = propagate(ei, x, n, w, "add")
        preds.append(x)
    pps = torch.stack(preds, dim=1)
    retain = torch.sigmoid(pps @ sd["prop.proj.weight"].t() + sd["prop.proj.bias"]).squeeze(-1)
    return _finish(torch.matmul(retain.unsqueeze(1), pps).squeeze(1))


def label_propagation_pyg(y, edge_index, num_layers, alpha, post_step=None):
    """torch_geometric.nn.LabelPropagation.forward [PyG] as used by CorrectAndSmooth: gcn_norm WITHOUT
    added self-loops, out <- alpha * propagate(out) + (1 - alpha) * y, then post_step (default clamp to
    [0, 1]). PARITY UNPINNED (no in-repo statement)."""
    post_step = post_step or (lambda t: t.clamp_(0.0, 1.0))
    n = y.size(0)
    ei, w = gcn_norm(edge_index, None, n, add_loops=False)
    out = y
    res = (1 - alpha) * out
    for _ in range(num_layers):
        out = propagate(ei, out, n, w, "add")
        out = post_step(out * alpha + res)
    return out


def correct_and_smooth(y_soft, y_train, mask, edge_index, num_correction_layers, correction_alpha,
                       num_smoothing_layers, smoothing_alpha, autoscale=True, scale=1.0):
    """CorrectAndSmooth.correct + .smooth [PyG] behind itexperiments.py:520-526. PARITY UNPINNED."""
    c = y_soft.size(1)
    onehot = torch.nn.functional.one_hot(y_train, c).to(y_soft.dtype)
    numel = int(mask.sum())
    error = torch.zeros_like(y_soft)
    error[mask] = onehot - y_soft[mask]
    if autoscale:
        sm = label_propagation_pyg(error, edge_index, num_correction_layers, correction_alpha,
                                   lambda t: t.clamp_(-1.0, 1.0))
        sigma = error[mask].abs().sum() / numel
        sc = sigma / sm.abs().sum(dim=1, keepdim=True)
        sc[sc.isinf() | (sc > 1000)] = 1.0
        y = y_soft + sc * sm
    else:
        def fix(t):
            t[mask] = error[mask]
            return t
        y = y_soft + scale * label_propagation_pyg(error, edge_index, num_correction_layers, correction_alpha, fix)
    y = y.clone()
    y[mask] = onehot
    return label_propagation_pyg(y, edge_index, num_smoothing_layers, smoothing_alpha)


# ---- C restatement (oracle/propagate_ref.c) ------------------------------------------------------------

def _have_c_lib():
    import os
    return os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_ref.so"))


_C_LIB = []


def _c_lib():
    import ctypes
    import os
    if _C_LIB:
        return _C_LIB[0]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_ref.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is not built (make -C oracle)")
    lib = ctypes.CDLL(path)
    P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.oracle_propagate_coo_f32.restype = None
    lib.oracle_propagate_coo_f32.argtypes = [P, P, P, I64, P, I64, P, I64, I64, I64, I]
    lib.oracle_propagate_csr_f32.restype = None
    lib.oracle_propagate_csr_f32.argtypes = [P, P, P, P, I64, P, I64, I64, I64, I, I]
    lib.oracle_max_threads.restype = I
    F = ctypes.c_float
    lib.oracle_gat_forward_csr_f32.restype = None
    lib.oracle_gat_forward_csr_f32.argtypes = [P, P, P, P, P, F, P, P, I64, I64, I64, I]
    lib.oracle_gat_backward_dst_csr_f32.restype = None
    lib.oracle_gat_backward_dst_csr_f32.argtypes = [P, P, P, P, P, F, P, P, P, P, I64, I64, I64, I]
    lib.oracle_gat_backward_src_csr_f32.restype = None
    lib.oracle_gat_backward_src_csr_f32.argtypes = [P, P, P, P, P, P, P, P, I64, I64, I64, I]
    lib.oracle_stable_group_i64.restype = None
    lib.oracle_stable_group_i64.argtypes = [P, I64, I64, P, P]
    _C_LIB.append(lib)
    return lib


def propagate_c_coo(edge_index, x, num_nodes, edge_weight=None, aggr="add"):
    """oracle_propagate_coo_f32: edge-order scatter-add in C (same arithmetic as `propagate`)."""
    lib = _c_lib()
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    x = x.contiguous().float()
    w = None if edge_weight is None else edge_weight.contiguous().float()
    out = torch.empty((num_nodes, x.size(1)), dtype=torch.float32)
    lib.oracle_propagate_coo_f32(src.data_ptr(), dst.data_ptr(), 0 if w is None else w.data_ptr(), src.numel(),
                                 x.data_ptr(), x.size(1), out.data_ptr(), x.size(1), num_nodes, x.size(1),
                                 int(aggr == "mean"))
    return out


def propagate_c_csr(rowptr, col, w, x, aggr="add", threads=0):
    """oracle_propagate_csr_f32: per-target sums over a CSR (int32 rowptr / col), OpenMP over rows."""
    lib = _c_lib()
    x = x.contiguous().float()
    n = rowptr.numel() - 1
    out = torch.empty((n, x.size(1)), dtype=torch.float32)
    lib.oracle_propagate_csr_f32(rowptr.data_ptr(), col.data_ptr(), 0 if w is None else w.data_ptr(), x.data_ptr(),
                                 x.size(1), out.data_ptr(), x.size(1), n, x.size(1), int(aggr == "mean"), threads)
    return out


def c_threads():
    return _c_lib().oracle_max_threads()
