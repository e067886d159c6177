"""CPU ORACLE for graphs too large for the PyG dataflow — TEST INFRASTRUCTURE ONLY. Not part of the product.

`oracle.ref_cpu` materialises [E', d] per propagate (32 GB per temporary at |E| = 60 M, d = 128) and lets autograd
keep them; at the BASELINE sizes that does not fit a host. This module states the SAME model forwards with the
propagate step computed by the C restatement (oracle/propagate_ref.c, `oracle_propagate_csr_f32`: per-target sums
over a CSR) and its adjoint by the same function over the TRANSPOSED CSR, wrapped in one torch.autograd.Function;
everything dense (Linear, BatchNorm from `ref_cpu.batch_norm`, log-softmax, the loss) stays plain CPU torch under
autograd. The conv layers keep `ref_cpu`'s operation order (GCNConv transforms then propagates; SAGEConv aggregates
then transforms), so they differ from the HIP path's aggregate-first fusion.

Pinned to `ref_cpu` (which the golden vectors pin): tests/test_oracle_large.py compares logits and every parameter
gradient of both statements on graphs `ref_cpu` can hold.

References: models/dagnn.py:12-31,57-59 (gcn_norm + message), models/graphsage.py:49-62 (my_SAGEConv),
models/graphsage2.py:29 -> SAGEConv [PyG, unpinned], models/pta.py:79-84 / appnp_stack.py:25-31 (APPNP),
models/gcn.py:25-31 (stack wiring)."""
import torch

from . import ref_cpu as O


class CsrGraph:
    """Forward CSR (grouped by target) and transposed CSR (grouped by source) of one rewritten edge list, with the
    per-slot weights of the aggregation and of its adjoint.
    kind 'gcn'  : add_remaining_self_loops + symmetric normalisation (ref_cpu.gcn_norm), aggr = add
    kind 'mean' : loops_mode 2 (my_SAGEConv: remove + add self-loops) or 0 (SAGEConv: edges as given), aggr = mean =
                  sum / max(count, 1); the adjoint scatters grad[i] / max(count_i, 1) back along every edge.
    kind 'gat'  : loops_mode 2 (GATConv: remove + add self-loops), no fixed weights: the attention coefficients are
                  functions of the features (gat_aggregate); keeps fwd_slot, the forward slot of every transposed slot."""

    def __init__(self, edge_index, num_nodes, kind, loops_mode=1, threads=0):
        n = int(num_nodes)
        self.n, self.kind, self.threads = n, kind, threads
        if kind == "gcn":
            ei, w = O.gcn_norm(edge_index, None, n)
        elif kind in ("mean", "gat"):
            ei, _ = O.rewrite_edges(edge_index, n, 2 if kind == "gat" else loops_mode)
            w = None
        else:
            raise ValueError(kind)
        ids = torch.arange(ei.size(1))
        self.rowptr, self.col, perm = O.csr_from_edges(ei[1], ei[0], ids, n)
        self.rowptr_t, self.col_t, perm_t = O.csr_from_edges(ei[0], ei[1], ids, n)
        self._perm, self._perm_t, self._loops_mode = perm, perm_t, (1 if kind == "gcn" else 2 if kind == "gat" else loops_mode)
        if kind == "gcn":
            self.w = w[perm.long()].contiguous()
            self.w_t = w[perm_t.long()].contiguous()
        elif kind == "gat":
            self.w = self.w_t = None
            inv = torch.empty(ei.size(1), dtype=torch.int64)
            inv[perm.long()] = torch.arange(ei.size(1))
            self.fwd_slot = inv[perm_t.long()].contiguous()  # transposed slot -> forward slot of the same edge
        else:
            cnt = torch.bincount(ei[1], minlength=n).clamp(min=1).to(torch.float32)
            self.w = None  # the C function's own mean
            self.w_t = (1.0 / cnt)[self.col_t.long()].contiguous()  # slot of source j holds target i: 1 / count_i
        self.nnz = int(ei.size(1))

    def as_gat(self):
        """The 'gat' graph over the same rewritten edge list (a 'mean' graph with loops_mode 2: GATConv and my_SAGEConv both
        remove + add self-loops): the CSRs are shared, only fwd_slot is added — two stable groupings of 62 M keys saved."""
        if self.kind != "mean" or self._loops_mode != 2:
            raise ValueError("as_gat: needs a 'mean' graph built with loops_mode 2")
        import copy
        g = copy.copy(self)
        g.kind, g.w, g.w_t = "gat", None, None
        inv = torch.empty(self.nnz, dtype=torch.int64)
        inv[self._perm.long()] = torch.arange(self.nnz)
        g.fwd_slot = inv[self._perm_t.long()].contiguous()
        return g

    def forward(self, x):
        return O.propagate_c_csr(self.rowptr, self.col, self.w, x, "add" if self.kind == "gcn" else "mean", self.threads)

    def adjoint(self, g):
        return O.propagate_c_csr(self.rowptr_t, self.col_t, self.w_t, g, "add", self.threads)


class _Propagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return graph.forward(x.detach())

    @staticmethod
    def backward(ctx, g):
        return ctx.graph.adjoint(g.contiguous()), None


def propagate(x, graph):
    return _Propagate.apply(x, graph)


class _GatAggregate(torch.autograd.Function):
    """out[i,k,:] = sum_p alpha_p h[j,k,:] with alpha the per-target softmax of leaky_relu(a_src[j] + a_dst[i]) —
    ref_cpu.gat_conv's message + aggregate (GATConv [PyG] behind models/gat.py:28,30; PARITY UNPINNED) through the C
    restatement (oracle_gat_forward_csr_f32) and its adjoint (target pass over the CSR, source pass over the transposed CSR):
    no [E', H, C] tensor, so the BASELINE-size graphs fit the host."""

    @staticmethod
    def forward(ctx, h, a_src, a_dst, graph, H, C, slope):
        lib = O._c_lib()
        h, a_src, a_dst = (t.detach().contiguous().float() for t in (h, a_src, a_dst))
        n = graph.n
        out = torch.empty((n, H * C), dtype=torch.float32)
        alpha = torch.empty((max(graph.nnz, 1), H), dtype=torch.float32)
        lib.oracle_gat_forward_csr_f32(graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), a_src.data_ptr(),
                                       a_dst.data_ptr(), float(slope), out.data_ptr(), alpha.data_ptr(), n, H, C, graph.threads)
        ctx.save_for_backward(h, a_src, a_dst, alpha)
        ctx.graph, ctx.H, ctx.C, ctx.slope = graph, H, C, slope
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = O._c_lib()
        h, a_src, a_dst, alpha = ctx.saved_tensors
        g, H, C = ctx.graph, ctx.H, ctx.C
        gout = gout.contiguous().float()
        gs = torch.empty_like(alpha)
        g_ad = torch.empty((g.n, H), dtype=torch.float32)
        lib.oracle_gat_backward_dst_csr_f32(g.rowptr.data_ptr(), g.col.data_ptr(), h.data_ptr(), a_src.data_ptr(),
                                            a_dst.data_ptr(), float(ctx.slope), alpha.data_ptr(), gout.data_ptr(),
                                            gs.data_ptr(), g_ad.data_ptr(), g.n, H, C, g.threads)
        g_h = torch.empty_like(h)
        g_as = torch.empty((g.n, H), dtype=torch.float32)
        lib.oracle_gat_backward_src_csr_f32(g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.fwd_slot.data_ptr(),
                                            alpha.data_ptr(), gs.data_ptr(), gout.data_ptr(), g_h.data_ptr(),
                                            g_as.data_ptr(), g.n, H, C, g.threads)
        return g_h, g_as, g_ad, None, None, None, None


def gat_aggregate(h, a_src, a_dst, graph, H, C, slope=0.2):
    return _GatAggregate.apply(h, a_src, a_dst, graph, H, C, slope)


def graphs_for(name, edge_index, num_nodes, threads=0):
    """The CsrGraph each model's conv layers aggregate over."""
    if name in ("gcn", "appnpstack"):
        return CsrGraph(edge_index, num_nodes, "gcn", threads=threads)
    if name == "graphsage":
        return CsrGraph(edge_index, num_nodes, "mean", loops_mode=2, threads=threads)
    if name == "graphsage2":
        return CsrGraph(edge_index, num_nodes, "mean", loops_mode=0, threads=threads)
    if name == "gat":
        return CsrGraph(edge_index, num_nodes, "gat", threads=threads)
    raise KeyError(name)


def forward(name, sd, x, graph, training, num_layers=2, K=10, alpha=0.1, heads=8):
    """{'out','emb'} of the model `name` from a product state_dict (tensors may require grad), as
    ref_cpu.{gcn,graphsage,graphsage2,appnp_stack}_forward."""
    if name == "gcn":  # ref_cpu.gcn_conv: x W^T, propagate(add), + bias
        conv = lambda i, v: propagate(v @ sd[f"convs.{i}.lin.weight"].t(), graph) + sd[f"convs.{i}.bias"]
        return O._stack(sd, x, None, num_layers, training, conv)
    if name == "graphsage":  # ref_cpu.my_sage_conv: mean(lin_l(x)) + lin_r(x)
        def conv(i, v):
            x_l = v @ sd[f"convs.{i}.lin_l.weight"].t() + sd[f"convs.{i}.lin_l.bias"]
            x_r = v @ sd[f"convs.{i}.lin_r.weight"].t() + sd[f"convs.{i}.lin_r.bias"]
            return propagate(x_l, graph) + x_r
        return O._stack(sd, x, None, num_layers, training, conv)
    if name == "graphsage2":  # ref_cpu.sage_conv: lin_l(mean_j x_j) + lin_r(x_i)
        def conv(i, v):
            return (propagate(v, graph) @ sd[f"convs.{i}.lin_l.weight"].t() + sd[f"convs.{i}.lin_l.bias"]
                    + v @ sd[f"convs.{i}.lin_r.weight"].t())
        return O._stack(sd, x, None, num_layers, training, conv)
    if name == "gat":  # ref_cpu.gat_forward / gat_conv: hidden layers `heads` heads concatenated, last layer one head
        n = x.size(0)

        def conv(i, v):
            last = i == num_layers - 1
            Hh = 1 if last else heads
            W = sd[f"convs.{i}.lin_src.weight"]
            C = W.size(0) // Hh
            h = v @ W.t()
            h3 = h.view(n, Hh, C)
            a_s = (h3 * sd[f"convs.{i}.att_src"].view(1, Hh, C)).sum(-1)
            a_d = (h3 * sd[f"convs.{i}.att_dst"].view(1, Hh, C)).sum(-1)
            return gat_aggregate(h, a_s, a_d, graph, Hh, C) + sd[f"convs.{i}.bias"]  # (one head: mean over heads = identity)
        return O._stack(sd, x, None, num_layers, training, conv)
    if name == "appnpstack":  # ref_cpu.appnp_stack_forward
        h = x @ sd["lin1.weight"].t() + sd["lin1.bias"]
        h = O.batch_norm(h, sd, "bn.", training)
        h = h @ sd["lin2.weight"].t() + sd["lin2.bias"]
        z = h
        for _ in range(K):
            z = (1 - alpha) * propagate(z, graph) + alpha * h
        return O._finish(z)
    raise KeyError(name)


def masked_nll(out, y, mask):
    """NLLLoss(log_softmax(emb)[mask], y[mask]), the loss of itexperiments.py:400,429."""
    return torch.nn.functional.nll_loss(out["out"][mask], y[mask])


def loss_and_grads(name, sd, x, y, mask, graph, **kw):
    """(loss, {parameter name: gradient}, train-mode logits) of one training forward + backward."""
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    out = forward(name, params, x, graph, True, **kw)
    loss = masked_nll(out, y, mask)
    loss.backward()
    return loss.item(), {k: v.grad for k, v in params.items() if v.requires_grad}, out["emb"].detach()


def compare_grads(named_grads, ref_grads):
    """Per parameter max |g - g_ref|, relative to max(1, |g_ref|_inf) (the north-star style bound) and relative to
    the gradient's own scale (mean-reduced losses over 10^5..10^6 rows make small gradients: the second figure is the
    one a wrong-but-small gradient cannot pass). The scale of a parameter is |g_ref|_inf, but not below 1 % of the
    largest |g_ref|_inf of the model: a conv bias in front of a BatchNorm has a TRUE gradient of exactly zero and both
    sides hold rounding noise there. Returns {'max_abs', 'max_vs_bound', 'max_rel', 'worst', 'per_param'}."""
    per, worst = {}, (0.0, None)
    assert set(named_grads) == set(ref_grads), (sorted(named_grads), sorted(ref_grads))
    top = max([rg.abs().max().item() for rg in ref_grads.values() if rg.numel()] or [0.0])
    for k, rg in ref_grads.items():
        g = named_grads[k]
        assert g is not None and g.shape == rg.shape, k
        d = (g.double() - rg.double()).abs().max().item() if rg.numel() else 0.0
        own = rg.abs().max().item() if rg.numel() else 0.0
        scale = max(own, 0.01 * top)
        per[k] = {"abs": d, "ref_inf": own, "rel": d / scale if scale > 0 else (0.0 if d == 0 else float("inf"))}
        if per[k]["rel"] > worst[0]:
            worst = (per[k]["rel"], k)
    return {"max_abs": max(v["abs"] for v in per.values()),
            "max_vs_bound": max(v["abs"] / max(1.0, v["ref_inf"]) for v in per.values()),
            "max_rel": worst[0], "worst": worst[1], "per_param": per}
