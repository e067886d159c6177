"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product.

Sampled-row check of WHOLE-MODEL logits at graph sizes where the full oracle cannot run (|V| = 2M, |E| = 60M:
the PyG dataflow of oracle.ref_cpu materialises [E', d] tensors, ~32 GB each).

The logits of a target node of an L-layer message-passing model depend only on its L-hop in-neighbourhood. So:
pick T targets, collect every edge whose target lies within L-1 hops of them (all in-edges of those "inner"
nodes, hence complete aggregations for them), relabel, and run the UNCHANGED oracle forward
(`oracle.ref_cpu.*_forward`, eval mode, running statistics) on that subgraph; its rows at the targets must equal
the rows the HIP path produced on the whole graph.

One thing the cut changes: gcn_norm (reference models/dagnn.py:12-31) weighs an edge j -> i by deg(j)^-1/2 deg(i)^-1/2
with deg = in-degree, and the outermost sources j lose their in-edges in the cut. `complete_in_degree=True` restores
their in-degree exactly by attaching the missing number of in-edges from one extra all-zero dummy node (whose own
output is never read): same arithmetic as on the whole graph, no algebra in the checker. Mean aggregation
(my_SAGEConv / SAGEConv) and GAT attention only involve per-target quantities and need no completion.
"""
import torch

from . import ref_cpu as O


def khop_in_subgraph(edge_index, num_nodes, targets, hops, complete_in_degree=False):
    """(sub_edge_index int64 [2, E_sub], node_ids int64 [n_sub] = original ids (the dummy node, if any, is the extra
    last index n_sub), target_pos int64 [T], n_total = n_sub (+ 1 with a dummy node))."""
    src, dst = edge_index[0], edge_index[1]
    flag = torch.zeros(num_nodes, dtype=torch.bool)
    flag[targets] = True
    for _ in range(hops - 1):
        flag[src[flag[dst]]] = True  # inner nodes: every in-edge of theirs is kept
    keep = flag[dst]
    s, t = src[keep], dst[keep]
    flag_nodes = flag.clone()
    flag_nodes[s] = True
    node_ids = flag_nodes.nonzero(as_tuple=True)[0]
    pos = torch.full((num_nodes,), -1, dtype=torch.int64)
    pos[node_ids] = torch.arange(node_ids.numel())
    sub = torch.stack([pos[s], pos[t]])
    n_total = int(node_ids.numel())
    if complete_in_degree:
        indeg = torch.bincount(dst[src != dst], minlength=num_nodes)  # self-loops are rewritten by the conv layers
        outer = node_ids[~flag[node_ids]]
        missing = indeg[outer]
        if int(missing.sum()) > 0:
            dummy = n_total
            n_total += 1
            d_dst = torch.repeat_interleave(pos[outer], missing)
            sub = torch.cat([sub, torch.stack([torch.full_like(d_dst, dummy), d_dst])], dim=1)
    return sub, node_ids, pos[targets], n_total


def _sub_features(x, node_ids, n_total):
    xs = x[node_ids]
    if n_total > node_ids.numel():
        xs = torch.cat([xs, xs.new_zeros((n_total - node_ids.numel(), x.size(1)))])
    return xs


# model name -> (hops the logits depend on, gcn_norm in the path, oracle forward)
def _forward(name, sd, xs, sub, kw):
    if name == "gcn":
        return O.gcn_forward(sd, xs, sub, kw["num_layers"], training=False)
    if name == "graphsage":
        return O.graphsage_forward(sd, xs, sub, kw["num_layers"], training=False)
    if name == "graphsage2":
        return O.graphsage2_forward(sd, xs, sub, kw["num_layers"], training=False)
    if name == "gat":
        return O.gat_forward(sd, xs, sub, kw["num_layers"], kw["heads"], training=False)
    if name == "appnpstack":
        return O.appnp_stack_forward(sd, xs, sub, kw["K"], kw["alpha"], training=False)
    if name == "sgc":
        return O.sgc_forward(sd, xs, sub, kw["K"])
    if name == "gin":
        return O.gin_forward(sd, xs, sub, kw["num_layers"], training=False)
    if name == "dagnn":
        return O.dagnn_forward(sd, xs, sub, kw["K"])
    raise KeyError(name)


def sampled_logits(name, state_dict, x, edge_index, targets, **kw):
    """Oracle logits [T, C] of `targets` for the model `name` with the product's `state_dict` (CPU tensors),
    eval mode. kw: num_layers (conv stacks), heads (gat), K / alpha (appnpstack: keep K small, the
    neighbourhood grows by the mean degree per hop)."""
    hops = kw["K"] if name in ("appnpstack", "sgc", "dagnn") else kw["num_layers"]
    needs_degree = name in ("gcn", "appnpstack", "sgc", "dagnn")
    sub, node_ids, tpos, n_total = khop_in_subgraph(edge_index, x.size(0), targets, hops, needs_degree)
    xs = _sub_features(x, node_ids, n_total)
    sd = {k: v for k, v in state_dict.items()}
    with torch.no_grad():
        emb = _forward(name, sd, xs, sub, kw)["emb"]
    return emb[tpos], {"sub_nodes": n_total, "sub_edges": int(sub.size(1)), "targets": int(targets.numel())}


def pick_targets(num_nodes, count, seed=20261004):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(num_nodes, generator=g)[:count].sort()[0]
