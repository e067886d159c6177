"""Edge-list utilities with the semantics of the PyG helpers the reference calls once per run, in front of the hot
loop: to_undirected (itexperiments.py:238), coalesce / remove_self_loops / add_remaining_self_loops (rd2pd.py:92-101)
[PyG].

Device tensors take the HIP library: coalesce and to_undirected are rgbx_coalesce_keys_i64 + rgbx_split_edge_keys_i64
(64-bit keys row * N + col, rocPRIM radix sort over 2 * bits(N) bits, unique, split; csrc/ingest.hip) and raise
RuntimeError when the library is missing. CPU tensors (the reference does these edits on the CPU, rd2pd.py:3) take the
same arithmetic in plain torch — which is also what the `-m gpu` tests compare the device result against, bit for bit."""
import ctypes

import torch


def _coalesce_device(edge_index, num_nodes, mirror):
    from .. import _lib
    lib = _lib.load()
    _lib.require_device(edge_index)
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise RuntimeError(f"edge_index must be int64 [2, E], got {edge_index.dtype} {tuple(edge_index.shape)}")
    dev = edge_index.device
    E, N = int(edge_index.size(1)), int(num_nodes)
    M = 2 * E if mirror else E
    if M == 0:
        return edge_index.new_empty((2, 0))
    row, col = edge_index[0].contiguous(), edge_index[1].contiguous()
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.rgbx_coalesce_workspace_bytes(E, N, int(mirror), ctypes.byref(nbytes)), "rgbx_coalesce_workspace_bytes")
    keys = torch.empty(M, dtype=torch.int64, device=dev)  # uint64 keys below 2^62
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=dev)
    stream = _lib.stream_ptr()
    _lib.check(lib.rgbx_coalesce_keys_i64(_lib.ptr(row), _lib.ptr(col), E, N, int(mirror), _lib.ptr(keys),
                                          _lib.ptr(counts), _lib.ptr(ws), ws.numel(), stream), "rgbx_coalesce_keys_i64")
    n_out, n_bad = counts.tolist()  # the one read-back: the output's size is a result
    if n_bad:
        raise RuntimeError(f"edge_index has {n_bad} endpoints outside [0, {N})")
    out = torch.empty((2, n_out), dtype=torch.int64, device=dev)
    _lib.check(lib.rgbx_split_edge_keys_i64(_lib.ptr(keys), _lib.ptr(counts), n_out, N, _lib.ptr(out[0]),
                                            _lib.ptr(out[1]), stream), "rgbx_split_edge_keys_i64")
    return out


def _check_range(edge_index, num_nodes):
    """The CPU path refuses what the device path refuses (rgbx_coalesce_keys_i64 counts endpoints outside [0, N) and the
    result is discarded): the keys row * N + col of such an edge would alias another pair's. The reference's helpers
    (torch_sparse.coalesce, to_undirected [PyG]) produce a silent wrong result there; one behaviour on both devices."""
    if edge_index.numel():
        lo, hi = int(edge_index.min()), int(edge_index.max())
        if lo < 0 or hi >= num_nodes:
            bad = int(((edge_index < 0) | (edge_index >= num_nodes)).sum())
            raise RuntimeError(f"edge_index has {bad} endpoints outside [0, {int(num_nodes)})")


def coalesce(edge_index, num_nodes):
    """Sort edges by (row, col) and drop duplicates."""
    if edge_index.numel() == 0:
        return edge_index
    if edge_index.is_cuda:
        return _coalesce_device(edge_index, num_nodes, False)
    _check_range(edge_index, num_nodes)
    key = edge_index[0] * num_nodes + edge_index[1]
    key = torch.unique(key, sorted=True)
    return torch.stack([torch.div(key, num_nodes, rounding_mode="floor"), key % num_nodes])


def to_undirected(edge_index, num_nodes=None):
    """Add every reverse edge, then coalesce."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    if edge_index.is_cuda and edge_index.numel():
        return _coalesce_device(edge_index, num_nodes, True)
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    return coalesce(both, num_nodes)


def remove_self_loops(edge_index):
    return edge_index[:, edge_index[0] != edge_index[1]]


def add_remaining_self_loops(edge_index, num_nodes):
    """Non-loop edges in order, followed by one self-loop per node."""
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([remove_self_loops(edge_index), loops.unsqueeze(0).repeat(2, 1)], dim=1)
