"""Train/val/test split masks, bit-exact with the reference (utils/mask.py) for a given seed —
the masks select the loss rows, so they are part of the path's inputs (golden fixture G4).
Labels equal to -1 mean "unlabelled" and never enter a split."""
import random

import torch


def _split_sizes(ratio, n):
    parts = [int(p) for p in ratio.split("-")]
    total = sum(parts)
    n_train = int(parts[0] / total * n)
    n_val = int(parts[1] / total * n)
    return n_train, n_val


def _to_mask(index, total):
    mask = torch.zeros(total, dtype=torch.bool)
    mask[index] = True
    return mask


def get_order(ratio, masked_index, total_node_num, seed=1234567):
    """Shuffle positions 0..len-1 with Python's `random` seeded by `seed`, cut at the ratio
    boundaries (floor), map back through `masked_index` (reference utils/mask.py:66-102)."""
    order = _py_shuffled_range(len(masked_index), seed)
    n_train, n_val = _split_sizes(ratio, len(order))
    pieces = (order[:n_train], order[n_train:n_train + n_val], order[n_train + n_val:])
    return tuple(_to_mask(masked_index[p], total_node_num) for p in pieces)


def _py_shuffled_range(n, seed):
    """list(range(n)) after `random.seed(seed); random.shuffle(...)` as an int64 tensor: by the library's C++ restatement of
    CPython's generator (rgbx_py_random_shuffle_i64: ~15 ms for 2 M positions where CPython takes seconds; bit-exact,
    tests/test_host_logic.py) when the library is built and the seed is an int that fits 64 bits; by `random` itself
    otherwise. The module-level `random` generator is left SEEDED with `seed` as the reference's call leaves it seeded — but on
    the C++ path not advanced by the shuffle's draws (nothing on the path draws from it afterwards: model initialisation uses
    torch's generators, RD2PD's sampling reseeds, experiment(need_to_reappear=True) reseeds)."""
    random.seed(seed)  # the reference's call has this side effect on the module-level generator
    if isinstance(seed, int) and not isinstance(seed, bool) and -2**63 <= seed < 2**63 and n < 2**32:
        try:
            from .. import _lib
            lib = _lib.load()
        except RuntimeError:
            lib = None
        if lib is not None:
            out = torch.empty(n, dtype=torch.int64)
            _lib.check(lib.rgbx_py_random_shuffle_i64(seed, n, out.data_ptr()), "rgbx_py_random_shuffle_i64")
            return out
    order = list(range(n))
    random.shuffle(order)
    return torch.tensor(order, dtype=torch.int64)


def check_train_containing(train_mask, y):
    """True iff every label other than -1 occurs under train_mask (reference utils/mask.py:24-32)."""
    present = torch.unique(y[train_mask])  # (sets of a 1.2 M-element .tolist() took a second at the benchmark's size)
    labels = torch.unique(y)
    return bool(torch.isin(labels[labels != -1], present).all())


def get_whole_mask(y, ratio, seed=1234567, max_seed_tries=20000):
    """Ratio split over all labelled nodes; the seed is bumped by one until the train part holds
    every class (reference utils/mask.py:10-22). `max_seed_tries` (an addition): seeds tried before giving up with a
    ValueError; None = the reference's unbounded loop (splits that are feasible but rare)."""
    labelled = torch.arange(len(y), dtype=torch.int64)[y != -1]
    classes, first_seed = None, seed
    while True:
        masks = get_order(ratio, labelled, len(y), seed)
        if check_train_containing(masks[0], y):
            return masks
        if max_seed_tries is not None and seed - first_seed >= max_seed_tries:
            # feasible on paper, hopeless in practice (130 classes of two nodes each and a 60 % train part: one seed in 1e10)
            raise ValueError(f"get_whole_mask: {max_seed_tries} seeds tried from {first_seed}, none puts every class into the train "
                             f"part of ratio {ratio!r} (the reference's loop over seeds would never end in practice)")
        if classes is None:
            # no seed can succeed when the train part has fewer rows than there are classes: the reference spins here for
            # ever (mask.py:16-21, `seed += 1` without an end); the same inputs get an error instead of a hang
            classes = int((torch.unique(y) != -1).sum())
            if int(masks[0].sum()) < classes:
                raise ValueError(f"get_whole_mask: the train part of ratio {ratio!r} holds {int(masks[0].sum())} of "
                                 f"{labelled.numel()} labelled nodes, fewer than the {classes} classes it must contain "
                                 "(the reference's loop over seeds would never end)")
        seed += 1


def get_classification_mask(y, ratio, seed=1234567):
    """Ratio split inside every class, unioned (reference utils/mask.py:37-62)."""
    total = len(y)
    out = [torch.zeros(total, dtype=torch.bool) for _ in range(3)]
    for lbl in y.unique().tolist():
        if lbl == -1:
            continue
        members = (y == lbl).nonzero(as_tuple=True)[0]
        for acc, part in zip(out, get_order(ratio, members, total, seed)):
            acc |= part
    return tuple(out)


def get_random_mask(y, num_train_per_class, num_val, num_test, seed):
    """Planetoid-style split (reference utils/mask.py:106-138): per class a seeded randperm picks
    the train nodes; val/test come from an UNSEEDED (global RNG) permutation of the remaining
    labelled nodes, exactly as the reference does (mask.py:133)."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    total = len(y)
    train = torch.zeros(total, dtype=torch.bool)
    for lbl in y.unique().tolist():
        if lbl == -1:
            continue
        members = (y == lbl).nonzero(as_tuple=False).view(-1)
        train[members[torch.randperm(members.size(0), generator=gen)[:num_train_per_class]]] = True
    rest = ((~train) & (y != -1)).nonzero(as_tuple=False).view(-1)
    rest = rest[torch.randperm(rest.size(0))]
    return train, _to_mask(rest[:num_val], total), _to_mask(rest[num_val:num_val + num_test], total)
