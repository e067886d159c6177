"""experiment() as one of several ranks (new capability: the reference's experiment() is single-device,
itexperiments.py:246 `cuda_index`). The loop body, early stopping and the returned metrics stay those of
rgb_experiment_amd.itexperiments.experiment (reference :417-504, :603-605); what changes is where an epoch runs:
DistRunner.epoch() on the 1-D node partition, one process per GPU, RCCL collectives.

Launch: `python -m torch.distributed.run --nproc-per-node N your_script.py`, the script calling experiment() exactly as
on one GPU — every rank loads the same data and passes the same arguments; `distributed=None` (default) sees
WORLD_SIZE > 1 and takes this route. All ranks return the same result dict."""
import os

import torch
import torch.distributed as dist

SUPPORTED = ("gcn", "graphsage", "graphsage2", "gat", "appnpstack")


class DistContext:
    def __init__(self, rank, world, device, test_backend):
        self.rank, self.world, self.device, self.test_backend = rank, world, device, test_backend
        self.cuda_index = device.index if device.type == "cuda" else 0

    @classmethod
    def open(cls, model_name, post_cs, use_cpu):
        """Join (or create from the launcher's environment) the process group and pick this rank's device."""
        if model_name not in SUPPORTED:
            raise NotImplementedError(f"distributed experiment(): model_name={model_name!r} has no node-partitioned form "
                                      f"(supported: {', '.join(SUPPORTED)}); run it on one GPU (distributed=False)")
        if post_cs:
            raise NotImplementedError("distributed experiment(): post_cs (Correct & Smooth) runs on one GPU only")
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world < 2 and not dist.is_initialized():
            raise RuntimeError("distributed=True needs several ranks: start the script with "
                               "`python -m torch.distributed.run --nproc-per-node N ...` (RANK / WORLD_SIZE in the environment)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks of one node
        backend = os.environ.get("RGBX_DIST_BACKEND", "nccl")
        test_backend = None
        spec = os.environ.get("RGBX_TEST_AGGREGATOR")
        if torch.cuda.is_available() and not use_cpu:
            local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
        elif backend == "gloo" and spec:
            # CPU rehearsal of the host logic (tests): gloo collectives and an aggregator the TEST injects; the product
            # has no CPU aggregation path of its own
            mod, cls_name = spec.split(":")
            test_backend = getattr(__import__(mod, fromlist=[cls_name]), cls_name)()
            device = torch.device("cpu")
        else:
            raise RuntimeError("rgb_experiment_amd runs message passing in HIP kernels on an MI355X device; "
                               "no visible GPU / use_cpu=True is not supported (no CPU fallback)")
        if not dist.is_initialized():
            if backend == "nccl":
                from . import sharing
                if sharing.init_rccl(device) and int(os.environ.get("RANK", "0")) == 0:
                    print(f"rgb_experiment_amd: some of the {world} ranks share a GPU - RCCL over its socket transport "
                          "(dist/sharing.py): a rehearsal of the distributed path, not a faster run", flush=True)
            else:
                dist.init_process_group(backend)
        return cls(dist.get_rank(), dist.get_world_size(), device, test_backend)

    def _from_rank0(self, t):
        """Rank 0's values of a tensor every rank holds (gloo has no device broadcast: staged through the host)."""
        if dist.get_backend() == "nccl" or not t.is_cuda:
            dist.broadcast(t, 0)
            return t
        h = t.cpu()
        dist.broadcast(h, 0)
        return t.copy_(h)

    def runner(self, net, edge_index, features, y, masks, lr, weight_decay, cache_input_aggregate, task_split="auto",
               share_eval_forward=False):
        """DistRunner over this rank's node range. Parameters, buffers and masks are taken from rank 0, so that the
        ranks agree even when the caller seeded nothing (need_to_reappear=False, or a splitter that draws from the
        global generator: utils/mask.py get_random_mask's val / test shuffle, reference mask.py:133)."""
        from . import tasksplit
        from .comm import Comm
        from .runner import DistRunner
        with torch.no_grad():
            for t in list(net.parameters()) + list(net.buffers()):
                self._from_rank0(t.data)
        masks = tuple(self._from_rank0(m.to(torch.uint8)).bool() for m in masks)
        widths = sorted({features.size(1)} | {p.size(0) for p in net.parameters() if p.dim() == 2})
        if tasksplit.resolve(task_split, net, self.world, features.size(0), edge_index.size(1), widths, self.device):
            # training steps on one half of the ranks, eval forwards on the other (APPNP stacks whose column slices
            # would fall below the 128-byte line; two ranks when one GPU can hold the whole graph: dist/tasksplit.py);
            # same loop, same numbers. task_split="off" (or a graph one GPU cannot hold) = the partitioned DistRunner
            return tasksplit.TaskSplitRunner(net, edge_index, features, y, masks, self.rank, self.world, self.device,
                                             lr=lr, weight_decay=weight_decay, comm=Comm(), backend=self.test_backend,
                                             share_eval_forward=share_eval_forward)
        return DistRunner(net, edge_index, features, y, masks, self.rank, self.world, self.device, lr=lr,
                          weight_decay=weight_decay, comm=Comm(), backend=self.test_backend,
                          cache_input_aggregate=cache_input_aggregate, share_eval_forward=share_eval_forward)

    def final_test(self, runner, y, test_mask, need_all_metrics, compare_pred_label):
        """Eval-mode forward of the (best) model on every rank's rows; predictions and labels of the test rows are
        gathered so that every rank computes the same metrics dict (reference test(), :611-634, on the whole graph)."""
        emb = runner.logits(training=False)
        out = torch.log_softmax(emb, dim=1)
        m = runner.masks[2]  # this rank's rows of the test mask every rank agreed on (runner())
        if getattr(runner, "role", "train") != "train":  # task split: both groups hold every row; one of them reports
            m = torch.zeros_like(m)
        pred = out.max(dim=1)[1][m].cpu()
        label = runner.y[m].cpu()
        parts = [None] * self.world
        dist.all_gather_object(parts, (pred, label))
        pred = torch.cat([p for p, _ in parts])
        label = torch.cat([lab for _, lab in parts])
        res = compare_pred_label(pred, label, need_all_metrics)
        res.update({"pred": pred, "label": label, "emb": emb, "test_op": out})  # emb / test_op: this rank's rows
        return res
