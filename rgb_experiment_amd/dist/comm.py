"""Thin collective layer over torch.distributed. Product: backend 'nccl' (= RCCL over xGMI), device
tensors, the all-to-all runs asynchronously on RCCL's stream so the local-edge SpMM overlaps it.
Rehearsal/tests: backend 'gloo'; device tensors are staged through host memory (gloo has no device
all-to-all), CPU tensors go straight through."""
import threading

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


class TakeTurns:
    """Two (or more) host threads issue work for ONE communicator in a fixed, rank-independent order: a thread
    runs until it is about to WAIT for an exchange it has issued, then hands the turn to the next thread that is
    still running. With one eval forward per thread (DistRunner.epoch) the second forward's kernels and exchanges
    are enqueued while the first one's exchange is in flight, so the collectives of both forwards interleave on
    RCCL's one stream in the same order on every rank (the hand-over points depend only on the model and the
    scheme, which all ranks share) and each forward's exchange overlaps the other's aggregation."""

    def __init__(self, n):
        self.cv = threading.Condition()
        self.turn = 0
        self.alive = [True] * n
        self.local = threading.local()

    def enter(self, me):
        self.local.me = me
        with self.cv:
            self.cv.wait_for(lambda: self.turn == me)

    def _pass_on(self, me):
        n = len(self.alive)
        for step in range(1, n + 1):
            cand = (me + step) % n
            if self.alive[cand]:
                self.turn = cand
                break
        self.cv.notify_all()

    def hand_over(self):
        me = getattr(self.local, "me", None)
        if me is None:
            return
        with self.cv:
            self._pass_on(me)
            self.cv.wait_for(lambda: self.turn == me)

    def leave(self):
        me = self.local.me
        with self.cv:
            self.alive[me] = False
            self._pass_on(me)
        self.local.me = None


class _TimedWait:
    """`work.wait()` bracketed by HIP events on the waiting stream when bench.py listens (ops.set_event_sink): the
    elapsed time of an "exchange_wait" record is how long that stream stood still for the exchange — the EXPOSED part
    of it, measured rather than modelled."""

    def __init__(self, work):
        self.work = work

    def wait(self):
        from .. import ops
        with ops._Timed("exchange_wait"):
            return self.work.wait()


class _TurnWork:
    """`work.wait()` that first lets the other forward issue its share (TakeTurns)."""

    def __init__(self, work, turns):
        self.work, self.turns = work, turns

    def wait(self):
        self.turns.hand_over()
        return self.work.wait()


class Comm:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.turns = None  # a TakeTurns while two forwards are being interleaved
        self.bytes_sent = 0  # payload this rank handed to all-to-alls (rows * width * 4), for the bench line
        self.exchanges = 0
        self.link_gbs = None  # measured by measure_link_gbs(); the cost model's link rate when set
        self.link_latency_us = None

    def _account(self, send, send_counts):
        own = send_counts[self.rank] if self.rank < len(send_counts) else 0
        self.bytes_sent += (int(sum(send_counts)) - int(own)) * send.size(1) * send.element_size()
        self.exchanges += 1

    def all_to_all_rows(self, send, send_counts, recv_counts, tag=None):
        """Row-wise all-to-all: rank q receives send[offs[q]:offs[q+1]] of every peer, concatenated in
        rank order. Returns (recv, work); call work.wait() before reading recv. `tag` names the exchange's role
        ("in", "out k/n", "halo", "resident", ...) for the emulated run's link table."""
        n_recv = int(sum(recv_counts))
        recv = torch.empty((n_recv, send.size(1)), dtype=send.dtype, device=send.device)
        if self.world == 1:
            return recv, _Done()
        self._account(send, send_counts)
        if self.backend == "nccl" or not send.is_cuda:
            work = dist.all_to_all_single(recv, send.contiguous(), list(recv_counts), list(send_counts),
                                          group=self.group, async_op=True)
            if send.is_cuda:
                work = _TimedWait(work)
            return recv, (work if self.turns is None else _TurnWork(work, self.turns))
        # three statements, so that a stack dump of a stalled rehearsal says WHICH of them it sat in: the device-to-host
        # copy with its stream wait, the host collective, or the copy back (round 4's dumps could not tell the first two
        # apart: DESIGN.md 4.6)
        host_send = self._to_host(send)
        host_recv = self._host_buffer(recv.shape, recv.dtype, recv.is_cuda)
        dist.all_to_all_single(host_recv, host_send, list(recv_counts), list(send_counts), group=self.group)
        self._from_host(recv, host_recv)
        done = _TimedWait(_Done()) if send.is_cuda else _Done()  # same wrappers as the RCCL path (rehearsals run them)
        return recv, (done if self.turns is None else _TurnWork(done, self.turns))

    def _to_host(self, t):
        """gloo rehearsals with device tensors: the rows go through a PINNED staging buffer, copied on the caller's
        current stream and waited for on that stream alone. (A pageable `.cpu()` takes the runtime's own staging path;
        with several ranks sharing one GPU and two host threads per rank — the interleaved eval forwards on their side
        streams — that copy was seen to stall for minutes on some ranks of a 4-rank S-size run while the peers sat in the
        collective: tests/test_gpu_dist.py's APPNP reshard(4) case at workload S, twice in a row inside the suite.)"""
        if not t.is_cuda:
            return t.contiguous()
        t = t.contiguous()
        pool = self.__dict__.setdefault("_pinned", {})
        key = (self._staging_slot(), t.dtype)
        buf = pool.get(key)
        if buf is None or buf.numel() < t.numel():
            buf = torch.empty(max(t.numel(), 1 << 20), dtype=t.dtype, pin_memory=True)
            pool[key] = buf
        host = buf[:t.numel()].view(t.shape)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host

    def _staging_slot(self):
        """Which pinned staging buffer the calling thread uses: the forward's index while two eval forwards take turns
        (TakeTurns hands every thread its `me`), "main" otherwise — a bounded set of keys. (Keyed by thread ident, as in
        round 4, every epoch's fresh pair of eval threads could leave four more page-locked buffers behind.)"""
        turns = self.turns
        me = getattr(turns.local, "me", None) if turns is not None else None
        return "main" if me is None else int(me)

    def _host_buffer(self, shape, dtype, pinned):
        """Receive side of the same staging: one pinned buffer per host thread and dtype, grown on demand."""
        n = 1
        for d in shape:
            n *= int(d)
        if not pinned:
            return torch.empty(shape, dtype=dtype)
        pool = self.__dict__.setdefault("_pinned", {})
        key = (self._staging_slot(), dtype, "recv")
        buf = pool.get(key)
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1 << 20), dtype=dtype, pin_memory=True)
            pool[key] = buf
        return buf[:n].view(shape)

    def _from_host(self, dst, host):
        dst.copy_(host, non_blocking=True)
        if dst.is_cuda:
            torch.cuda.current_stream(dst.device).synchronize()

    def all_to_all_views(self, send, recv, tag=None):
        """All-to-all of tensors where they lie: `send[q]` (contiguous, possibly empty) goes to rank q, `recv[q]`
        (contiguous view, possibly empty) is filled by rank q's `send[self.rank]`. The same tensor may stand in several
        slots of `send` (a column slice that several ranks need is not duplicated) and the `recv` views may be row
        ranges of one blocked buffer (received slices land where their consumer reads them: no unpack pass).
        RCCL: one grouped send / recv (torch.distributed.all_to_all). gloo rehearsals (no list form there): the pieces
        are staged through one buffer each way. Returns a work object; call .wait() before reading `recv`."""
        if self.world == 1:
            if recv[0].numel():
                recv[0].copy_(send[0])
            return _Done()
        width = max((t.size(-1) for t in send if t.numel()), default=0)
        self.bytes_sent += sum(t.numel() for q, t in enumerate(send) if q != self.rank) * 4
        self.exchanges += 1
        cuda = any(t.is_cuda for t in send) or any(t.is_cuda for t in recv)
        if self.backend == "nccl":
            work = dist.all_to_all(list(recv), list(send), group=self.group, async_op=True)
            work = _TimedWait(work)
            return work if self.turns is None else _TurnWork(work, self.turns)
        # gloo: flatten, exchange, scatter back (host-staged for device tensors)
        sc = [t.numel() for t in send]
        rc = [t.numel() for t in recv]
        flat = torch.cat([t.reshape(-1) for t in send]) if sum(sc) else send[0].new_empty(0)
        host_send = self._to_host(flat)  # (separate statements: see all_to_all_rows)
        host_recv = self._host_buffer((sum(rc),), flat.dtype, cuda)
        dist.all_to_all_single(host_recv, host_send, rc, sc, group=self.group)
        off = 0
        for t, n in zip(recv, rc):
            if n:
                t.copy_(host_recv[off:off + n].view(t.shape), non_blocking=True)
            off += n
        if cuda:
            torch.cuda.current_stream().synchronize()  # the pinned buffer is handed out again by the next exchange
        del width
        done = _TimedWait(_Done()) if cuda else _Done()
        return done if self.turns is None else _TurnWork(done, self.turns)

    def self_test_views(self, device):
        """One small view exchange with a known answer, before the fused schedule relies on `all_to_all_views`: blocks
        of one buffer handed to several peers (the same view in more than one slot), row ranges of one buffer as
        receive targets, and EMPTY entries for some pairs — what the schedule does, at 8 rows; then the same with several
        works IN FLIGHT at once and waited for out of order (_self_test_in_flight). Every rank checks what it
        received, the verdict is all-reduced (MIN) so that all ranks agree; False means this backend / build does not
        deliver the views as assumed and the caller must stay on the single-buffer exchanges."""
        P, r = self.world, self.rank
        if P == 1:
            return True
        C, rows, w = (2 if P % 2 == 0 else 1), 8, 4
        base = torch.arange(C * rows * w, dtype=torch.float32, device=device).view(C, rows, w)
        src = base + 1000.0 * r
        dst = torch.full((P, rows, w), -1.0, dtype=torch.float32, device=device)
        silent = lambda a, b: a != b and (a + b) % 3 == 0  # these pairs exchange nothing (both ways)
        send = [src[0, 0:0] if silent(r, q) else src[q % C] for q in range(P)]
        recv = [dst[0, 0:0] if silent(r, q) else dst[q] for q in range(P)]
        self.all_to_all_views(send, recv, tag="self-test").wait()
        ok = True
        for q in range(P):
            want = torch.full((rows, w), -1.0, device=device) if silent(r, q) else base[r % C] + 1000.0 * q
            ok = ok and bool(torch.equal(dst[q], want))
        ok = ok and self._self_test_in_flight(device)
        verdict = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
        if self.backend == "nccl" or not verdict.is_cuda:
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
        else:
            h = verdict.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=self.group)
            verdict.copy_(h)
        self.bytes_sent, self.exchanges = 0, 0
        return bool(verdict.item() == 1.0)

    def _self_test_in_flight(self, device):
        """What the fused schedule actually does with the communicator (dist/stack.py GridStack._interleave), at 8 rows,
        with a known answer: TWO piece-wise view exchanges (row pieces of one blocked send buffer into row ranges of one
        receive buffer) and an async all-reduce outstanding TOGETHER on the same communicator, waited for OUT of issue
        order (second piece, the all-reduce, first piece). A backend that serialises, reorders or mis-pairs grouped
        send / recv lists under several works in flight shows here, before the first epoch, not in trained weights."""
        P, r = self.world, self.rank
        rows, w, pieces = 8, 4, 2
        half = rows // pieces
        src = (torch.arange(P * rows * w, dtype=torch.float32, device=device).view(P, rows, w) + 1000.0 * r)
        dst = torch.full((P, rows, w), -1.0, dtype=torch.float32, device=device)
        red = torch.full((16,), float(r + 1), dtype=torch.float32, device=device)
        works = []
        for k in range(pieces):  # piece k: rows [k * half, (k + 1) * half) of every peer's block
            lo, hi = k * half, (k + 1) * half
            works.append(self.all_to_all_views([src[q, lo:hi] for q in range(P)], [dst[q, lo:hi] for q in range(P)],
                                               tag=f"self-test piece {k}"))
        red_work = self.all_reduce_sum_async(red)
        works[1].wait()
        red_work.wait()
        works[0].wait()
        base = torch.arange(P * rows * w, dtype=torch.float32, device=device).view(P, rows, w)
        ok = bool(torch.equal(red, torch.full_like(red, P * (P + 1) / 2.0)))
        for q in range(P):  # rank q sent its block number r
            ok = ok and bool(torch.equal(dst[q], base[r] + 1000.0 * q))
        return ok

    def measure_link_gbs(self, device, mb_per_peer=16, reps=3):
        """GB/s one xGMI link carries per direction under an all-to-all (every pair busy at once), measured: `reps`
        timed all-to-alls of `mb_per_peer` MB per peer after two warm-ups; the slowest rank's time counts and all ranks
        get the same figure (it feeds the exchange cost model, which must agree everywhere). RCCL only: None
        otherwise (gloo rehearsals, one rank)."""
        if self.world == 1 or self.backend != "nccl":
            return None
        import time
        rows = mb_per_peer * 1024 * 1024 // (64 * 4)
        send = torch.ones((rows * self.world, 64), dtype=torch.float32, device=device)
        counts = [rows] * self.world
        for _ in range(2):
            self.all_to_all_rows(send, counts, counts)[1].wait()
        torch.cuda.synchronize(device)
        dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(reps):
            self.all_to_all_rows(send, counts, counts)[1].wait()
        torch.cuda.synchronize(device)
        dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX, group=self.group)
        self.link_gbs = rows * 64 * 4 / dt.item() / 1e9
        # and what one SMALL all-to-all costs end to end (one row per peer, issued and waited for back to back): the
        # per-exchange latency a scheme with many pieces pays; on the bench line and in DistGraph.costs (piece count)
        tiny = torch.ones((self.world, 64), dtype=torch.float32, device=device)
        ones = [1] * self.world
        for _ in range(3):
            self.all_to_all_rows(tiny, ones, ones)[1].wait()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(20):
            self.all_to_all_rows(tiny, ones, ones)[1].wait()
        torch.cuda.synchronize(device)
        lat = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64, device=device)
        dist.all_reduce(lat, op=dist.ReduceOp.MAX, group=self.group)
        self.link_latency_us = lat.item() * 1e6
        self.bytes_sent, self.exchanges = 0, 0
        return self.link_gbs

    def all_reduce_sum_(self, t):
        if self.world == 1:
            return t
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(h)
        return t

    def all_reduce_sum_async(self, t):
        """all_reduce_sum_ in place, returned as a work object: RCCL enqueues it on its own stream and .wait() makes
        the caller's stream depend on it — a schedule that has other kernels to enqueue does that first (dist/stack.py)."""
        if self.world == 1:
            return _Done()
        if self.backend == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.all_reduce_sum_(t)
        return _Done()

    def all_reduce_max_(self, t):
        if self.world == 1:
            return t
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
        t.copy_(h)
        return t

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

    def abort(self, exc, where=""):
        """An error on this rank that its peers cannot see (raised on a helper thread, or between two collectives
        the peers are already waiting in): end the process NOW with the traceback on stderr, so that the launcher
        (torch.distributed.run / bench.py's supervisors) ends the peers, instead of a job-wide hang. No-op for a
        one-process run, where the caller re-raises."""
        if self.world == 1 or not dist.is_initialized():
            return
        import os
        import sys
        import traceback
        sys.stderr.write(f"[rgb_experiment_amd.dist rank {self.rank}] fatal error in {where}: ending the process so that "
                         "the peers do not wait in a collective\n")
        traceback.print_exception(type(exc), exc, exc.__traceback__)
        sys.stderr.flush()
        os._exit(70)


class _TracedDone:
    """Work of an emulated exchange: .wait() leaves a marker in the event trace bench.py listens to, so that the
    schedule (which kernels were enqueued between an exchange's issue and the wait for it) can be replayed against a
    link model."""

    def __init__(self, xid, done=None):
        self.xid = xid
        self.done = done  # HIP event on the emulated link stream (contended emulation), else None

    def wait(self):
        from .. import ops
        if ops._EVENT_SINK is not None:
            ops._EVENT_SINK.append(("@wait", self.xid))
        if self.done is not None:  # contended emulation: the waiting stream stands still until "the links" have delivered
            with ops._Timed("exchange_wait"):
                torch.cuda.current_stream().wait_event(self.done)
        return None


class EmulatedComm(Comm):
    """ONE process standing in for rank `rank` of a `world`-rank job (bench.py --emulate-rank): every structure
    and every kernel launch is exactly that rank's; an exchange delivers stand-in rows (copies of the rows being
    sent — the values are meaningless, the shapes and the memory traffic of the kernels that consume them are not)
    and records the bytes every link would carry. Timing such a run gives the rank's COMPUTE per epoch; the
    exchange time is then link arithmetic on `link_bytes`."""

    def __init__(self, world, rank=0):
        self.group = None
        self.world, self.rank, self.backend = int(world), int(rank), "emulated"
        self.turns = None
        self.bytes_sent = 0
        self.exchanges = 0
        self.log = []  # (tag, max bytes sent to one peer, max bytes received from one peer) per exchange
        self._pool = {}
        self.link_gbs = None
        self.link_latency_us = None

    def all_to_all_rows(self, send, send_counts, recv_counts, tag=None):
        n_recv = int(sum(recv_counts))
        width = send.size(1)
        self._account(send, send_counts)
        peers_out = [c for q, c in enumerate(send_counts) if q != self.rank]
        peers_in = [c for q, c in enumerate(recv_counts) if q != self.rank]
        self.log.append((tag, max(peers_out, default=0) * width * 4, max(peers_in, default=0) * width * 4))
        work = self._traced(tag, max(max(peers_out, default=0), max(peers_in, default=0)) * width * 4)
        if send.size(0) == 0 or n_recv == 0:
            return send.new_zeros((n_recv, width)), work
        # a real exchange lands the rows by DMA, at no cost in kernel time: the stand-in buffer of every exchange
        # shape is filled once (with rows being sent, so the values are ordinary activations) and handed out again
        key = (tag, n_recv, width, getattr(self.turns, "local", None) and getattr(self.turns.local, "me", None))
        recv = self._pool.get(key)
        if recv is None:
            reps = -(-n_recv // send.size(0))
            recv = (send if reps == 1 else send.repeat(reps, 1))[:n_recv].clone()
            self._pool[key] = recv
        return recv, (work if self.turns is None else _TurnWork(work, self.turns))

    def all_to_all_views(self, send, recv, tag=None):
        """Stand-in for the view exchange: the receive views are filled ONCE per (tag, shapes) with rows being sent
        (ordinary activations, so the kernels that consume them see ordinary values) and the bytes the busiest link
        would carry are logged; a real exchange lands the rows by DMA at no cost in kernel time."""
        out_b = [t.numel() * 4 for q, t in enumerate(send) if q != self.rank]
        in_b = [t.numel() * 4 for q, t in enumerate(recv) if q != self.rank]
        self.bytes_sent += sum(out_b)
        self.exchanges += 1
        self.log.append((tag, max(out_b, default=0), max(in_b, default=0)))
        work = self._traced(tag, max(max(out_b, default=0), max(in_b, default=0)))
        key = ("views", tag, tuple(t.data_ptr() for t in recv if t.numel()),
               getattr(self.turns, "local", None) and getattr(self.turns.local, "me", None))
        if key not in self._pool:
            src = next((t for t in send if t.numel()), None)
            for t in recv:
                if t.numel() and src is not None:
                    flat = src.reshape(-1)
                    reps = -(-t.numel() // flat.numel())
                    t.copy_((flat if reps == 1 else flat.repeat(reps))[:t.numel()].view(t.shape))
            self._pool[key] = True
            if len(self._pool) > 4096:
                self._pool.clear()
        return work if self.turns is None else _TurnWork(work, self.turns)

    def _traced(self, tag, link_bytes):
        """Marker of an exchange's ISSUE in bench.py's event trace (kernels recorded before it are its producers) and a
        work object whose wait() marks where the schedule starts to depend on it."""
        from .. import ops
        self._xid = getattr(self, "_xid", 0) + 1
        if ops._EVENT_SINK is not None:
            ops._EVENT_SINK.append(("@issue", self._xid, tag, link_bytes))
        return _TracedDone(self._xid, self._contend_issue(link_bytes))

    # ---- contended emulation (bench.py --emulate-contend GBS) ---------------------------------------------------------
    def enable_contention(self, device, link_gbs, max_bytes=1 << 30, nontemporal=False, sync_only=False):
        """From now on every exchange MOVES its bytes on this GPU while the rank computes: a paced device-to-device copy
        of (world - 1) x [bytes on the busiest link] on a separate "link" stream, at (world - 1) x link_gbs GB/s — the
        inbound rows written into this rank's HBM plus as many outbound bytes read from it, through `wgs` workgroups
        (RCCL's copy kernels run on CUs too) — started when the launches enqueued before the exchange's issue have
        finished (FIFO over one stream, like the link model of bench.replay_schedule), and the consumer WAITS for it.
        The measured step time then contains the exposed exchange time AND what the traffic costs the kernels it runs
        beside (HBM bandwidth, L2 / Infinity Cache space, CU slots): the pessimistic one-GPU figure."""
        from .. import _lib
        import time
        n = max_bytes // 4
        if getattr(self, "_link", None) is None:
            self._link = torch.cuda.Stream(device)
            self._paced_src = torch.empty(n, dtype=torch.float32, device=device).normal_()
            self._paced_dst = torch.empty(n, dtype=torch.float32, device=device)
        target = (self.world - 1) * float(link_gbs)
        lib = _lib.load()

        def rate(wgs, nbytes=256 << 20):
            k = min(nbytes // 4, n) // 4 * 4
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            with torch.cuda.stream(self._link):
                for _ in range(3):
                    _lib.check(lib.rgbx_paced_copy_f32(self._paced_src.data_ptr(), self._paced_dst.data_ptr(), k, wgs,
                                                       int(nontemporal), self._link.cuda_stream), "rgbx_paced_copy_f32")
            self._link.synchronize()
            return 3 * k * 4 / (time.perf_counter() - t0) / 1e9

        rate(64)  # warm-up
        table = {w: rate(w) for w in (4, 8, 16, 32, 64, 128, 256)}
        # the smallest workgroup count that reaches the target rate (copy kernels cannot be slowed below one workgroup's
        # rate; a rate above the target ends an exchange early, i.e. errs towards LESS exposure but MORE contention)
        wgs = next((w for w in sorted(table) if table[w] >= target), max(table))
        self._contend = {"link_gbs": float(link_gbs), "target_copy_gbs": target, "workgroups": wgs,
                         "nontemporal": bool(nontemporal), "sync_only": bool(sync_only),
                         "copy_gbs_measured": table[wgs], "calibration_gbs_by_workgroups": table}
        return self._contend

    def _contend_issue(self, link_bytes):
        c = getattr(self, "_contend", None)
        if c is None or link_bytes <= 0:
            return None
        from .. import _lib
        n = min(int(link_bytes) * (self.world - 1) // 4, self._paced_src.numel()) // 4 * 4
        if n == 0:
            return None
        if c["sync_only"]:  # control: the same stream hand-overs and launches, 4 KB instead of the exchange's bytes
            n = min(n, 1024)
        cur = torch.cuda.current_stream()
        self._link.wait_stream(cur)  # an exchange starts when its producers (everything enqueued so far) are done
        _lib.check(_lib.load().rgbx_paced_copy_f32(self._paced_src.data_ptr(), self._paced_dst.data_ptr(), n,
                                                   c["workgroups"], int(c["nontemporal"]), self._link.cuda_stream),
                   "rgbx_paced_copy_f32")
        done = torch.cuda.Event()
        done.record(self._link)
        return done

    def all_reduce_sum_(self, t):
        return t.mul_(self.world)  # as if every rank had contributed this rank's share

    def all_reduce_sum_async(self, t):
        t.mul_(self.world)
        return self._traced("all-reduce", t.numel() * t.element_size())

    def all_reduce_max_(self, t):
        return t

    def self_test_views(self, device):
        return True

    def barrier(self):
        pass

    def abort(self, exc, where=""):
        pass  # one process: the caller re-raises
