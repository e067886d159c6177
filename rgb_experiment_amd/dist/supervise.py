"""Supervision of the ranks of a multi-GPU benchmark run (bench.py --gpus N; new capability — the reference is
single-device, itexperiments.py:246, so there is nothing of it to mirror here).

Every rank process that torch.distributed.run starts is a SUPERVISOR: it makes no GPU call, starts the actual worker
(the same script with RGBX_SUPERVISED=1) as a child in its own process group, and watches it:

  * the worker touches a heartbeat file at every milestone (imports done, process group up, graph built, each step);
    a worker that stays silent longer than the stall limit, or runs past the deadline, is killed with its group;
  * a supervisor whose worker failed says so in a file of the run's shared directory (/tmp, one node); every other
    supervisor that sees the file kills its own worker at once, so one rank's error ends the attempt for all ranks
    instead of leaving the peers waiting in a collective;
  * then all supervisors start a FRESH set of workers with the next, more conservative set of flags (ATTEMPTS) on a
    fresh rendezvous port chosen by rank 0's supervisor. A process that has touched the GPU is never re-executed
    or reused: each attempt is new processes;
  * rank 0's supervisor relays the surviving JSON line with a "launcher" object naming the attempt that produced it
    and, if it was not the first, what failed before it.

stdlib only: importing this module must not initialise anything."""
import glob
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

# flags appended to the command line as typed (argparse: the last occurrence wins), most ambitious first
ATTEMPTS = [
    ("as asked", []),
    ("the scheme as asked, through the modules: no fused schedule, no step computed ahead, no task split, sequential "
     "evals, one-piece exchanges",
     ["--no-fused", "--no-interleave", "--task-split", "off", "--pieces", "1", "--pieces-in", "1"]),
    ("conservative: val and test forward one after the other, one-piece transpose exchange",
     ["--no-fused", "--no-interleave", "--task-split", "off", "--pieces", "1", "--exchange", "reshard"]),
    ("halo exchange, sequential evals",
     ["--no-fused", "--no-interleave", "--task-split", "off", "--pieces", "1", "--exchange", "halo"]),
    ("replicate: no activation exchange, all-reduces only",
     ["--no-fused", "--no-interleave", "--task-split", "off", "--exchange", "replicate"]),
]


def _env_s(name, default):
    try:
        return float(os.environ.get(name, default))
    except ValueError:
        return float(default)


def limits():
    """(deadline per attempt, stall limit once the worker has reported in, limit for the first report) in seconds.
    The first `import torch` on a fresh box can take two minutes, hence the separate first limit. A healthy attempt
    at the benchmark's size is through in about a minute (setup_s of the rehearsal records: < 10 s after the import)."""
    return (_env_s("RGBX_LAUNCH_DEADLINE_S", 300), _env_s("RGBX_LAUNCH_STALL_S", 120),
            _env_s("RGBX_LAUNCH_IMPORT_S", 240))


def total_budget():
    """Seconds the WHOLE supervised run may take, all attempts together (RGBX_LAUNCH_TOTAL_S): the driver ends a bench
    command after 600 s, and rank 0's line — a benchmark line or the diagnostic one naming what failed — must be out
    before that. No attempt starts with less than MIN_ATTEMPT_S of it left, and every attempt's deadline is cut to what
    is left."""
    return _env_s("RGBX_LAUNCH_TOTAL_S", 540)


MIN_ATTEMPT_S = 75.0   # an attempt that cannot even import torch and build the graph is not started
LINE_RESERVE_S = 15.0  # kept back for ending the workers and printing rank 0's line


def run_key():
    """Identifies this run on this node: all rank processes are children of the one torch.distributed.run agent."""
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time of the agent
    except (OSError, IndexError):
        pass
    return f"{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}"


def shared_dir():
    d = os.path.join(tempfile.gettempdir(), "rgbx_bench_" + run_key())
    os.makedirs(d, exist_ok=True)
    return d


def _write(path, text):
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _wait_for(paths, timeout):
    """First existing path of `paths` (glob patterns allowed) within `timeout` seconds, else None."""
    t_end = time.monotonic() + timeout
    while True:
        for p in paths:
            hit = glob.glob(p)
            if hit:
                return hit[0]
        if time.monotonic() > t_end:
            return None
        time.sleep(0.05)


def _mtime(path):
    try:
        return os.stat(path).st_mtime
    except OSError:
        return None


def _await_rank0(d, k, paths, quiet_limit, hard_limit):
    """A rank other than 0 waits for rank 0's decision: the first existing path of `paths`, for as long as rank 0 shows
    signs of life — its supervisor touches `sup.0` once a second, its worker of attempt k the heartbeat `attempt{k}.hb.0`
    — and no longer than `hard_limit` seconds. Returns the path, or None (rank 0 silent for `quiet_limit` s, or the
    hard limit passed). Rank 0 alone decides how a run ends; the others never give up on it while it is alive."""
    t0 = time.monotonic()
    while True:
        for p in paths:
            hit = glob.glob(p)
            if hit:
                return hit[0]
        now = time.monotonic()
        if now - t0 > hard_limit:
            return None
        seen = [m for m in (_mtime(os.path.join(d, "sup.0")), _mtime(os.path.join(d, f"attempt{k}.hb.0"))) if m]
        if seen and time.time() - max(seen) > quiet_limit:
            return None
        if not seen and now - t0 > quiet_limit:
            return None
        time.sleep(0.05)


def die_with_parent(sig=signal.SIGKILL):
    """preexec_fn for a child that must not outlive this process (PR_SET_PDEATHSIG): a launcher or supervisor that is
    killed outright (SIGKILL: no handler runs) would otherwise leave ranks behind that hold their GPUs."""
    def set_pdeathsig():
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(sig))  # PR_SET_PDEATHSIG = 1
        except OSError:
            pass
    return set_pdeathsig


def kill_group(proc, grace=5.0):
    """SIGTERM to the child's process group, SIGKILL after `grace` seconds."""
    if proc.poll() is not None:
        return
    for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 5.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=wait)
            return
        except subprocess.TimeoutExpired:
            continue


def beat(phase):
    """Worker side: report a milestone (no-op outside a supervised run)."""
    path = os.environ.get("RGBX_HEARTBEAT")
    if path:
        try:
            _write(path, f"{phase}\n")
        except OSError:
            pass


class Ticker:
    """Worker side: beat() from inside a hot loop at most once per `every` seconds (the file write is ~20 us; a step
    of the timed region must not pay for it every time)."""

    def __init__(self, every=1.0):
        self.every, self.last = every, 0.0
        self.on = bool(os.environ.get("RGBX_HEARTBEAT"))

    def __call__(self, phase):
        if self.on:
            now = time.monotonic()
            if now - self.last >= self.every:
                self.last = now
                beat(phase)


def test_fault(point, rank):
    """Test hook (tests/test_bench_cli.py): RGBX_TEST_FAULT="stall|raise:<rank>:<attempt>:<point>" makes that rank
    of that attempt hang or fail at the named point, to exercise the supervision end to end."""
    spec = os.environ.get("RGBX_TEST_FAULT")
    if not spec:
        return
    kind, r, attempt, where = spec.split(":")
    if int(r) == rank and attempt in ("*", os.environ.get("RGBX_ATTEMPT", "0")) and where == point:
        if kind == "stall":
            while True:
                time.sleep(3600)
        raise RuntimeError(f"injected test fault at {point} on rank {rank}")


def _last_json_line(path):
    text = _read(path) or ""
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if not lines:
        return None
    try:
        return json.loads(lines[-1])
    except ValueError:
        return None


def supervise(script, argv, rank, world, attempts=None, out=sys.stdout):
    """Run the worker for this rank under the rules above; returns the exit status of this rank process."""
    attempts = ATTEMPTS if attempts is None else attempts
    max_attempts = int(_env_s("RGBX_LAUNCH_ATTEMPTS", len(attempts)))
    attempts = attempts[:max(1, max_attempts)]
    deadline_s, stall_s, import_s = limits()
    budget_s, t_run = total_budget(), time.monotonic()
    d = shared_dir()
    child = [None]

    def on_signal(signum, _frame):  # the agent (or the driver above it) ends the run: take the worker along
        if child[0] is not None:
            kill_group(child[0], grace=2.0)
        sys.exit(128 + signum)

    for s in (signal.SIGTERM, signal.SIGINT):
        signal.signal(s, on_signal)

    agent = os.getppid()
    history = []
    for k, (name, flags) in enumerate(attempts):
        port_file = os.path.join(d, f"attempt{k}.port")
        left = budget_s - (time.monotonic() - t_run)
        if rank == 0 and k > 0 and left < _env_s("RGBX_LAUNCH_MIN_ATTEMPT_S", MIN_ATTEMPT_S):
            history.append({"attempt": k, "settings": name, "extra_flags": flags,
                            "reason": f"not started: {left:.0f} s of the run's {budget_s:g} s budget left"})
            break
        # this attempt's deadline: what is left of the run's budget, so that rank 0's line is out in time either way
        deadline_k = max(min(deadline_s, left - LINE_RESERVE_S), 5.0)
        if rank == 0:
            _write(os.path.join(d, "sup.0"), str(k))
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            _write(port_file, str(port))
        elif _await_rank0(d, max(k - 1, 0), [port_file, os.path.join(d, "gave_up")], stall_s + import_s,
                          deadline_s + 60) != port_file:
            # rank 0 has given up (its line, a diagnostic one, is out) or is gone: only then does this rank leave
            print(f"[bench supervisor rank {rank}] no rendezvous port for attempt {k}: leaving", file=sys.stderr)
            time.sleep(1.0)  # rank 0 writes `gave_up` and then prints: do not make the agent end it in between
            return 1
        port = int(_read(port_file))
        hb = os.path.join(d, f"attempt{k}.hb.{rank}")
        out_path = os.path.join(d, f"attempt{k}.out.{rank}")
        env = dict(os.environ)
        env.update({"MASTER_PORT": str(port), "RGBX_SUPERVISED": "1", "RGBX_ATTEMPT": str(k), "RGBX_HEARTBEAT": hb})
        # the workers rendezvous among themselves on the attempt's own port (rank 0's worker hosts the store): the
        # agent's store belongs to the supervisors' generation and must not be joined by a second one
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        t0 = time.monotonic()
        with open(out_path, "w") as fout:
            proc = subprocess.Popen([sys.executable, script] + list(argv) + flags, env=env, stdout=fout,
                                    start_new_session=True, preexec_fn=die_with_parent())
        child[0] = proc
        failed_glob = os.path.join(d, f"attempt{k}.failed.*")
        ok_file = os.path.join(d, f"attempt{k}.ok")
        reason = None
        sup_beat = 0.0
        while True:
            rc = proc.poll()
            if rc is not None:
                break
            if rank == 0 and time.monotonic() - sup_beat > 1.0:  # what the other ranks' supervisors watch (_await_rank0)
                sup_beat = time.monotonic()
                _write(os.path.join(d, "sup.0"), str(k))
            if os.getppid() != agent:  # the launcher above is gone (killed outright): nobody is waiting for a line
                kill_group(proc, grace=2.0)
                print(f"[bench supervisor rank {rank}] the launcher is gone: worker ended, leaving", file=sys.stderr)
                return 1
            now = time.monotonic()
            peer = glob.glob(failed_glob)
            if peer:
                reason = f"a peer failed first ({os.path.basename(peer[0])}: {(_read(peer[0]) or '').strip()[:200]})"
            elif now - t0 > deadline_k:
                reason = (f"deadline of {deadline_k:.0f} s passed (per attempt {deadline_s:g} s, run budget {budget_s:g} s; "
                          f"last milestone: {(_read(hb) or 'none').strip()})")
            else:
                try:  # heartbeat files carry wall-clock mtimes
                    quiet, limit = time.time() - os.stat(hb).st_mtime, stall_s
                except OSError:  # the worker has not reported in yet (interpreter start, `import torch`)
                    quiet, limit = now - t0, import_s
                if quiet > limit:
                    reason = f"no milestone for {quiet:.0f} s (last: {(_read(hb) or 'none').strip()})"
            if reason:
                kill_group(proc)
                rc = proc.returncode
                break
            time.sleep(0.1)
        child[0] = None
        line = _last_json_line(out_path) if rank == 0 else None
        # a worker that reported "done" (its line is out, only the teardown was left) counts even if it had to be ended
        # afterwards or left with a non-zero status: a process group that does not come down must not cost the result
        done = (_read(hb) or "").startswith("done")
        good = ((reason is None and rc == 0) or done) and (rank != 0 or line is not None)
        if good and rank != 0:
            # my worker is through; the attempt counts only if rank 0 got its line (a peer may still have failed, rank 0's
            # worker may still be busy): wait for rank 0's verdict while it is alive, and report success on `ok` alone
            hit = _await_rank0(d, k, [ok_file, os.path.join(d, f"attempt{k}.failed.0"),
                                      os.path.join(d, f"attempt{k + 1}.port"), os.path.join(d, "gave_up")],
                               stall_s + import_s, max(deadline_k - (time.monotonic() - t0), 0) + 60)
            if hit == ok_file:
                return 0
            if hit is None or hit.endswith("gave_up"):
                print(f"[bench supervisor rank {rank}] no verdict from rank 0 for attempt {k}: leaving", file=sys.stderr)
                time.sleep(1.0)
                return 1
            history.append({"attempt": k, "flags": flags, "reason": "rank 0 did not get its line"})
            continue
        if good:
            _write(ok_file, "ok")
            line["launcher"] = {"attempt": k, "settings": name, "extra_flags": flags, "attempts_made": k + 1,
                                "supervised": True, "wall_s": time.monotonic() - t0,
                                "fallback": None if not history else {"from": attempts[0][0], "failed": history}}
            print(json.dumps(line), file=out, flush=True)
            return 0
        if os.path.exists(ok_file):  # rank 0 has the line: what happened to this rank afterwards changes nothing
            return 0
        reason = reason or (f"worker exited with status {rc}" if rc != 0 else "worker printed no JSON line")
        history.append({"attempt": k, "settings": name, "extra_flags": flags, "reason": reason})
        _write(os.path.join(d, f"attempt{k}.failed.{rank}"), reason)
        print(f"[bench supervisor rank {rank}] attempt {k} ({name}) failed: {reason}", file=sys.stderr, flush=True)
        if rank != 0:  # rank 0 decides: its line (-> done) or the next attempt's port (-> again)
            hit = _await_rank0(d, k, [ok_file, os.path.join(d, f"attempt{k + 1}.port"), os.path.join(d, "gave_up")],
                               stall_s + import_s, deadline_s + 60)
            if hit == ok_file:
                return 0
            if hit is None or hit.endswith("gave_up"):
                time.sleep(1.0)  # rank 0 prints its (diagnostic) line right after `gave_up`
                return 1
    if rank == 0:
        print(json.dumps({"metric": "aggregated edges/sec (multi-GPU run failed in every attempt)", "value": None,
                          "unit": "edges/s", "n_gpus": world, "higher_is_better": True,
                          "error": "no attempt produced a benchmark line",
                          "launcher": {"attempts_made": len(attempts), "supervised": True, "failed": history}}),
              file=out, flush=True)
        _write(os.path.join(d, "gave_up"), "1")  # after the line: the other ranks leave (non-zero) only now
    return 1

