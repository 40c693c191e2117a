"""bench_supervisor.py -- keeps `bench.py --gpus N` (N > 1) value-bearing when a rank dies or hangs.

Why it exists: the failure mode of RCCL on first contact with a node is not a Python exception.  A collective that
never completes is torn down by the ProcessGroupNCCL watchdog -- a C++ abort (SIGABRT) -- and a rank can as well be
killed by the kernel or sit in a device-side wait for ever.  None of that can be caught inside the rank.  So every
process the launcher starts (`python -m torch.distributed.run ... bench.py --gpus N`, the driver's form, or the ranks
`python bench.py --gpus N` launches itself) is only a SUPERVISOR: it holds no GPU context, starts the real rank as a
child process (`H2GCN_BENCH_WORKER=1`) and watches it.  The supervisors agree on what happened through a key-value
store (the launcher's own TCPStore on MASTER_ADDR:MASTER_PORT when torch.distributed.run provides one, otherwise one
hosted by rank 0's supervisor) and walk down a ladder of attempts, each a fresh set of worker processes on a fresh
rendezvous port:

    requested      what was asked for: first-contact table, calibration over every exchange x chunking, K timed steps
    exclude        (at most twice) the same sweep WITHOUT the exchange form that was in flight when the attempt died
    conservative   one ncclAllGather per chunk, 2 chunks, no first-contact table, no calibration sweep
    no_rccl        no RCCL at all: gloo bootstrap + the library's own IPC copy-kernel exchange, 2 chunks

The first attempt whose rank 0 produces a line with a value wins -- that line exists only after the max-over-ranks of
the K timed steps, so a rank that gets stuck or dies in the tear-down after it no longer matters (the workers are given
a few seconds, then removed).  Rank 0's supervisor prints that line (stdout carries exactly one line), with what went
wrong before it under `config.diagnostics.attempts` / `.first_attempt`.  A worker that dies between the timed region and
its print leaves the measurement on record; the supervisor rebuilds the line from it.  If the ladder runs out, rank 0's
supervisor prints ONE error line itself, carrying every calibration entry that did complete as `partial`.  A worker is
given a wall-clock budget; when one rank's worker fails, the others are taken down after a short grace period instead
of waiting for a collective time-out.

Nothing here touches the measurement: the timed region, the barriers and the max-over-ranks live in the worker.
"""
from __future__ import annotations

import datetime
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

METRIC = "aggregated edges/sec (1+2-hop SpMM)"
USAGE_ERROR = 64   # bench.py's exit code for a command line that cannot run under any schedule (EX_USAGE)

#: the rungs of the ladder: what the workers of an attempt are told (environment overrides)
RUNG_REQUESTED = {"rung": "requested", "name": "as requested", "env": {}, "excluded": []}
RUNG_CONSERVATIVE = {"rung": "conservative", "name": "conservative: ncclAllGather, 2 chunks, no calibration sweep", "excluded": [],
                     "env": {"H2GCN_BENCH_FORCE_EXCHANGE": "allgather", "H2GCN_BENCH_FORCE_CHUNKS": "2", "H2GCN_BENCH_SKIP_DRY": "1"}}
RUNG_NO_RCCL = {"rung": "no_rccl", "name": "no RCCL: gloo bootstrap + IPC copy-kernel exchange, 2 chunks", "excluded": [],
                "env": {"H2GCN_BENCH_FORCE_EXCHANGE": "ipc_kernel", "H2GCN_BENCH_FORCE_CHUNKS": "2", "H2GCN_BENCH_SKIP_DRY": "1",
                        "H2GCN_DIST_BACKEND": "gloo"}}
MAX_RUNGS = 5            # requested + up to 2 exclusions + conservative + no-RCCL


def next_rung(history):
    """What to try after the attempts in `history` (rank 0 decides, the others are told).  If the worker's record shows WHICH
    exchange family was in flight when the attempt died (`in_flight_family`: the last candidate that was started and never
    finished -- e.g. the grouped send/recv form on its first contact with a second device), the sweep is repeated without that
    family, so that the best of the remaining schedules is still found (at most twice); then the conservative rung; then the
    rung that does not touch RCCL.  An attempt that died with the plain ncclAllGather in flight skips straight to the last."""
    tried = [h["rung"] for h in history]
    last = history[-1]
    fam, excluded = last.get("in_flight_family"), set(last.get("excluded") or [])
    if last["rung"] in ("requested", "exclude") and fam and fam != "allgather" and fam not in excluded and tried.count("exclude") < 2:
        ex = sorted(excluded | {fam})
        env = {"H2GCN_BENCH_EXCLUDE_EXCHANGES": ",".join(ex), "H2GCN_BENCH_SKIP_DRY": "1"}
        # economy: what earlier attempts already timed is on record -- keep the fastest of it (it has to be set up again to be
        # usable) and do not re-time the rest; candidates that failed their correctness check stay out as well
        cal = [e for h in history for e in h.get("calibration", []) if "calibration" in e]
        timed = {e["calibration"]: e["ms_per_step"] for e in cal if "ms_per_step" in e}
        usable = {k_: v for k_, v in timed.items() if k_.split("/")[0] not in ex}      # (the excluded form's own timings are moot)
        if usable:
            best = min(usable, key=usable.get)
            skip = sorted((set(timed) | {e["calibration"] for e in cal if "rejected" in e}) - {best})
            if skip:
                env["H2GCN_BENCH_SKIP_CANDIDATES"] = ",".join(skip)
                env["H2GCN_BENCH_EARLIER_TIMINGS"] = json.dumps({k_: timed[k_] for k_ in skip if k_ in timed})
        return {"rung": "exclude", "excluded": ex,
                "name": f"as requested without the exchange form(s) {', '.join(ex)} (in flight when an attempt died), earlier timings kept; "
                        "no first-contact table",
                "env": env}
    if "conservative" not in tried and "no_rccl" not in tried and fam != "allgather":
        return RUNG_CONSERVATIVE
    if "no_rccl" not in tried:
        return RUNG_NO_RCCL
    return None


def in_flight_family(entries):
    """Exchange family of the last candidate the worker STARTED (first-contact table or calibration) and never finished."""
    open_key = None
    for e in entries:
        if "starting" in e:
            open_key = e["starting"]
        elif e.get("finished") == open_key:
            open_key = None
    return None if open_key is None else str(open_key).split("/")[0]


def _free_port() -> int:
    """A TCP port for a rendezvous.  NOT one of the kernel's ephemeral ports (32768-60999 on Linux): those are what RCCL's and
    gloo's own listening sockets and every outgoing connection of the job get, so a port that is free when probed can be taken a
    moment later (seen once in 3 suite runs: EADDRINUSE on the store's port).  A random port below that range, verified by binding."""
    import random

    rng = random.Random(os.getpid() ^ int(time.time() * 1e6))
    for _ in range(200):
        port = rng.randrange(20000, 32000)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", port))
                return port
            except OSError:
                continue
    with socket.socket() as sk:          # (every probe taken: let the kernel choose after all)
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _value_line(text: str):
    """The last stdout line that parses as a bench line (dict with "metric"); None if there is none."""
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                obj = json.loads(ln)
            except ValueError:
                continue
            if isinstance(obj, dict) and "metric" in obj:
                return obj
    return None


def _progress_entries(path: Path):
    """What rank 0's worker put on record while it ran (calibration entries, the first-contact table): JSON lines."""
    out = []
    try:
        for ln in path.read_text().splitlines():
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    except OSError:
        pass
    return out


class _Store:
    """The supervisors' key-value store: a client of the launcher's TCPStore when there is one (torch.distributed.run sets
    TORCHELASTIC_USE_AGENT_STORE=True and serves it on MASTER_PORT), else hosted by rank 0's supervisor on MASTER_PORT
    (ranks started by hand or by a test).  Keys are prefixed; nothing of the workers' rendezvous ever goes through it."""

    def __init__(self, rank: int, world: int, timeout_s: float):
        import torch.distributed as dist

        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ["MASTER_PORT"])
        td = datetime.timedelta(seconds=timeout_s)
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
        self.hosted = (not agent) and rank == 0
        base = dist.TCPStore(host, port, None, self.hosted, timeout=td, wait_for_workers=False)
        self._s = dist.PrefixStore("h2gcn_bench/", base)
        self._td = datetime.timedelta

    def set(self, key: str, value) -> None:
        self._s.set(key, str(value))

    def has(self, key: str) -> bool:
        return bool(self._s.check([key]))

    def get(self, key: str, timeout_s: float):
        """Value of `key`, or None if it does not appear within `timeout_s`.  Polled (never a blocking store wait): the
        supervisor must stay responsive to the launcher's SIGTERM."""
        deadline = time.monotonic() + max(timeout_s, 0.0)
        while True:
            try:
                if self._s.check([key]):
                    return self._s.get(key).decode()
            except Exception:  # noqa: BLE001 -- the store is gone (launcher died): same as "never appeared"
                return None
            if time.monotonic() >= deadline:
                return None
            time.sleep(0.05)


class _Worker:
    """One rank's real bench process, in its own process group (so that everything it started dies with it)."""

    def __init__(self, cmd, env, stdout_path: Path):
        self.out_path = stdout_path
        self._out = open(stdout_path, "w")
        self.p = subprocess.Popen(cmd, env=env, stdout=self._out, start_new_session=True)

    def poll(self):
        return self.p.poll()

    def kill(self, grace_s: float = 3.0) -> int:
        """SIGTERM to the worker's process group, SIGKILL after `grace_s`; returns the exit code."""
        if self.p.poll() is None:
            for sig, wait in ((signal.SIGTERM, grace_s), (signal.SIGKILL, 10.0)):
                try:
                    os.killpg(self.p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass
                try:
                    self.p.wait(timeout=wait)
                    break
                except subprocess.TimeoutExpired:
                    continue
        self._out.close()
        return self.p.returncode if self.p.returncode is not None else -9

    def peek_stdout(self) -> str:
        """What the (running) worker has written so far."""
        try:
            return self.out_path.read_text()
        except OSError:
            return ""

    def stdout(self) -> str:
        try:
            self._out.close()
        except Exception:  # noqa: BLE001
            pass
        try:
            return self.out_path.read_text()
        except OSError:
            return ""


def _describe_rc(rc) -> str:
    if rc is None:
        return "still running"
    if rc < 0:
        try:
            return f"killed by {signal.Signals(-rc).name}"
        except ValueError:
            return f"killed by signal {-rc}"
    return f"exit code {rc}"


def supervise(argv, rank: int, world: int) -> int:
    """Run this rank's worker(s) down the ladder; returns the exit code of the supervisor (0 = a value line was printed by
    rank 0's supervisor).  `argv` = bench.py's own command line (without the program name)."""
    # wall-clock limit of the first attempt (a healthy 8-rank run takes one to two minutes; a collective that never completes is
    # torn down by the watchdog after H2GCN_DIST_TIMEOUT_S = 60 s: the budget is the backstop behind that) and of a retry
    budget_first = float(os.environ.get("H2GCN_BENCH_ATTEMPT_BUDGET_S", "420"))
    budget_retry = float(os.environ.get("H2GCN_BENCH_RETRY_BUDGET_S", os.environ.get("H2GCN_BENCH_ATTEMPT_BUDGET_S", "240")))
    budget0 = budget_first
    grace = float(os.environ.get("H2GCN_BENCH_PEER_FAILURE_GRACE_S", "8"))       # how long a worker outlives a failed peer
    teardown = float(os.environ.get("H2GCN_BENCH_TEARDOWN_GRACE_S", "20"))       # ... and how long it may take to exit after the line
    n_attempts = max(1, min(MAX_RUNGS, int(os.environ.get("H2GCN_BENCH_MAX_ATTEMPTS", str(MAX_RUNGS)))))
    total_budget = float(os.environ.get("H2GCN_BENCH_TOTAL_BUDGET_S", "1500"))      # all rungs together
    t_start = time.monotonic()
    try:
        store = _Store(rank, world, timeout_s=90.0)      # (time to find the store: the supervisors start within seconds of each other)
    except Exception as e:  # noqa: BLE001 -- no key-value store to coordinate through: run the rank unsupervised rather than not at all
        print(json.dumps({"supervisor": f"rank {rank}: no store on {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')} "
                                        f"({type(e).__name__}: {e}); running the rank without supervision"}), file=sys.stderr, flush=True)
        env = dict(os.environ, H2GCN_BENCH_WORKER="1")
        cmd0 = [sys.executable, str(Path(__file__).resolve().parent / "bench.py")] + list(argv)
        os.execve(cmd0[0], cmd0, env)
    tmp = Path(tempfile.mkdtemp(prefix=f"h2gcn_bench_r{rank}_", dir="/tmp"))
    worker_cmd = os.environ.get("H2GCN_BENCH_WORKER_CMD")     # test hook: a JSON list that replaces `python bench.py <argv>`
    cmd = json.loads(worker_cmd) if worker_cmd else [sys.executable, str(Path(__file__).resolve().parent / "bench.py")] + list(argv)
    current = {"w": None}

    printed = {"line": False}        # stdout carries exactly ONE line: whoever prints it sets this first

    def on_signal(signum, _frame):   # the launcher is taking the job down (time limit, ^C): no orphans, and still one line
        if current["w"] is not None:
            current["w"].kill(1.0)
        if rank == 0 and not printed["line"]:
            printed["line"] = True
            # os.write, not print: the main thread may be inside a print of its own (a signal handler must not re-enter the
            # buffered stream); one write call = one line
            txt = json.dumps({"metric": METRIC, "value": None, "unit": "edges/s", "n_gpus": world,
                              "error": f"supervisor received {signal.Signals(signum).name}",
                              "partial": _progress_entries(tmp / "progress.jsonl")}) + "\n"
            try:
                sys.stdout.flush()
            except Exception:  # noqa: BLE001
                pass
            os.write(1, txt.encode())
        os._exit(128 + signum)

    for s_ in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(s_, on_signal)

    def report_failure(k_, why_):
        """This rank's worker failed in attempt k_: its own record (with the time: the FIRST failure is what matters, a later one
        is usually its consequence -- a peer's connection reset) plus the flag the other supervisors watch."""
        store.set(f"a{k_}/failed_by{rank}", json.dumps({"t": time.time(), "why": f"rank {rank}: {why_}"}))
        store.set(f"a{k_}/failed", "1")

    def first_failure(k_):
        recs = []
        for q in range(world):
            v = store.get(f"a{k_}/failed_by{q}", 0.0)
            if v:
                recs.append(json.loads(v))
        return min(recs, key=lambda r: r["t"])["why"] if recs else None

    history = []          # rank 0: one entry per failed attempt
    final = 1
    for k in range(n_attempts):
        if k == 0:
            rung = RUNG_REQUESTED
        else:
            told = store.get(f"a{k}/rung", 60.0)
            if told is None:
                break
            rung = json.loads(told)
        name, overrides = rung["name"], rung["env"]
        remaining = total_budget - (time.monotonic() - t_start)
        budget0 = max(30.0, min(budget_first if k == 0 else budget_retry, remaining))
        # a fresh rendezvous port per attempt: the dead attempt's keys (ncclUniqueId, gloo addresses) must not be found
        if rank == 0:
            store.set(f"a{k}/port", _free_port())
        port = store.get(f"a{k}/port", 120.0)
        if port is None:
            history.append({"attempt": k, "schedule": name, "error": "rank 0's supervisor never published the rendezvous port"})
            break
        env = dict(os.environ)
        env.update(overrides)
        env.update(H2GCN_BENCH_WORKER="1", H2GCN_BENCH_ATTEMPT=str(k), MASTER_PORT=str(port),
                   H2GCN_BENCH_PROGRESS=str(tmp / "progress.jsonl"), H2GCN_BENCH_SCRATCH=str(tmp))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)     # the workers rendezvous among themselves: rank 0's worker hosts the store
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env.get("OMP_NUM_THREADS") == "1" and "H2GCN_BENCH_OMP_NUM_THREADS" not in env:
            # torch.distributed.run exports OMP_NUM_THREADS=1 per rank; the reported CPU baselines (rank 0, after the timed region)
            # are labelled with the threads they use, so give every worker its share of the host instead
            env["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))
        elif "H2GCN_BENCH_OMP_NUM_THREADS" in env:
            env["OMP_NUM_THREADS"] = env["H2GCN_BENCH_OMP_NUM_THREADS"]
        if k > 0:
            env["H2GCN_DIST_TIMEOUT_S"] = os.environ.get("H2GCN_BENCH_RETRY_DIST_TIMEOUT_S", os.environ.get("H2GCN_DIST_TIMEOUT_S", "60"))
        w = _Worker(cmd, env, tmp / f"attempt{k}.stdout")
        current["w"] = w
        t0 = time.monotonic()
        why, peer_failed_at, done_at, line = None, None, None, None

        def publish(line_):
            """rank 0: the measurement is complete the moment its line exists (it is printed after the max-over-ranks of the
            timed region) -- whatever happens to a rank during tear-down no longer matters."""
            if history:
                diag = line_.setdefault("config", {}).setdefault("diagnostics", {})
                diag["attempts"] = [{"attempt": h["attempt"], "schedule": h["schedule"], "result": h.get("first_failure") or h.get("error")}
                                    for h in history] + [{"attempt": k, "schedule": name, "result": "ok"}]
                diag["first_attempt"] = history[0]          # in full: per-rank outcome + every calibration entry it completed
                if len(history) > 1:
                    diag["failed_attempts"] = history[1:]
            store.set(f"a{k}/verdict", "ok")
            if not printed["line"]:
                printed["line"] = True
                print(json.dumps(line_), flush=True)

        while True:
            rc = w.poll()
            if rc is not None:
                break
            now = time.monotonic()
            if done_at is None:
                if rank == 0:
                    cand = _value_line(w.peek_stdout())
                    if cand is not None and cand.get("value") is not None:
                        line, done_at = cand, now
                        publish(line)
                elif store.has(f"a{k}/verdict") and store.get(f"a{k}/verdict", 0.1) == "ok":
                    done_at = now
            if done_at is not None:
                if now - done_at > teardown:      # a rank stuck in its final barrier / destroy_process_group: the result stands
                    rc = w.kill()
                    break
                time.sleep(0.1)
                continue
            if now - t0 > budget0:
                why = f"no result within the attempt's budget of {budget0:.0f} s"
                report_failure(k, why)
                rc = w.kill()
                break
            if peer_failed_at is None and store.has(f"a{k}/failed"):
                peer_failed_at = now
            if peer_failed_at is not None and now - peer_failed_at > grace:
                why = "taken down after a peer failed: " + (first_failure(k) or "?")
                rc = w.kill()
                break
            time.sleep(0.1)
        current["w"] = None
        if done_at is not None:            # success was declared while the worker was still tearing down
            final = 0
            break
        if rc != 0 and why is None:
            why = _describe_rc(rc)
            report_failure(k, why)
        store.set(f"a{k}/rc{rank}", rc)
        if rank != 0:
            verdict = store.get(f"a{k}/verdict", budget0 + 120.0)
            if verdict == "ok":
                final = 0
                break
            if verdict is None or verdict == "fail":
                break
            continue
        # rank 0: its worker is gone.  A value line = success (see publish); otherwise collect the exit codes for the record
        line = _value_line(w.stdout())
        if line is not None and line.get("value") is not None:
            publish(line)
            final = 0
            break
        measured = [e["measured"] for e in _progress_entries(tmp / "progress.jsonl") if e.get("attempt", k) == k and "measured" in e]
        if measured:
            # the K timed steps completed on every rank (the entry is written after the max-over-ranks) and the worker died in
            # what follows (diagnostics, CPU legs): the measurement stands, the line is rebuilt from the record
            line = dict(measured[-1])
            line["roofline"] = None
            line["cpu_baseline"] = None
            line.setdefault("config", {})["diagnostics"] = {
                "rebuilt_by_supervisor": f"rank 0's worker ended ({_describe_rc(rc)}) after the timed region and before printing its line; "
                                         "value / ms_per_step are the worker's own max-over-ranks timing of the K steps",
                "calibration": [e for e in _progress_entries(tmp / "progress.jsonl") if e.get("attempt", k) == k and "calibration" in e]}
            publish(line)
            final = 0
            break
        rcs = {0: rc}
        for q in range(1, world):
            v = store.get(f"a{k}/rc{q}", grace + 30.0)     # the others follow within the peer-failure grace period
            rcs[q] = None if v is None else int(v)
        # what the store saw first is often a SYMPTOM (a peer's connection reset); a rank that died of a signal nobody here sent
        # (SIGABRT: the watchdog; SIGSEGV; the OOM killer's SIGKILL arrives as -9 too, but so do our own kills) is the cause
        died = [f"rank {q}: {_describe_rc(v)}" for q, v in rcs.items() if v is not None and v < 0 and -v not in (signal.SIGTERM, signal.SIGKILL)]
        attempt_entries = [e for e in _progress_entries(tmp / "progress.jsonl") if e.get("attempt", k) == k]
        entry = {"attempt": k, "schedule": name, "rung": rung["rung"], "excluded": rung.get("excluded", []),
                 "in_flight_family": in_flight_family(attempt_entries),
                 "ranks": {str(q): ("ok" if v == 0 else _describe_rc(v)) for q, v in rcs.items()},
                 "first_failure": died[0] if died else first_failure(k),
                 "calibration": [e for e in attempt_entries if "starting" not in e and "finished" not in e]}
        if line is not None and line.get("error"):
            entry["error_line"] = line["error"]
        history.append(entry)
        print(json.dumps({"supervisor": f"attempt {k} ({name}) failed after {time.monotonic() - t0:.0f} s", "ranks": entry["ranks"],
                          "first_failure": entry["first_failure"]}), file=sys.stderr, flush=True)
        # exit code 64 on any rank = the command line / the node cannot run this at all (bench.py's usage errors, fewer GPUs than
        # ranks): no schedule will fix that
        nxt = None
        out_of_time = total_budget - (time.monotonic() - t_start) < 45.0
        if k < n_attempts - 1 and not out_of_time and not any(v == USAGE_ERROR for v in rcs.values()):
            nxt = next_rung(history)
        if nxt is not None:
            store.set(f"a{k + 1}/rung", json.dumps(nxt))
        store.set(f"a{k}/verdict", "retry" if nxt is not None else "fail")
        if nxt is None:
            break
    if rank == 0 and final != 0 and not printed["line"]:
        printed["line"] = True
        partial = [e for h in history for e in h.get("calibration", [])]
        print(json.dumps({"metric": METRIC, "value": None, "unit": "edges/s", "n_gpus": world,
                          "error": f"every attempt failed ({len(history)}): "
                                   + "; ".join(f"[{h['attempt']}] {h.get('error_line') or h.get('first_failure') or h.get('error') or h.get('ranks')}"
                                               for h in history),
                          "attempts": history, "partial": partial}), flush=True)
    if os.environ.get("H2GCN_BENCH_KEEP_SCRATCH") != "1":      # worker stdout, progress records, RCCL's debug files
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    # a store hosted by rank 0's supervisor must outlive the other supervisors' last look at it
    # (a peer may need teardown + kill grace before it says bye; a store that is gone already is nobody's error -- the line is out)
    try:
        store.set(f"bye{rank}", 1)
        if store.hosted:
            for q in range(1, world):
                store.get(f"bye{q}", teardown + 13.0 + 12.0)
    except Exception:  # noqa: BLE001
        pass
    return final
