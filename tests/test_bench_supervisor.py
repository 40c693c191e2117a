"""bench_supervisor.py on CPU: the retry ladder behind `bench.py --gpus N`, driven with a scripted stand-in for the
worker (`H2GCN_BENCH_WORKER_CMD`, a test hook) -- a rank that dies with SIGABRT (what the ProcessGroupNCCL watchdog does
to a rank whose collective timed out), a rank that hangs, a tear-down that never returns.  Whatever happens, stdout of
rank 0's supervisor carries exactly ONE JSON line.  The same ladder with the real worker runs in tests/test_multirank_gpu.py."""
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

FAKE_WORKER = r'''
import json, os, sys, time
rank = int(os.environ["RANK"]); k = int(os.environ["H2GCN_BENCH_ATTEMPT"]); sc = os.environ["FAKE_SCENARIO"]
assert os.environ["H2GCN_BENCH_WORKER"] == "1" and "TORCHELASTIC_USE_AGENT_STORE" not in os.environ
def prog(o):
    if rank == 0:
        with open(os.environ["H2GCN_BENCH_PROGRESS"], "a") as f:
            f.write(json.dumps(dict(o, attempt=k)) + "\n")
def line():
    if rank == 0:
        print(json.dumps({"metric": "m", "value": 1.5 + k, "config": {"diagnostics": {
            "forced": [os.environ.get("H2GCN_BENCH_FORCE_EXCHANGE"), os.environ.get("H2GCN_BENCH_FORCE_CHUNKS"), os.environ.get("H2GCN_DIST_BACKEND")],
            "port": os.environ["MASTER_PORT"]}}}), flush=True)
time.sleep(0.2)
if sc == "ok":
    line(); sys.exit(0)
if sc in ("abort_first", "abort_twice", "abort_always"):
    if k == 0 or (k == 1 and sc != "abort_first") or sc == "abort_always":
        prog({"calibration": "allgather/2", "ms_per_step": 3.0})
        if rank == 1: os.abort()
        time.sleep(600)        # the peers sit in a collective that will never complete
    line(); sys.exit(0)
if sc == "p2p_dies":          # the grouped send/recv form takes a rank down on first contact; the other forms are fine
    assert k < 2
    if "p2p" not in os.environ.get("H2GCN_BENCH_EXCLUDE_EXCHANGES", "").split(","):
        for key, ms in (("allgather/2", 3.0), ("ipc_kernel/2", 1.0), ("allgather/1", 4.0)):
            prog({"starting": key, "stage": "calibration"}); prog({"calibration": key, "ms_per_step": ms}); prog({"finished": key, "stage": "calibration"})
        prog({"starting": "ipc_engine/1", "stage": "calibration"}); prog({"calibration": "ipc_engine/1", "rejected": "chunk 0 differs"})
        prog({"finished": "ipc_engine/1", "stage": "calibration"}); prog({"starting": "p2p/1", "stage": "calibration"})
        if rank == 1: os.abort()
        time.sleep(600)
    assert os.environ.get("H2GCN_BENCH_SKIP_DRY") == "1" and "H2GCN_BENCH_FORCE_EXCHANGE" not in os.environ
    # the fastest candidate of the dead attempt is kept, the others it had timed (and the one it had rejected) are not re-timed
    assert os.environ["H2GCN_BENCH_SKIP_CANDIDATES"] == "allgather/1,allgather/2,ipc_engine/1", os.environ.get("H2GCN_BENCH_SKIP_CANDIDATES")
    assert json.loads(os.environ["H2GCN_BENCH_EARLIER_TIMINGS"]) == {"allgather/1": 4.0, "allgather/2": 3.0}
    line(); sys.exit(0)
if sc == "hang":
    if k == 0: time.sleep(600)
    line(); sys.exit(0)
if sc == "teardown_hang":
    line(); time.sleep(600)
if sc == "measured_then_die":
    prog({"calibration": "allgather/2", "ms_per_step": 3.0})
    prog({"measured": {"metric": "m", "value": 7.0, "steps": 3}})
    if rank == 0: os.abort()
    time.sleep(600)
if sc == "usage":
    if rank == 0: print(json.dumps({"metric": "m", "value": None, "error": "bad flag"}), flush=True)
    sys.exit(64)
'''


def _free_port():
    """A rendezvous port BELOW the kernel's ephemeral range (see bench_supervisor._free_port: ephemeral ports are what RCCL's and
    gloo's own sockets get, so a probed-free one can be gone a moment later)."""
    sys.path.insert(0, str(ROOT))
    from bench_supervisor import _free_port as pick
    return pick()


def _env(tmp_path, scenario, **extra):
    fake = tmp_path / "fake_worker.py"
    fake.write_text(FAKE_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "H2GCN_BENCH_WORKER")}
    env.update(FAKE_SCENARIO=scenario, H2GCN_BENCH_WORKER_CMD=json.dumps([sys.executable, str(fake)]),
               H2GCN_BENCH_ATTEMPT_BUDGET_S="10", H2GCN_BENCH_PEER_FAILURE_GRACE_S="0.5", H2GCN_BENCH_TEARDOWN_GRACE_S="1")
    env.update(extra)
    return env


def _run(tmp_path, scenario, world=3, **extra):
    """`world` supervisors started the way a launcher starts ranks (RANK / WORLD_SIZE / MASTER_* in the environment)."""
    port = _free_port()
    procs = []
    t0 = time.time()
    for r in range(world):
        env = dict(_env(tmp_path, scenario, **extra), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for o in outs[1:]:
        assert o[0].strip() == "", o[0]                       # only rank 0's supervisor writes to stdout
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1, (outs[0][0], outs[0][1][-2000:])  # ... and exactly one line
    return json.loads(lines[0]), [p.returncode for p in procs], time.time() - t0, port


def test_clean_run_passes_the_workers_line_through(tmp_path):
    line, rcs, _, port = _run(tmp_path, "ok", world=2)
    assert line["value"] == 1.5 and rcs == [0, 0]
    assert "attempts" not in line["config"]["diagnostics"]
    assert line["config"]["diagnostics"]["port"] != str(port)      # the workers rendezvous on their own port, not the launcher's


def test_a_rank_that_aborts_costs_one_attempt_not_the_line(tmp_path):
    """Rank 1 dies with SIGABRT after the first timed candidate; its peers would wait in their collective.  They are taken
    down, the conservative schedule runs on a fresh rendezvous, and the line carries what the first attempt had measured."""
    line, rcs, elapsed, _ = _run(tmp_path, "abort_first")
    assert line["value"] == 2.5 and rcs == [0, 0, 0] and elapsed < 60
    diag = line["config"]["diagnostics"]
    assert diag["forced"] == ["allgather", "2", None]
    first = diag["first_attempt"]
    assert first["ranks"]["1"] == "killed by SIGABRT" and "SIGABRT" in first["first_failure"]
    assert first["calibration"] == [{"calibration": "allgather/2", "ms_per_step": 3.0, "attempt": 0}]
    assert [h["result"] for h in diag["attempts"]][-1] == "ok" and len(diag["attempts"]) == 2


def test_the_form_in_flight_when_an_attempt_died_is_left_out_of_the_next_sweep(tmp_path):
    """Rank 1 dies while the grouped send/recv candidate is in flight (started, never finished, in rank 0's record): the next
    rung repeats the SWEEP without that form -- the best of the remaining schedules is still found -- instead of falling back to
    the conservative single schedule."""
    line, rcs, _, _ = _run(tmp_path, "p2p_dies")
    assert line["value"] == 2.5 and rcs == [0, 0, 0]
    diag = line["config"]["diagnostics"]
    assert diag["forced"] == [None, None, None]                       # a sweep, not a forced schedule
    first = diag["first_attempt"]
    assert first["in_flight_family"] == "p2p" and first["rung"] == "requested"
    assert [e["calibration"] for e in first["calibration"]] == ["allgather/2", "ipc_kernel/2", "allgather/1", "ipc_engine/1"]
    assert "p2p" in diag["attempts"][-1]["schedule"] and diag["attempts"][-1]["result"] == "ok"


def test_second_fallback_avoids_rccl_altogether(tmp_path):
    line, rcs, _, _ = _run(tmp_path, "abort_twice")
    assert line["value"] == 3.5 and rcs == [0, 0, 0]
    assert line["config"]["diagnostics"]["forced"] == ["ipc_kernel", "2", "gloo"]
    assert len(line["config"]["diagnostics"]["failed_attempts"]) == 1


def test_every_attempt_failing_is_one_error_line_with_what_was_measured(tmp_path):
    line, rcs, _, _ = _run(tmp_path, "abort_always")
    assert line["value"] is None and "every attempt failed (3)" in line["error"] and rcs[0] != 0
    assert [e["attempt"] for e in line["partial"]] == [0, 1, 2] and all(e["calibration"] == "allgather/2" for e in line["partial"])
    assert len(line["attempts"]) == 3


def test_a_rank_that_hangs_is_bounded_by_the_attempt_budget(tmp_path):
    line, rcs, elapsed, _ = _run(tmp_path, "hang")
    assert line["value"] == 2.5 and rcs == [0, 0, 0] and elapsed < 60
    assert "budget" in line["config"]["diagnostics"]["first_attempt"]["first_failure"]


def test_a_teardown_that_never_returns_does_not_cost_the_result(tmp_path):
    """The line exists only after the max-over-ranks of the timed steps: a rank stuck in its final barrier /
    destroy_process_group is removed after a few seconds and the measurement stands."""
    line, rcs, elapsed, _ = _run(tmp_path, "teardown_hang")
    assert line["value"] == 1.5 and rcs == [0, 0, 0] and elapsed < 60


def test_a_worker_that_dies_after_the_timed_region_leaves_its_measurement(tmp_path):
    line, rcs, _, _ = _run(tmp_path, "measured_then_die")
    assert line["value"] == 7.0 and line["steps"] == 3 and rcs == [0, 0, 0]
    assert "rebuilt_by_supervisor" in line["config"]["diagnostics"] and line["config"]["diagnostics"]["calibration"]


def test_a_usage_error_is_not_retried(tmp_path):
    line, rcs, _, _ = _run(tmp_path, "usage", world=2)
    assert line["value"] is None and "bad flag" in line["error"] and len(line["attempts"]) == 1 and rcs[0] != 0


@pytest.mark.parametrize("scenario", ["abort_first", "abort_always"])
def test_under_torch_distributed_run(tmp_path, scenario):
    """The driver's launch form: `python -m torch.distributed.run ... bench.py --gpus N`.  The supervisors talk through the
    launcher's own store (TORCHELASTIC_USE_AGENT_STORE); stdout of the whole job is ONE line -- also when every rung fails and the
    launcher tears the remaining supervisors down with SIGTERM while rank 0's has already printed its error line."""
    env = _env(tmp_path, scenario)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-3000:])
    line = json.loads(lines[0])
    if scenario == "abort_first":
        assert r.returncode == 0
        assert line["value"] == 2.5 and line["config"]["diagnostics"]["first_attempt"]["ranks"]["1"] == "killed by SIGABRT"
    else:
        assert r.returncode != 0 and line["value"] is None and len(line["attempts"]) == 3 and len(line["partial"]) == 3
