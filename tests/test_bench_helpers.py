"""CPU suite: bench.py's byte models and its refusal to run without a GPU (no CPU fallback anywhere)."""
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_and_compulsory_bytes():
    b = _bench()
    nnz, n, d = [119979897, 119978572], 2_400_000, 128
    # 520 B per aggregated edge at d = 128, + row pointers per hop, + one write of Y
    want = sum(z * 520 + (n + 1) * 8 for z in nnz) + n * 2 * d * 4
    assert b.algorithmic_bytes(nnz, n, d, 2) == want == 127274403896
    assert b.compulsory_bytes(nnz, n, n, d, 2) == sum(z * 8 + (n + 1) * 8 for z in nnz) + n * d * 4 + n * 2 * d * 4
    assert b.HBM_PEAK_GBPS == 8000.0
    assert b.pmc_traffic("products", 128, 1, 0, 1)["bytes_per_launch"] > 1e11   # committed PMC summary (+ its source)
    assert b.pmc_traffic("products", 128, 1, 0, 8) is None           # never profiled -> null, not a guess


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)


RCCL_LOG = """\
host:123:123 [0] NCCL INFO NCCL version 2.22.3+hip6.4 HEAD:abc
host:123:140 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7
host:123:140 [0] NCCL INFO Channel 01/16 :    0   1   2   3   4   5   6   7
host:123:140 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1
host:123:140 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC
host:123:140 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC
host:123:140 [0] NCCL INFO Channel 00/0 : 0[0] -> 2[2] via SHM/direct/direct
host:123:140 [0] NCCL INFO Connected all rings
host:123:140 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer
host:123:140 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 nvmlDev 0 busId 5000 commId 0xabc - Init COMPLETE
host:123:140 [0] NCCL INFO AllGather: 39321600 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..15}
host:123:140 [0] NCCL INFO AllGather: 39321600 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..15}
host:123:140 [0] NCCL INFO AllGather: 19660800 Bytes -> Algo RING proto LL128 channel{Lo..Hi}={0..7}
host:123:140 [0] NCCL INFO AllReduce: 16 Bytes -> Algo TREE proto LL channel{Lo..Hi}={0..0}
"""


def test_rccl_log_summary(tmp_path, monkeypatch):
    """`config.diagnostics.rccl` of an N > 1 line: <= 10 strings out of RCCL's per-process debug file -- version, communicator
    size, channels, transports (P2P/IPC = xGMI vs SHM), the distinct algorithm/protocol picks."""
    import os
    import socket

    from h2gcn_amd.partition import enable_rccl_debug_log, summarize_rccl_log

    for k in ("NCCL_DEBUG_SUBSYS", "NCCL_DEBUG_FILE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("NCCL_DEBUG", "VERSION")          # what the GPU image exports: too quiet, and it prints to stdout
    enable_rccl_debug_log(str(tmp_path))
    assert os.environ["NCCL_DEBUG"] == "INFO" and os.environ["NCCL_DEBUG_FILE"].startswith(str(tmp_path))
    # a measuring run logs communicator set-up only; the per-collective TUNING lines (file I/O on the launching thread inside a
    # timed step) are for the stand-alone first-contact run / on request
    assert os.environ["NCCL_DEBUG_SUBSYS"] == "INIT,GRAPH"
    monkeypatch.delenv("NCCL_DEBUG_SUBSYS")
    monkeypatch.setenv("NCCL_DEBUG", "WARN")             # a level the caller chose on purpose is kept
    enable_rccl_debug_log(str(tmp_path), tuning=True)
    assert os.environ["NCCL_DEBUG_SUBSYS"] == "INIT,GRAPH,TUNING" and os.environ["NCCL_DEBUG"] == "WARN"
    assert summarize_rccl_log(str(tmp_path)) == []                       # no file yet
    (tmp_path / f"rccl.{socket.gethostname()}.{os.getpid()}").write_text(RCCL_LOG)
    got = summarize_rccl_log(str(tmp_path))
    assert 0 < len(got) <= 10
    text = "\n".join(got)
    assert "version 2.22.3" in text and "nranks 8" in text and "ring channels: 16" in text
    assert "P2P/IPC x2" in text and "SHM/direct/direct x1" in text
    assert "AllGather of 39321600 B -> algorithm RING, protocol SIMPLE, channels 0..15" in text   # the pick for the largest message
    assert text.count("AllGather") == 1 and "AllReduce of 16 B -> algorithm TREE, protocol LL" in text
    for k in ("NCCL_DEBUG", "NCCL_DEBUG_SUBSYS", "NCCL_DEBUG_FILE"):
        monkeypatch.delenv(k, raising=False)


def test_rccl_log_summary_on_the_format_rccl_2_26_writes(tmp_path):
    """Lines as RCCL 2.26.6 of this image writes them (world size 1 on the GPU box): the topology search pattern is kept, the
    per-node warnings collapse into one entry per kind, nothing else crowds the ten slots."""
    import os
    import socket

    from h2gcn_amd.partition import summarize_rccl_log

    log = "\n".join(
        ["runc:171:171 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6"]
        + [f"[2026-09-28 23:54:08] runc:171:241 [0] /long/path/alt_rsmi.cc:675 NCCL WARN Could not read node # {i}" for i in range(3, 40)]
        + ['[2026-09-28 23:54:05] runc:171:171 [0] /long/path/init.cc:161 NCCL WARN Missing "iommu=pt" from kernel command line',
           "runc:171:241 [0] NCCL INFO === System : maxBw 5000.0 totalBw 5000.0 ===",
           "runc:171:241 [0] NCCL INFO Pattern 4, crossNic 0, nChannels 64, bw 48.000000/48.000000, type LOC/PIX, sameChannels 1",
           "runc:171:241 [0] NCCL INFO Pattern 1, crossNic 0, nChannels 64, bw 48.000000/48.000000, type LOC/PIX, sameChannels 1",
           "runc:171:241 [0] NCCL INFO 128 coll channels, 128 collnet channels, 0 nvls channels, 64 p2p channels, 128 p2p channels per peer",
           "runc:171:241 [0] NCCL INFO ncclCommInitRankConfig_impl comm 0x56 rank 0 nranks 1 cudaDev 0 nvmlDev 0 busId d9000 commId 0x3a - Init COMPLETE"])
    (tmp_path / f"rccl.{socket.gethostname()}.{os.getpid()}").write_text(log)
    got = summarize_rccl_log(str(tmp_path))
    assert len(got) <= 10 and got[0].startswith("RCCL version") and "Init COMPLETE" in got[1] and "coll channels" in got[2]
    assert sum("Pattern" in g for g in got) == 2
    assert sum("Could not read node" in g for g in got) == 1 and sum("iommu" in g for g in got) == 1


def test_peak_source_reads_the_box_and_says_whether_it_confirms_the_constant():
    """`roofline.peak_source` ("confirm on the box"): memory clock x bus width as tools/gather_probe reports them from
    hipDeviceProp_t -> the rate at 4 transfers per reported clock, compared with the 8 TB/s spec constant the fractions are quoted on."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench

    ok = bench.peak_source({"memory_clock_khz": 2000000, "memory_bus_width_bits": 8192, "device_name": "x", "gcn_arch": "gfx950"})
    assert ok["peak_confirmed_on_box"] is True and abs(ok["peak_box_derived_GBps"] - 8192.0) < 1e-6 and "confirms the constant" in ok["peak_source"]
    mi300 = bench.peak_source({"memory_clock_khz": 1300000, "memory_bus_width_bits": 8192})      # another part: 5.3 TB/s, not this constant
    assert mi300["peak_confirmed_on_box"] is False and abs(mi300["peak_box_derived_GBps"] - 5324.8) < 1e-6 and "does NOT match" in mi300["peak_source"]
    for missing in (None, {}, {"error": "tools/gather_probe not built"}, {"memory_clock_khz": 0, "memory_bus_width_bits": 8192}):
        none = bench.peak_source(missing)
        assert none["peak_confirmed_on_box"] is False and "reported no memory clock" in none["peak_source"]
    assert set(bench.N1_ADJOINT_CHECKSUMS) <= set(bench.N1_CHECKSUMS)       # every adjoint constant belongs to a shape with a forward one
