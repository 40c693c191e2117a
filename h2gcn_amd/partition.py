"""Vertex/row partitioning of the hop aggregation over the GPUs of one node (new -- the reference is
single-process, single-device: no collective appears anywhere in it).

Scheme: rank p owns a contiguous block of rows of every hop matrix ``A_k[rows_p, :]`` (global
column ids) and the matching rows of the embedding ``X[rows_p, :]``.  ``Y[i, k, :]`` depends only on row ``i`` of
``A_k`` and on the rows of ``X`` its column ids name, so one exchange per layer suffices: all-gather ``X`` (RCCL
``ncclAllGather`` over xGMI through ``torch.distributed``, backend "nccl"), then the local fused SpMM.  The
per-row summation tree is canonical (``include/h2gcn_hip.h``, "Floating point"): it depends on the row's own nonzeros
only, never on the partitioning, the slice width or the feature chunking, so P ranks reproduce the 1-rank result
bit-for-bit whatever schedule either side picked.
Everything after the aggregation in H2GCN (concat, dropout, classifier) is row-local; the backward pass needs the
mirror-image reduce-scatter of ``dX``.

Partitioning rule (:class:`RowPartition`): contiguous row blocks, either equal row counts (synthetic graphs: random row
order balances the nonzeros by itself) or NNZ-BALANCED -- a prefix-sum-of-work split, for real graphs whose degrees are
power-law and unshuffled (the planetoid / generator files the reference loads, ``_dataset.py:195-305``).  Blocks then
differ in height; every exchange works on a PADDED row space (``per`` = the tallest block; rank q's rows land at
``[q * per, q * per + rows_q)`` of the gathered buffer) so that the all-gather stays one fixed-count collective, and a
shard's column ids are remapped into that space once, when the shard is built (``RowPartition.to_padded``).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_rccl_process_group(device: torch.device, timeout_s: Optional[float] = None) -> None:
    """``torch.distributed`` over RCCL with a HIGH-PRIORITY communication stream: the exchange kernels must be
    scheduled promptly while a 150k-workgroup SpMM grid saturates every CU, otherwise the pipelining of
    :class:`PipelinedHopAggregation` degenerates into serial execution.  ``timeout_s`` bounds every collective so
    that a rank that failed leaves its peers with an error instead of a hang."""
    import datetime

    if os.environ.get("H2GCN_SHARE_GPU") == "1" and "NCCL_HOSTID" not in os.environ:
        # Test mode (several ranks on ONE GPU): RCCL refuses two ranks with the same (host hash, PCI bus id) -- "Duplicate GPU
        # detected".  A host id of its own per rank makes the ranks look like different hosts: RCCL then connects them through
        # its NET/Socket transport, and ProcessGroupNCCL, its streams, its watchdog and every collective run for real with
        # world > 1 on a one-GPU box.  (The transport is not xGMI; everything above the transport is what an 8-GPU node runs.)
        os.environ["NCCL_HOSTID"] = f"h2gcn-shared-gpu-rank-{os.environ.get('RANK', '0')}"
    kw = {} if timeout_s is None else {"timeout": datetime.timedelta(seconds=float(timeout_s))}
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        dist.init_process_group("nccl", device_id=device, pg_options=opts, **kw)
    except (AttributeError, TypeError):  # older torch: no options object
        dist.init_process_group("nccl", device_id=device, **kw)


def enable_rccl_debug_log(directory: str, tuning: Optional[bool] = None) -> None:
    """Have RCCL write what it decides at communicator set-up (topology graph, channels, transports) into one FILE per process
    under ``directory`` -- never to stdout/stderr.  Must run before the process group is created.  Ring-vs-direct is what decides
    the 8-GPU outcome of the per-layer all-gather, so a multi-rank run keeps this next to its number.

    ``tuning`` adds the TUNING subsystem: one "AllGather: N Bytes -> Algo ... proto ..." line (a write + flush on the launching
    thread) per collective ENQUEUE.  That is I/O inside a timed step, so a measuring run leaves it off (default: off unless
    ``H2GCN_RCCL_LOG_TUNING=1``); the stand-alone first-contact table (``bench.py --dry-exchange``), which times nothing that is
    reported as the metric, turns it on.  INIT + GRAPH write at communicator set-up only.

    ``NCCL_DEBUG``: this image exports VERSION, which makes RCCL print its banner to STDOUT -- raised to INFO (the file takes it);
    WARN, or a more verbose level, a subsystem list or a file name the caller chose are kept."""
    os.makedirs(directory, exist_ok=True)
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE", "ABORT", "WARN"):
        os.environ["NCCL_DEBUG"] = "INFO"
    if tuning is None:
        tuning = os.environ.get("H2GCN_RCCL_LOG_TUNING") == "1"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING" if tuning else "INIT,GRAPH")
    os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(directory, "rccl.%h.%p"))


def summarize_rccl_log(directory: Optional[str] = None, pid: Optional[int] = None, max_lines: int = 10):
    """At most ``max_lines`` strings saying what RCCL chose in this process: version, communicator size, channel count,
    transports per connection (P2P/IPC over xGMI vs SHM vs NET), rings/trees, and the distinct algorithm/protocol picks.
    Reads the file :func:`enable_rccl_debug_log` named (``$NCCL_DEBUG_FILE`` with %h/%p expanded); an empty list when
    there is no such file."""
    import glob
    import re
    import socket

    pattern = os.environ.get("NCCL_DEBUG_FILE", "") if directory is None else os.path.join(directory, "rccl.%h.%p")
    if not pattern:
        return []
    pid = os.getpid() if pid is None else pid
    path = pattern.replace("%h", socket.gethostname()).replace("%p", str(pid))
    files = [path] if os.path.exists(path) else sorted(glob.glob(pattern.replace("%h", "*").replace("%p", str(pid))))
    if not files:
        return []
    try:
        text = open(files[0], errors="replace").read()
    except OSError:
        return []
    body = [re.sub(r"^.*?NCCL (INFO|WARN) ", lambda m: "WARN " if m.group(1) == "WARN" else "", ln).strip() for ln in text.splitlines()]
    out, seen = [], set()

    def add(s, key=None):
        s = s[:240]
        key = key or s
        if s and key not in seen and len(out) < max_lines:
            seen.add(key)
            out.append(s)

    def first(pat):
        for ln in body:
            if re.search(pat, ln):
                add(ln)
                return

    first(r"(RCCL|NCCL) version")
    first(r"nranks \d+.*Init COMPLETE")
    first(r"\d+ coll channels")
    chans = {int(m.group(1)) for ln in body for m in [re.match(r"Channel \d+/(\d+)\s*:", ln)] if m}
    if chans:
        add(f"ring channels: {max(chans)}")
    # the search patterns of the topology graph: "Pattern 4, crossNic 0, nChannels 64, bw 48.0/48.0, type LOC/PIX, ..." -- ring vs
    # tree vs direct and the link type (XGMI / PIX / ...) they were found on
    n_pat = 0
    for ln in body:
        if re.match(r"Pattern \d+, crossNic", ln) and n_pat < 3 and ln[:240] not in seen:
            add(ln)
            n_pat += 1
    via = {}
    for ln in body:
        m = re.search(r"\bvia (\S+)", ln)
        if m:
            via[m.group(1)] = via.get(m.group(1), 0) + 1
    if via:
        add("connections by transport: " + ", ".join(f"{k} x{v}" for k, v in sorted(via.items())))
    # TUNING: "AllGather: 39321600 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..15}" (RCCL 2.26; older builds print
    # numbers).  One pick per collective -- the one for its LARGEST message, the regime the exchange lives in -- AllGather first
    picks = {}
    for ln in body:
        m = re.search(r"(\w+): (\d+) Bytes -> Algo (\S+) proto (\S+)(?: channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\})?", ln)
        if m and (m.group(1) not in picks or int(m.group(2)) > picks[m.group(1)][0]):
            picks[m.group(1)] = (int(m.group(2)), m.group(3), m.group(4), m.group(5), m.group(6))
    for name in sorted(picks, key=lambda k: (k != "AllGather", -picks[k][0]))[:3]:
        nbytes, algo, proto, lo, hi = picks[name]
        add(f"{name} of {nbytes} B -> algorithm {algo}, protocol {proto}" + (f", channels {lo}..{hi}" if lo is not None else ""))
    for ln in body:
        if re.search(r"^(Ring|Trees?) ", ln) or "Connected all" in ln:
            add(ln)
    for ln in body:   # warnings last, one per kind (the numbers in them vary: "Could not read node # 7")
        if ln.startswith("WARN"):
            add(ln, key=re.sub(r"\d+", "N", ln))
    if not out:   # a log that matches none of the patterns: its first lines are still better than nothing
        for ln in body[:max_lines]:
            add(ln)
    return out


def block_bounds(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[r0, r1) of ``rank``'s row block: equal blocks of ceil(n/P) rows."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world size {world_size}")
    per = -(-n_rows // world_size)
    r0 = min(rank * per, n_rows)
    return r0, min(r0 + per, n_rows)


def rows_per_rank(n_rows: int, world_size: int) -> int:
    return -(-n_rows // world_size)


class RowPartition:
    """Contiguous row blocks ``[bounds[p], bounds[p+1])`` of an ``n``-row operand over ``world`` ranks.

    ``per`` = rows of the tallest block = the fixed shard height of every exchange; the padded row space has
    ``world * per`` rows and global row ``i`` of rank ``q`` sits at ``q * per + (i - bounds[q])``.  For equal blocks
    (``RowPartition.equal``) the padded space coincides with the global one (``to_padded`` is the identity)."""

    def __init__(self, bounds):
        self.bounds = [int(b) for b in bounds]
        if len(self.bounds) < 2 or self.bounds[0] != 0 or any(b1 < b0 for b0, b1 in zip(self.bounds, self.bounds[1:])):
            raise ValueError(f"bad partition bounds {self.bounds}")
        self.world = len(self.bounds) - 1
        self.n = self.bounds[-1]
        self.per = max(1, max(b1 - b0 for b0, b1 in zip(self.bounds, self.bounds[1:]))) if self.n else 0
        self.is_equal = self.bounds == [min(p * rows_per_rank(self.n, self.world), self.n) for p in range(self.world + 1)]

    @classmethod
    def equal(cls, n_rows: int, world_size: int) -> "RowPartition":
        per = rows_per_rank(n_rows, world_size)
        return cls([min(p * per, n_rows) for p in range(world_size + 1)])

    @classmethod
    def balanced(cls, row_work, world_size: int) -> "RowPartition":
        """Prefix-sum-of-work split: block p ends at the first row where the running work reaches ``(p+1)/P`` of the
        total (``row_work``: non-negative per-row cost, e.g. nonzeros over all hops + a constant per row for the output
        write).  Each block's work exceeds the mean by less than one row's work."""
        import numpy as np

        w = np.asarray(row_work, dtype=np.float64).reshape(-1)
        n = w.shape[0]
        cum = np.cumsum(w)
        total = float(cum[-1]) if n else 0.0
        if total <= 0.0:
            return cls.equal(n, world_size)
        targets = total * np.arange(1, world_size) / world_size
        # a block takes the row that crosses its target only if that leaves it closer to the target
        cut = np.searchsorted(cum, targets, side="left")
        lo = np.where(cut > 0, cum[np.maximum(cut - 1, 0)], 0.0)
        take = (cum[np.minimum(cut, n - 1)] - targets) <= (targets - lo)
        cut = np.minimum(cut + take.astype(np.int64), n)
        bounds = np.concatenate([[0], np.maximum.accumulate(cut), [n]])
        return cls(bounds.tolist())

    def rows(self, rank: int) -> Tuple[int, int]:
        return self.bounds[rank], self.bounds[rank + 1]

    def to_padded(self, colidx: torch.Tensor) -> torch.Tensor:
        """Global row ids -> positions in the padded row space (int32 in, int32 out; identity for equal blocks)."""
        if self.is_equal:
            return colidx
        b = torch.tensor(self.bounds, dtype=torch.int64, device=colidx.device)
        c = colidx.to(torch.int64)
        q = torch.bucketize(c, b[1:], right=True)
        return (q * self.per + (c - b[q])).to(torch.int32)

    def imbalance(self, row_work) -> float:
        """max over ranks of the block's work / mean block work."""
        import numpy as np

        w = np.asarray(row_work, dtype=np.float64).reshape(-1)
        cum = np.concatenate([[0.0], np.cumsum(w)])
        per_rank = np.diff(cum[self.bounds])
        return float(per_rank.max() / max(per_rank.mean(), 1e-300))


class EmbeddingAllGather:
    """Reusable buffers + the per-layer all-gather of the row-sharded embedding.

    ``gather(x_local)`` returns the full ``[n_rows, d]`` matrix (a view of an internal padded buffer that is
    overwritten by the next call).  With ``world_size == 1`` it returns ``x_local`` itself -- no copy, no
    collective."""

    def __init__(self, n_rows: int, d: int, device, group: Optional[dist.ProcessGroup] = None,
                 dtype=torch.float32):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n_rows, self.d = int(n_rows), int(d)
        self.per = rows_per_rank(self.n_rows, self.world)
        self.r0, self.r1 = block_bounds(self.n_rows, self.world, self.rank)
        if self.world > 1:
            self.full = torch.empty((self.world * self.per, self.d), dtype=dtype, device=device)
            self.send = torch.zeros((self.per, self.d), dtype=dtype, device=device)
        else:
            self.full = self.send = None

    def gather(self, x_local: torch.Tensor) -> torch.Tensor:
        if tuple(x_local.shape) != (self.r1 - self.r0, self.d):
            raise ValueError(f"local embedding has shape {tuple(x_local.shape)}, expected {(self.r1 - self.r0, self.d)}")
        if self.world == 1:
            return x_local
        if self.r1 - self.r0 == self.per and x_local.is_contiguous():
            send = x_local
        else:  # last (short) block: pad to the common count
            self.send[: self.r1 - self.r0].copy_(x_local)
            send = self.send
        _all_gather_rows(self.full, send, self.group)
        return self.full[: self.n_rows]


def shard_rows_scipy(mats, world_size: int, rank: int):
    """Row block of each scipy hop matrix for ``rank`` (global column space kept)."""
    out = []
    for m in mats:
        r0, r1 = block_bounds(m.shape[0], world_size, rank)
        out.append(m[r0:r1])
    return out


class IpcExchange:
    """All-gather of row shards through IPC-exported device buffers (``h2gcn_xchg_*`` of ``libh2gcn_hip.so``): each
    rank stages its shard into an exported slot, flags its peers over xGMI, and pulls the peers' shards with
    copy-engine transfers on one stream per peer (``mode="engine"``: no CU is taken from the SpMM, every
    point-to-point link carries its own transfer) or one small copy kernel (``mode="kernel"``).  An alternative to
    ``ncclAllGather`` that needs only a bootstrap channel (``all_gather_object`` of 192-byte handles, any backend)
    -- and, unlike RCCL, also runs with several ranks on ONE GPU, which is how the GPU suite tests it.

    ``begin(channel, x_shard, full)`` returns at once (everything is enqueued device-side); ``end(channel)`` makes
    the current stream wait for the shards.  ``check()`` raises if any device-side wait ever timed out."""

    def __init__(self, n_channels: int, slot_bytes: int, device, group: Optional[dist.ProcessGroup] = None,
                 mode: str = "engine", timeout_ms: Optional[int] = None):
        import ctypes as C

        from . import _capi
        if mode not in ("engine", "kernel"):
            raise ValueError(f"unknown IPC exchange mode {mode!r}")
        if timeout_ms is None:   # how long a GPU waits for a peer's shard before it gives up, poisons the block and flags it
            timeout_ms = int(os.environ.get("H2GCN_XCHG_TIMEOUT_MS", "10000"))
        self._C, self._capi = C, _capi
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.device = torch.device(device)
        self.mode = mode
        self._handle = C.c_void_p()
        L = _capi.lib()
        # Collectively safe: a rank whose local set-up fails still takes part in the handle exchange (sending None),
        # so its peers raise instead of waiting for it.
        blob, err = None, None
        with torch.cuda.device(self.device):
            try:
                if os.environ.get("H2GCN_XCHG_FAIL_ON_RANK") == str(self.rank):  # failure injection (tests): one rank's
                    raise RuntimeError("injected set-up failure")                 # set-up fails, nobody may hang
                _capi.check(L.h2gcn_xchg_create(self.world, self.rank, int(n_channels), int(slot_bytes),
                                                _capi.XCHG_COPY_ENGINE if mode == "engine" else _capi.XCHG_COPY_KERNEL,
                                                int(timeout_ms), C.byref(self._handle)))
                buf = C.create_string_buffer(_capi.XCHG_BLOB_BYTES)
                _capi.check(L.h2gcn_xchg_export(self._handle, buf))
                blob = bytes(buf.raw)
            except Exception as e:  # noqa: BLE001 -- reported below, after the collective
                err = e
            if self.world > 1:
                blobs = [None] * self.world
                dist.all_gather_object(blobs, blob, group=group)
                if err is None and all(b is not None for b in blobs):
                    try:
                        _capi.check(L.h2gcn_xchg_connect(self._handle, b"".join(blobs)))
                    except Exception as e:  # noqa: BLE001
                        err = e
                oks = [None] * self.world
                dist.all_gather_object(oks, err is None and all(b is not None for b in blobs), group=group)
                if not all(oks):
                    self._destroy_local()
                    raise RuntimeError(f"IPC exchange unavailable (ranks ok: {oks})" + (f": {err}" if err else ""))
            elif err is not None:
                raise err

    def _destroy_local(self) -> None:
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._capi.lib().h2gcn_xchg_destroy(h)
            self._handle = self._C.c_void_p()

    def begin(self, channel: int, x_shard: torch.Tensor, full: torch.Tensor, rows_per_rank: int, pull: bool = True) -> None:
        """Start gathering ``x_shard`` ([rows <= rows_per_rank, w], last dim contiguous) into ``full``
        ([world * rows_per_rank, w] contiguous).  ``pull=False``: only stage and announce (``h2gcn_xchg_allgather_post``);
        call :meth:`pull` afterwards -- pipelines post every channel before pulling any."""
        if x_shard.dim() != 2 or x_shard.dtype != torch.float32 or (x_shard.shape[1] > 1 and x_shard.stride(1) != 1):
            raise ValueError("shard must be a float32 [rows, w] view with a contiguous last dimension")
        w = int(x_shard.shape[1])
        if tuple(full.shape) != (self.world * rows_per_rank, w) or not full.is_contiguous() or full.dtype != torch.float32:
            raise ValueError(f"full must be a contiguous float32 [{self.world * rows_per_rank}, {w}] buffer")
        C = self._C
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            fn = self._capi.lib().h2gcn_xchg_allgather_begin if pull else self._capi.lib().h2gcn_xchg_allgather_post
            self._capi.check(fn(
                self._handle, int(channel), C.c_void_p(x_shard.data_ptr()), x_shard.stride(0) if x_shard.shape[0] > 0 else w,
                int(x_shard.shape[0]), int(rows_per_rank), w, C.c_void_p(full.data_ptr()), C.c_void_p(stream)))

    def pull(self, channel: int, full: torch.Tensor, rows_per_rank: int, halo=None) -> None:
        """Issue the pulls of a posted channel.  ``halo``: per peer an ascending int32 tensor of the LOCAL rows of that peer's
        shard to fetch (``None`` / missing = nothing from that peer is listed); only those rows land in ``full``
        (``h2gcn_xchg_allgather_pull_rows``; copy-kernel mode, width a multiple of 4).  Default: whole shards."""
        C = self._C
        with torch.cuda.device(self.device):
            if halo is None:
                self._capi.check(self._capi.lib().h2gcn_xchg_allgather_pull(self._handle, int(channel), int(rows_per_rank),
                                                                            int(full.shape[1]), C.c_void_p(full.data_ptr())))
                return
            ptrs = (C.c_void_p * self.world)(*[(t.data_ptr() if (t is not None and t.numel()) else None) for t in halo])
            counts = (C.c_int64 * self.world)(*[(int(t.numel()) if t is not None else 0) for t in halo])
            self._capi.check(self._capi.lib().h2gcn_xchg_allgather_pull_rows(self._handle, int(channel), int(rows_per_rank),
                                                                             int(full.shape[1]), C.c_void_p(full.data_ptr()), ptrs, counts))

    def end(self, channel: int) -> None:
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._capi.check(self._capi.lib().h2gcn_xchg_allgather_end(self._handle, int(channel), self._C.c_void_p(stream)))

    def reduce_scatter_begin(self, channel: int, src_full: torch.Tensor, rows_per_rank: int) -> torch.Tensor:
        """Start the sum over ranks of ``src_full`` ([world * rows_per_rank, w] contiguous): stage it, tell the peers, start
        pulling this rank's row block of every peer (on the library's own streams).  Returns the ``[rows_per_rank, w]``
        tensor that :meth:`reduce_scatter_end` (same channel) fills -- blocks are added in ascending rank order.  Other work
        may be issued on the current stream in between: that is the overlap.  The slot must cover the whole matrix."""
        w = int(src_full.shape[1])
        if tuple(src_full.shape) != (self.world * rows_per_rank, w) or not src_full.is_contiguous() or src_full.dtype != torch.float32:
            raise ValueError(f"src must be a contiguous float32 [{self.world * rows_per_rank}, w] matrix")
        out = torch.empty((rows_per_rank, w), dtype=torch.float32, device=self.device)
        C, L = self._C, self._capi.lib()
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._capi.check(L.h2gcn_xchg_reduce_scatter_begin(self._handle, int(channel), C.c_void_p(src_full.data_ptr()),
                                                               int(rows_per_rank), w, C.c_void_p(out.data_ptr()), stream))
        return out

    def reduce_scatter_end(self, channel: int) -> None:
        C = self._C
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._capi.check(self._capi.lib().h2gcn_xchg_reduce_scatter_end(self._handle, int(channel), stream))

    def reduce_scatter(self, channel: int, src_full: torch.Tensor, rows_per_rank: int) -> torch.Tensor:
        """``reduce_scatter_begin`` + ``reduce_scatter_end``."""
        out = self.reduce_scatter_begin(channel, src_full, rows_per_rank)
        self.reduce_scatter_end(channel)
        return out

    def check(self) -> None:
        self._capi.check(self._capi.lib().h2gcn_xchg_status(self._handle))

    def reset_dependencies(self) -> None:
        """Forget the events of earlier steps (call with the device idle, right before a hipGraph capture begins: a
        capturing stream must not wait on events recorded outside its capture)."""
        self._capi.check(self._capi.lib().h2gcn_xchg_reset_dependencies(self._handle))

    def all_reduce_sum(self, channel: int, flat: torch.Tensor, scratch: torch.Tensor) -> torch.Tensor:
        """Sum of a small fp32 vector over the ranks WITHOUT a collective library: all-gather the vectors through this
        exchange (``scratch``: contiguous ``[world, n]``), then add the rows -- every rank adds the same rows in the same
        order, so the replicas stay bit-identical.  Capturable in a hipGraph (copy-kernel mode), unlike a gloo / host
        staged all-reduce; used for the dense-kernel gradients and the loss scalars of a row-partitioned step."""
        n = flat.numel()
        self.begin(channel, flat.view(1, n), scratch, 1)
        self.end(channel)
        return scratch.sum(dim=0)

    def close(self) -> None:
        """Collective: synchronises, makes sure no peer is still pulling from this rank, then frees the buffers."""
        h = getattr(self, "_handle", None)
        if h is None or not h.value:
            return
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier(group=self.group)
        self._destroy_local()


class PipelinedHopAggregation:
    """One layer's exchange + aggregation with the all-gather hidden behind the SpMM.

    The embedding is exchanged in ``n_chunks`` feature-column chunks on a side stream; the fused 1+2-hop SpMM of
    chunk ``c`` (main stream) runs while chunk ``c+1`` is still in flight over xGMI.  Output columns are
    independent sums and the kernel's per-row summation tree is the same for every chunk width, so a P-rank run equals
    the 1-rank run bit-for-bit WHATEVER chunking either of them uses.  Costs: the column ids /
    values are re-read once per chunk (8 B per edge against ``4*d/C`` B of gathered features) and the local shard is
    staged once into chunk-major send buffers.

    xGMI arithmetic: at P ranks each GPU receives ``(P-1)/P * N * d * 4`` bytes per layer over
    ``P-1`` point-to-point links; on the products shape that is of the same order as the per-rank SpMM time,
    so an un-overlapped all-gather would cap the 8-GPU speed-up near 4x.
    """

    FAST_WIDTHS = (32, 64, 128, 256)

    def __init__(self, plan, n_rows_global: int, d: int, n_chunks, device,
                 group: Optional[dist.ProcessGroup] = None, exchange: str = "allgather",
                 partition: Optional[RowPartition] = None, ipc_timeout_ms: Optional[int] = None, halo="auto"):
        """``halo``: fetch from every peer only the rows of the embedding this rank's hop matrices name (``ipc_kernel`` exchange,
        chunk widths multiples of 4): ``True`` / ``False`` / ``"auto"`` = when those are less than 90 % of the remote rows (real
        graphs with locality; never the synthetic shapes or a products-like 2-hop ring, where every row is named).  The named rows
        are exactly the column ids of the plan, so the result cannot differ from the dense exchange.
        ``n_chunks``: number of equal feature chunks, or an explicit list of chunk widths summing to ``d``
        (e.g. ``[32, 32, 64]``: a narrow first chunk shortens the un-overlapped head of the exchange).
        ``partition``: the row blocks (default: equal blocks).  With unequal blocks the plan's column ids must already
        live in the padded row space (``RowPartition.to_padded``; ``plan.n_cols == world * per``)."""
        if exchange not in ("allgather", "p2p", "ipc_engine", "ipc_kernel"):
            raise ValueError(f"unknown exchange {exchange!r}")
        self.ipc_timeout_ms = ipc_timeout_ms   # None: $H2GCN_XCHG_TIMEOUT_MS, else 10 s
        self._gather = _all_gather_rows_p2p if exchange == "p2p" else _all_gather_rows
        self.exchange = exchange
        self.ipc = None
        self.halo, self.halo_ratio = None, 1.0
        if isinstance(n_chunks, (list, tuple)):
            widths = [int(w) for w in n_chunks]
            if sum(widths) != d or min(widths) < 1:
                raise ValueError(f"chunk widths {widths} do not sum to d = {d}")
        else:
            if d % int(n_chunks) != 0:
                raise ValueError(f"d = {d} is not divisible into {n_chunks} chunks")
            widths = [d // int(n_chunks)] * int(n_chunks)
        min_cols = int(getattr(plan, "min_chunk_cols", 1))
        if len(widths) > 1 and min(widths) < min_cols:
            raise ValueError(f"feature chunks {widths}: chunks narrower than {min_cols} columns are not worth their extra pass over the indices")
        self.widths = widths
        self.offsets = [sum(widths[:c]) for c in range(len(widths))]
        self.plan = plan
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n, self.d, self.C = int(n_rows_global), int(d), len(widths)
        self.partition = partition if partition is not None else RowPartition.equal(self.n, self.world)
        if self.partition.world != self.world or self.partition.n != self.n:
            raise ValueError(f"partition is {self.partition.world} blocks of {self.partition.n} rows, run has {self.world} ranks / {self.n} rows")
        self.per = self.partition.per
        self.r0, self.r1 = self.partition.rows(self.rank)
        self.device = device
        #: rows of the gather source the plan addresses: the global rows (equal blocks) or the padded row space
        self.n_src = self.n if (self.partition.is_equal and plan.n_cols == self.n) else self.world * self.per
        if plan.n_cols != self.n_src or plan.n_rows != self.r1 - self.r0:
            raise ValueError(f"plan is {plan.n_rows} x {plan.n_cols}, expected {self.r1 - self.r0} x {self.n_src}")
        self.use_streams = torch.device(device).type == "cuda"  # CPU/gloo (tests): same schedule, no streams
        if self.world > 1:
            # high priority: its event waits / staging must not queue behind the SpMM grid
            self.comm_stream = torch.cuda.Stream(device=device, priority=-1) if self.use_streams else None
            self.send = [torch.zeros((self.per, w), dtype=torch.float32, device=device) for w in widths]
            self.full = [torch.empty((self.world * self.per, w), dtype=torch.float32, device=device) for w in widths]
            if self.use_streams:
                self.staged = [torch.cuda.Event() for _ in range(self.C)]
                self.ready = [torch.cuda.Event() for _ in range(self.C)]
            if exchange.startswith("ipc_"):
                if not self.use_streams:
                    raise ValueError("the IPC exchange needs GPU buffers")
                self.ipc = IpcExchange(self.C, self.per * max(widths) * 4, device, group, mode=exchange[4:],
                                       timeout_ms=self.ipc_timeout_ms)
            #: per peer the rows to pull (None: dense pulls); halo_ratio = named remote rows / all remote rows
            env = os.environ.get("H2GCN_HALO")
            if env is not None:
                halo = {"0": False, "1": True}.get(env, halo)
            from . import _capi
            if (halo and exchange == "ipc_kernel" and all(w % 4 == 0 for w in widths) and hasattr(plan, "colidx")
                    and _capi.has("h2gcn_xchg_allgather_pull_rows")):
                lists, named, remote = self._halo_lists(plan)
                self.halo_ratio = named / max(remote, 1)
                if halo is True or self.halo_ratio < 0.9:
                    self.halo = lists
        #: set to a list to have (start, end) timing-event pairs appended around every SpMM launch
        self.kernel_events = None

    def _halo_lists(self, plan):
        """Per peer q the ascending LOCAL row ids of q's block that some column id of this rank's hop matrices names (column ids
        live in the row space the exchange lands in: global for equal blocks, padded otherwise)."""
        key = (self.world, self.rank, self.per, tuple(self.partition.bounds))
        cache = getattr(plan, "_halo_cache", None)
        if cache is not None and cache[0] == key:        # (bench.py builds one pipeline per candidate schedule on the same plan)
            return cache[1]
        cols = torch.unique(torch.cat([c.to(torch.int64) for c in plan.colidx])) if sum(c.numel() for c in plan.colidx) else torch.zeros(0, dtype=torch.int64, device=self.device)
        lists, named, remote = [], 0, 0
        for q in range(self.world):
            q0, q1 = self.partition.rows(q)
            base = q * self.per                       # where q's rows start in the landing space (== q0 for equal blocks)
            if q == self.rank:
                lists.append(None)
                continue
            sel = cols[(cols >= base) & (cols < base + (q1 - q0))] - base
            lists.append(sel.to(torch.int32).contiguous())
            named += int(sel.numel())
            remote += q1 - q0
        try:
            plan._halo_cache = (key, (lists, named, remote))
        except AttributeError:
            pass
        return lists, named, remote

    def close(self) -> None:
        """Collective (when the IPC exchange is in use): release the exported buffers."""
        if self.ipc is not None:
            self.ipc.close()
            self.ipc = None
        if getattr(self, "ipc_rs", None) is not None:
            self.ipc_rs.close()
            self.ipc_rs = None

    def check(self) -> None:
        """Raise if a device-side wait of this layer's IPC exchanges has ever given up (a peer stalled beyond the time
        limit or died; the affected blocks were overwritten with NaN).  Reads a host-mapped word: no synchronisation."""
        if self.ipc is not None:
            self.ipc.check()
        if getattr(self, "ipc_rs", None) is not None:
            self.ipc_rs.check()

    def exchange_only(self) -> None:
        """The step's exchange without the SpMM (diagnostics): every chunk of the last staged shard again."""
        if self.world == 1:
            return
        if self.ipc is not None:
            for c in range(self.C):
                self.ipc.begin(c, self.send[c], self.full[c], self.per, pull=False)
            for c in range(self.C):
                self.ipc.pull(c, self.full[c], self.per, halo=self.halo)
            for c in range(self.C):
                self.ipc.end(c)
        else:
            for c in range(self.C):
                self._gather(self.full[c], self.send[c], self.group)

    # ---- chunk-level schedule (world > 1): what __call__ does, one chunk at a time, so that a caller can start the exchange
    #      of a chunk the moment its columns exist (cross-round pipelining in _ShardedFusedPropagation) ----
    def start_chunk(self, c: int, x_chunk: torch.Tensor) -> None:
        """Stage this rank's rows of chunk ``c`` (``[n_local, widths[c]]``, any row stride) and start its exchange."""
        n_local = self.r1 - self.r0
        if self.ipc is not None:
            self.ipc.begin(c, x_chunk, self.full[c], self.per, pull=False)
            self.ipc.pull(c, self.full[c], self.per, halo=self.halo)
        elif not self.use_streams:
            self.send[c][:n_local].copy_(x_chunk)
            self._gather(self.full[c], self.send[c], self.group)
        else:
            main = torch.cuda.current_stream(self.device)
            self.send[c][:n_local].copy_(x_chunk)
            self.staged[c].record(main)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self.staged[c])
                self._gather(self.full[c], self.send[c], self.group)
                self.ready[c].record(self.comm_stream)

    def wait_chunk(self, c: int) -> torch.Tensor:
        """Make the current stream wait for chunk ``c``'s exchange; returns the gathered ``[n_src, widths[c]]`` source."""
        if self.ipc is not None:
            self.ipc.end(c)
        elif self.use_streams:
            torch.cuda.current_stream(self.device).wait_event(self.ready[c])
        return self.full[c][: self.n_src]

    def _spmm(self, x, out, hops=None):
        if self.kernel_events is None or not self.use_streams:
            self.plan.spmm(x, hops=hops, out=out)
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        self.plan.spmm(x, hops=hops, out=out)
        e.record()
        self.kernel_events.append((s, e))

    def __call__(self, x_local: torch.Tensor, out: Optional[torch.Tensor] = None, hops=None) -> torch.Tensor:
        """``x_local`` [n_local, d] (any row stride) -> ``out`` [n_local, H_sel, d] (any strides, last dim contiguous);
        ``hops`` = the hop filter of ``GCNLayer(hops=...)`` (reference ``_layers.py:57-59,80-81``)."""
        n_local = self.r1 - self.r0
        if tuple(x_local.shape) != (n_local, self.d):
            raise ValueError(f"local embedding has shape {tuple(x_local.shape)}, expected {(n_local, self.d)}")
        H = self.plan.n_selected(hops)
        if out is None:
            out = torch.empty((n_local, H, self.d), dtype=torch.float32, device=self.device)
        cols = [slice(o, o + w) for o, w in zip(self.offsets, self.widths)]
        if self.world == 1:
            for c in range(self.C):  # chunked on one GPU: same schedule without the exchange
                self._spmm(x_local[:, cols[c]], out[:, :, cols[c]], hops)
            return out
        if not self.use_streams:
            for c in range(self.C):
                self.send[c][:n_local].copy_(x_local[:, cols[c]])
                self._gather(self.full[c], self.send[c], self.group)
                self._spmm(self.full[c][: self.n_src], out[:, :, cols[c]], hops)
            return out
        if self.ipc is not None:
            # staging + notification of every chunk first (device-side order matters, see exchange.hip), pulls run on
            # the library's own streams; the SpMM of chunk c only waits for chunk c's shards
            for c in range(self.C):
                self.ipc.begin(c, x_local[:, cols[c]], self.full[c], self.per, pull=False)
            for c in range(self.C):
                self.ipc.pull(c, self.full[c], self.per, halo=self.halo)
            for c in range(self.C):
                self.ipc.end(c)
                self._spmm(self.full[c][: self.n_src], out[:, :, cols[c]], hops)
            return out
        main = torch.cuda.current_stream(self.device)
        for c in range(self.C):  # stage chunk by chunk so that the first exchange can start after the first copy
            self.send[c][:n_local].copy_(x_local[:, cols[c]])
            self.staged[c].record(main)
        with torch.cuda.stream(self.comm_stream):
            for c in range(self.C):
                self.comm_stream.wait_event(self.staged[c])
                self._gather(self.full[c], self.send[c], self.group)
                self.ready[c].record(self.comm_stream)
        for c in range(self.C):
            main.wait_event(self.ready[c])
            self._spmm(self.full[c][: self.n_src], out[:, :, cols[c]], hops)
        return out


def _all_gather_rows(full: torch.Tensor, send: torch.Tensor, group=None) -> None:
    """``full`` ([P*per, d]) <- concatenation of every rank's ``send`` ([per, d]).  RCCL: one ``ncclAllGather``.
    gloo has no flat all-gather for device tensors, so the list form is used there (CPU tests, and the
    shared-GPU debugging mode of the GPU suite)."""
    if dist.get_backend(group) == "gloo" and full.is_cuda:
        dist.all_gather(list(full.chunk(dist.get_world_size(group), dim=0)), send, group=group)
    else:
        dist.all_gather_into_tensor(full, send, group=group)


def _all_gather_rows_p2p(full: torch.Tensor, send: torch.Tensor, group=None) -> None:
    """Same result as :func:`_all_gather_rows`, expressed as P-1 concurrent point-to-point transfers per rank
    (``ncclSend``/``ncclRecv`` grouped): on a fully connected xGMI node every pair has its own link, so the direct
    form keeps all 7 links busy instead of forwarding shards around a ring.  Which form is faster is a property of
    the node/RCCL build; ``bench.py`` times both during warm-up."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = send.shape[0]
    full[rank * per:(rank + 1) * per].copy_(send)
    ops = []
    for shift in range(1, world):
        dst = (rank + shift) % world
        src = (rank - shift) % world
        ops.append(dist.P2POp(dist.isend, send, dst, group))
        ops.append(dist.P2POp(dist.irecv, full[src * per:(src + 1) * per], src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def _reduce_scatter_rows(full: torch.Tensor, per: int, rank: int, group=None) -> torch.Tensor:
    """Sum ``full`` ([P*per, d]) over ranks and return this rank's block of ``per`` rows.  RCCL: one
    ``ncclReduceScatter``; gloo (CPU tests) has no reduce-scatter, so all-reduce + slice."""
    if dist.get_backend(group) == "gloo":
        dist.all_reduce(full, group=group)
        return full[rank * per:(rank + 1) * per].clone()
    out = torch.empty((per, full.shape[1]), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, full, group=group)
    return out


def _adjoint_reduce_scatter(layer: "PipelinedHopAggregation", grad_y: torch.Tensor, hops=None) -> torch.Tensor:
    """Backward of one row-sharded hop aggregation: ``grad_y`` [n_local, H_sel, d] (any strides, last dim contiguous) -> this
    rank's rows of ``sum_ranks A_k[rows_p, :]^T grad_y_p`` ([n_local, d]).

    The shard adjoint yields a full-height contribution that has to be summed over the ranks and scattered to the row owners
    -- the mirror image of the forward all-gather, the same bytes.  Like the forward it is pipelined in the
    layer's feature-column chunks: the adjoint launch of chunk ``c+1`` (main stream) runs while the reduce-scatter of chunk
    ``c`` is in flight (RCCL on the communication stream, or the library's IPC pulls on its own streams).  Output columns are
    independent sums, so the chunking never shows in the adjoint's bits."""
    plan = layer.plan
    if layer.world == 1:
        return plan.spmm_t(grad_y, hops=hops)
    n_local, per, rows = layer.r1 - layer.r0, layer.per, layer.world * layer.per
    device = layer.device
    cols = [slice(o, o + w) for o, w in zip(layer.offsets, layer.widths)]
    out = torch.empty((n_local, layer.d), dtype=torch.float32, device=device)

    def adjoint_chunk(c):   # [world * per, w_c]: the padded row space of the exchange, rows beyond the plan's columns zero
        g = grad_y[:, :, cols[c]]
        dx = plan.spmm_t(g, hops=hops)
        if dx.shape[0] == rows and dx.is_contiguous():
            return dx
        full = torch.zeros((rows, layer.widths[c]), dtype=torch.float32, device=device)
        full[: dx.shape[0]] = dx
        return full

    if layer.ipc is not None:   # the library's own exchange: pulls + a fixed-order sum, no collective library
        if getattr(layer, "ipc_rs", None) is None:
            layer.ipc_rs = IpcExchange(layer.C, rows * max(layer.widths) * 4, device, layer.group, mode=layer.exchange[4:],
                                       timeout_ms=layer.ipc_timeout_ms)
        pending = []
        for c in range(layer.C):
            full = adjoint_chunk(c)
            pending.append((full, layer.ipc_rs.reduce_scatter_begin(c, full, per)))
        for c, (full, mine) in enumerate(pending):
            layer.ipc_rs.reduce_scatter_end(c)
            out[:, cols[c]] = mine[:n_local]
        return out
    if not layer.use_streams:   # CPU / gloo (tests): same schedule, no streams
        for c in range(layer.C):
            out[:, cols[c]] = _reduce_scatter_rows(adjoint_chunk(c), per, layer.rank, layer.group)[:n_local]
        return out
    main = torch.cuda.current_stream(device)
    pending = []
    for c in range(layer.C):
        full = adjoint_chunk(c)
        produced = torch.cuda.Event()
        produced.record(main)
        with torch.cuda.stream(layer.comm_stream):
            layer.comm_stream.wait_event(produced)
            mine = _reduce_scatter_rows(full, per, layer.rank, layer.group)
            reduced = torch.cuda.Event()
            reduced.record(layer.comm_stream)
        full.record_stream(layer.comm_stream)   # allocated on the main stream, consumed on the communication stream
        mine.record_stream(main)                # and the other way round
        pending.append((mine, reduced))
    for c, (mine, reduced) in enumerate(pending):
        main.wait_event(reduced)
        out[:, cols[c]] = mine[:n_local]
    return out


class _ShardedHopSpMM(torch.autograd.Function):
    """Row-sharded GCNLayer with autograd: forward = all-gather(X) + local fused SpMM (pipelined); backward = the
    adjoint on the local shard, which yields a full-height contribution ``A_k[rows_p, :]^T dY_p``, summed across
    ranks and scattered back to the row owners (reduce-scatter: the mirror image of the forward all-gather,
    SURVEY.md §8e)."""

    @staticmethod
    def forward(ctx, x_local, layer, hops):
        ctx.layer, ctx.hops = layer, hops
        return layer(x_local, hops=hops)

    @staticmethod
    def backward(ctx, grad_y):
        return _adjoint_reduce_scatter(ctx.layer, grad_y, ctx.hops), None, None


def sharded_hop_spmm(layer: "PipelinedHopAggregation", x_local: torch.Tensor, hops=None) -> torch.Tensor:
    """Differentiable row-sharded hop aggregation: ``[n_local, d] -> [n_local, H_sel, d]``."""
    if x_local.requires_grad and torch.is_grad_enabled():
        return _ShardedHopSpMM.apply(x_local, layer, hops)
    return layer(x_local, hops=hops)


def _cross_round_ok(layers, widths, H) -> bool:
    """Cross-round pipelining applies when the run is distributed, every round is exchanged in chunks of ONE common width
    and that width divides the rounds' input widths -- then the launch (chunk c, hop h) of round k writes exactly chunk
    ``h * C_k + c`` of round k+1's input.  (``H2GCN_CROSS_ROUND=0`` switches it off.)"""
    if len(layers) < 2 or layers[0].world == 1 or os.environ.get("H2GCN_CROSS_ROUND", "1") == "0":
        return False
    cw = layers[0].widths[0]
    for k, layer in enumerate(layers):
        if any(w != cw for w in layer.widths) or widths[k] % cw != 0 or layer.plan.n_selected(None) != H:
            return False
    return True


class _ShardedFusedPropagation(torch.autograd.Function):
    """Row-sharded form of :class:`h2gcn_amd.layers._FusedPropagation`: the K aggregation rounds of H2GCN-K write
    straight into this rank's ``[n_local, W]`` concat buffer ``[r_K | r_0 | ... | r_{K-1}]``.  Round k all-gathers the
    slot holding ``r_{k-1}`` (read in place, row stride W) and lands ``r_k`` in its slot through the kernel's output
    strides -- none of the stack / flatten / concat copies the layer-by-layer interpreter would make.  Backward walks
    the rounds in reverse: shard adjoint, reduce-scatter, add the slot's incoming gradient."""

    @staticmethod
    def forward(ctx, r0, hops_obj, rounds, out=None, reuse=False):
        """``out`` / ``reuse``: as in :func:`h2gcn_amd.layers.fused_propagation` -- a caller-owned buffer to fill, or (``reuse``)
        one that already holds the propagation of this ``r0``: no exchange and no SpMM then, on any rank."""
        plan = hops_obj.plan
        n_local, w0 = r0.shape
        H = plan.n_hops
        widths = [w0 * H ** k for k in range(rounds + 1)]
        off = [0] * (rounds + 1)
        pos = widths[rounds]
        for k in range(rounds):
            off[k] = pos
            pos += widths[k]
        from .layers import concat_buffer
        total = sum(widths)
        if out is None:
            buf = concat_buffer(n_local, total, r0.device)
        else:
            if out.shape != (n_local, total) or out.dtype != torch.float32 or out.device != r0.device or not out.is_contiguous():
                raise ValueError(f"fused_propagation: out must be a contiguous float32 [{n_local}, {total}] tensor on {r0.device}")
            buf = out.view(n_local, total)
        if not reuse:
            buf[:, off[0]:off[0] + w0].copy_(r0)
            layers = [hops_obj.pipeline(widths[k - 1]) for k in range(1, rounds + 1)]
            if _cross_round_ok(layers, widths, H):
                _ShardedFusedPropagation._cross_round(buf, layers, widths, off, H)
            else:
                for k in range(1, rounds + 1):
                    src = buf[:, off[k - 1]:off[k - 1] + widths[k - 1]]
                    dst = buf[:, off[k]:off[k] + widths[k]].unflatten(1, (H, widths[k - 1]))
                    layers[k - 1](src, out=dst)
        ctx.hops_obj, ctx.rounds, ctx.widths, ctx.off = hops_obj, rounds, widths, off
        return buf

    @staticmethod
    def _cross_round(buf, layers, widths, off, H):
        """The rounds with the exchange of round k+1 started as soon as its columns exist: round k is launched per (feature
        chunk, hop) -- the hop matrices gather independently, so splitting the fused launch by hop costs only launch overhead
        -- and the moment the launch (c, h) is enqueued, the chunk of ``r_k`` it writes (= chunk ``h * C_k + c`` of round
        k+1's input) is staged and sent.  Round k+1's first exchange then runs under round k's remaining launches instead of
        in front of round k+1 (the un-overlapped head of every round but the first).  Same launches per (row, hop, column),
        hence the same bits as the round-by-round schedule."""
        rounds = len(layers)
        first = layers[0]
        src0 = buf[:, off[0]:off[0] + widths[0]]
        for c in range(first.C):
            first.start_chunk(c, src0[:, first.offsets[c]:first.offsets[c] + first.widths[c]])
        for k in range(1, rounds + 1):
            cur = layers[k - 1]
            nxt = layers[k] if k < rounds else None
            dst = buf[:, off[k]:off[k] + widths[k]].unflatten(1, (H, widths[k - 1]))
            for c in range(cur.C):
                src = cur.wait_chunk(c)
                cols = slice(cur.offsets[c], cur.offsets[c] + cur.widths[c])
                if nxt is None:
                    cur._spmm(src, dst[:, :, cols], None)          # last round: nothing waits for it, keep the fused launch
                    continue
                for h in range(H):
                    cur._spmm(src, dst[:, h:h + 1, cols], [h])
                    nxt.start_chunk(h * cur.C + c, dst[:, h, cols])

    @staticmethod
    def backward(ctx, grad):
        hops_obj, K, widths, off = ctx.hops_obj, ctx.rounds, ctx.widths, ctx.off
        plan = hops_obj.plan
        H = plan.n_hops
        g_k = grad[:, off[K]:off[K] + widths[K]]
        for k in range(K, 0, -1):
            layer = hops_obj.pipeline(widths[k - 1])
            g_prev = _adjoint_reduce_scatter(layer, g_k.unflatten(1, (H, widths[k - 1])))
            g_prev = g_prev + grad[:, off[k - 1]:off[k - 1] + widths[k - 1]]
            g_k = g_prev
        return g_k, None, None, None, None


class ShardedHops:
    """One rank's view of ``adj_hops`` in a row-partitioned run: the local row block ``A_k[rows_p, :]`` of every hop
    matrix (a rectangular :class:`~h2gcn_amd.hops.HopPlan`, global column ids, transposed operands for the
    backward) plus the per-width exchange pipelines.  ``GCNLayer`` accepts it in place of a ``HopPlan``:
    ``layer(sharded_hops, x_local) -> [n_local, H, d]`` all-gathers ``x_local`` (feature-chunk pipelined) and
    aggregates; its backward is the shard adjoint followed by a reduce-scatter."""

    def __init__(self, plan, n_global: int, device, group: Optional[dist.ProcessGroup] = None, chunk_cols: int = 64,
                 max_chunks: int = 4, exchange: Optional[str] = None, partition: Optional[RowPartition] = None):
        #: "allgather" (RCCL) | "p2p" | "ipc_engine" | "ipc_kernel"; default from $H2GCN_EXCHANGE, else "allgather"
        self.exchange = exchange or os.environ.get("H2GCN_EXCHANGE", "allgather")
        self.plan = plan
        self.n_global = int(n_global)
        self.device = torch.device(device)
        self.group = group
        self.chunk_cols, self.max_chunks = int(chunk_cols), int(max_chunks)
        self.n_hops, self.n_rows, self.n_cols = plan.n_hops, plan.n_rows, plan.n_cols
        self.partition = partition
        self._pipes = {}
        #: why the IPC exchange this run asked for was replaced by the all-gather, if it was (see _ipc_exchange_mismatch)
        self.verify_fallback = None
        self._small = None          # (IpcExchange, scratch) of all_reduce_small
        self._small_views = {}
        if (self.exchange.startswith("ipc_") and self.device.type == "cuda" and dist.is_available() and dist.is_initialized()
                and dist.get_world_size(group) > 1 and os.environ.get("H2GCN_XCHG_VERIFY", "1") != "0"):
            why = self._ipc_exchange_mismatch()
            if why is not None:
                import warnings
                warnings.warn(f"h2gcn_amd: the IPC exchange ({self.exchange}) does not reproduce an all-gather of the same tensor on this node "
                              f"({why}); this run uses exchange='allgather' instead (and no hipGraph replay of row-partitioned steps)")
                self.exchange, self.verify_fallback = "allgather", why

    def pipeline(self, d: int) -> "PipelinedHopAggregation":
        if d not in self._pipes:
            chunks = 1
            while chunks * 2 <= self.max_chunks and d % (chunks * 2) == 0 and d // (chunks * 2) >= self.chunk_cols:
                chunks *= 2
            self._pipes[d] = PipelinedHopAggregation(self.plan, self.n_global, d, chunks, self.device, self.group,
                                                     exchange=self.exchange if self.device.type == "cuda" or not self.exchange.startswith("ipc_") else "allgather",
                                                     partition=self.partition,
                                                     # training: a rank may legitimately stall for a long time (first-epoch
                                                     # module loads, rank 0 writing a checkpoint) -- two minutes by default
                                                     ipc_timeout_ms=int(os.environ.get("H2GCN_XCHG_TIMEOUT_MS", "120000")))
        return self._pipes[d]

    def _ipc_exchange_mismatch(self) -> Optional[str]:
        """The run-time gate of csrc/exchange.hip ("Visibility across devices").  Collective, eager, once per run, BEFORE anything
        decides on hipGraph replay: FOUR different test patterns (both slot parities, each used twice -- the re-use of a slot is where
        a reader could be served a stale cached line) go through a small IPC exchange of the mode this run asked for (1 MiB per rank:
        L2-sized, the most exposed case) and through the process group's all-gather; every rank compares, the verdict is
        all-reduced.  None = identical everywhere; otherwise what differed (the same string on every rank)."""
        dev = self.device
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        per, w = 4096, 64
        xc = IpcExchange(1, per * w * 4, dev, self.group, mode=self.exchange[4:], timeout_ms=int(os.environ.get("H2GCN_XCHG_TIMEOUT_MS", "120000")))
        full = torch.empty((world * per, w), dtype=torch.float32, device=dev)
        ref = torch.empty_like(full)
        rows = torch.arange(rank * per, (rank + 1) * per, device=dev, dtype=torch.int64)[:, None]
        cols = torch.arange(w, device=dev, dtype=torch.int64)[None, :]
        bad = 0
        for trial in (1, 2, 3, 4):
            x = ((rows * 131 + cols * 7 + trial * 1000003) % 65521).to(torch.float32)
            full.fill_(-1.0)
            xc.begin(0, x, full, per)
            xc.end(0)
            _all_gather_rows(ref, x.contiguous(), self.group)
            torch.cuda.synchronize(dev)
            if not bad and not torch.equal(full, ref):
                bad = trial
        try:
            xc.check()
        except Exception:  # noqa: BLE001 -- a wait that gave up is a mismatch too (the blocks are NaN-poisoned)
            bad = bad or 99
        verdict = torch.tensor([bad], dtype=torch.int64, device=dev)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX, group=self.group)
        xc.close()                                       # collective
        v = int(verdict.item())
        if v == 0:
            return None
        return "a wait of the exchange gave up" if v == 99 else f"gathered bytes differ from test pattern {v} on"

    def check(self) -> None:
        """Raise if any exchange of any pipeline ever timed out (see :meth:`PipelinedHopAggregation.check`); the step
        closures of a row-partitioned run call this every step."""
        for pipe in self._pipes.values():
            pipe.check()
        if self._small is not None:
            self._small[0].check()

    @property
    def capturable(self) -> bool:
        """True when a whole training / evaluation step of this shard can be captured into a hipGraph: every exchange
        goes through the library's copy-kernel IPC path (device-side sequence numbers), none through torch.distributed."""
        return self.exchange == "ipc_kernel" and self.device.type == "cuda"

    def all_reduce_small(self, flat: torch.Tensor) -> torch.Tensor:
        """Sum a small contiguous fp32 vector over the ranks through the IPC exchange (see ``IpcExchange.all_reduce_sum``)."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return flat
        n = flat.numel()
        if self._small is None or self._small[1].shape[1] < n:
            if self._small is not None:
                self._small[0].close()
            cap = max(n, 1 << 16)
            xc = IpcExchange(1, cap * 4, self.device, self.group, mode="kernel",
                             timeout_ms=int(os.environ.get("H2GCN_XCHG_TIMEOUT_MS", "120000")))
            self._small = (xc, torch.empty((world, cap), dtype=torch.float32, device=self.device))
        xc, buf = self._small
        if buf.shape[1] != n:   # the exchange moves whole rows of the scratch: keep its width == n
            buf = self._small_views.setdefault(n, torch.empty((world, n), dtype=torch.float32, device=self.device))
        return xc.all_reduce_sum(0, flat, buf)

    def prepare_capture(self) -> None:
        """Call (collectively, device idle) right before a hipGraph capture of a step that uses this shard."""
        torch.cuda.synchronize(self.device)
        for pipe in self._pipes.values():
            for xc in (pipe.ipc, getattr(pipe, "ipc_rs", None)):
                if xc is not None:
                    xc.reset_dependencies()
        if self._small is not None:
            self._small[0].reset_dependencies()

    def close(self) -> None:
        """Collective: release the IPC-exported buffers of every pipeline (after a barrier: no peer may still be pulling)."""
        for d in sorted(self._pipes):
            self._pipes[d].close()
        self._pipes = {}
        if self._small is not None:
            self._small[0].close()
            self._small = None

    def aggregate(self, x_local: torch.Tensor, hops=None) -> torch.Tensor:
        """``GCNLayer(hops=...)`` on the shard: ``hops`` keeps the listed hop indices (unknown ones are ignored like the
        reference's ``if ind in self.hops`` filter, ``_layers.py:80-81``)."""
        sel = None
        if hops is not None:
            sel = tuple(h for h in range(self.n_hops) if h in set(int(v) for v in hops))
            if not sel:
                raise ValueError(f"GCNLayer(hops={sorted(hops)}) selects none of the {self.n_hops} hops")
        return sharded_hop_spmm(self.pipeline(int(x_local.shape[1])), x_local, sel)

    def fused_propagation(self, r0_local: torch.Tensor, rounds: int, out: Optional[torch.Tensor] = None,
                          reuse: bool = False) -> torch.Tensor:
        """``[r_K | r_0 | ... | r_{K-1}]`` of this rank's rows without intermediate copies (see
        :class:`_ShardedFusedPropagation`); ``out`` / ``reuse`` as in :func:`h2gcn_amd.layers.fused_propagation` (every rank must
        make the same choice: a reusing rank takes no part in the exchange)."""
        if rounds < 1 or r0_local.dim() != 2 or r0_local.shape[0] != self.n_rows:
            raise ValueError(f"r0 must be [{self.n_rows}, d] and rounds >= 1")
        if reuse and out is None:
            raise ValueError("fused_propagation: reuse=True needs the buffer that holds the propagation (out=)")
        if r0_local.requires_grad and torch.is_grad_enabled():
            return _ShardedFusedPropagation.apply(r0_local, self, rounds, out, reuse)

        class _Ctx:
            pass
        return _ShardedFusedPropagation.forward(_Ctx(), r0_local, self, rounds, out, reuse)


def slice_csr_rows(rowptr: torch.Tensor, colidx: torch.Tensor, vals: torch.Tensor, r0: int, r1: int):
    """Rows [r0, r1) of a device CSR (global column ids kept)."""
    lo, hi = int(rowptr[r0]), int(rowptr[r1])
    return (rowptr[r0:r1 + 1] - lo).contiguous(), colidx[lo:hi].contiguous(), vals[lo:hi].contiguous()
