"""HopPlan -- the device-resident ``adj_hops`` operand list of H2GCN and its launch plan.

Mirror of what the reference keeps in ``args.objects["tensors"]["adj_hops"]``: a Python list of H normalised
``tf.SparseTensor`` built once before training (reference ``h2gcn/models/H2GCN.py:46-54`` ->
``h2gcn/datasets/_dataset.py:559-576``, conversion ``sparse2Tensor`` ``:528-535``).  Here the list is one object:
CSR arrays on the GPU (int64 row pointers, int32 column ids in ascending order per row -- the canonical order
``tf.sparse.reorder`` establishes -- fp32 values) plus the opaque plan of ``libh2gcn_hip.so`` (the segment-class bins of the
CSR-adaptive schedule: lists of the long and of the short (row, hop) segments, see :meth:`HopPlan.segment_classes`;
optional transposed operands for the backward pass).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import torch

from . import _capi


def _require(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)


class HopPlan:
    """H hop matrices sharing one row space, resident on one GPU.

    Results are bit-reproducible functions of the operands: every launch builds the library's canonical per-row
    summation tree (``include/h2gcn_hip.h``), whatever slice width, scratch copy or segment walk the schedule picks.

    Parameters
    ----------
    rowptr, colidx, vals : sequences of H CUDA tensors (int64 ``[n_rows+1]``, int32 ``[nnz]``, float32 ``[nnz]``)
    n_cols : number of columns (rows of the dense operand ``X``)
    build_transpose : also build ``A_k^T`` on the device (needed for ``backward``)
    long_row_threshold, rows_per_wave, variant, slice_cols : schedule tunables (0 = library default)
    validate : run the one-time column-range check (TensorFlow validates indices per call; this once)
    """

    def __init__(self, rowptr: Sequence[torch.Tensor], colidx: Sequence[torch.Tensor],
                 vals: Sequence[torch.Tensor], n_cols: int, *, build_transpose: bool = False,
                 long_row_threshold: int = 0, rows_per_wave: int = 0, variant: int = 0,
                 slice_cols: int = 0, validate: bool = True, host_transpose: bool = False,
                 keep_permutation: bool = False):
        H = len(rowptr)
        _require(1 <= H <= _capi.MAX_HOPS, f"need 1..{_capi.MAX_HOPS} hop matrices, got {H}")
        _require(len(colidx) == H and len(vals) == H, "rowptr/colidx/vals lists differ in length")
        dev = rowptr[0].device
        _require(dev.type == "cuda", f"HopPlan operands must live on a GPU, got {dev} (no CPU fallback)")
        n_rows = rowptr[0].numel() - 1
        _require(n_rows >= 0, "rowptr must have n_rows+1 entries")
        for k in range(H):
            rp, ci, va = rowptr[k], colidx[k], vals[k]
            _require(rp.dtype == torch.int64 and ci.dtype == torch.int32 and va.dtype == torch.float32,
                     f"hop {k}: dtypes must be int64/int32/float32, got {rp.dtype}/{ci.dtype}/{va.dtype}")
            _require(rp.device == dev and ci.device == dev and va.device == dev, f"hop {k}: operands on different devices")
            _require(rp.dim() == 1 and rp.numel() == n_rows + 1, f"hop {k}: rowptr has {rp.numel()} entries, expected {n_rows + 1}")
            _require(ci.dim() == 1 and va.dim() == 1 and ci.numel() == va.numel(), f"hop {k}: colidx/vals sizes differ")
            _require(rp.is_contiguous() and ci.is_contiguous() and va.is_contiguous(), f"hop {k}: operands must be contiguous")
        self.n_hops = H
        self.n_rows = int(n_rows)
        self.n_cols = int(n_cols)
        self.device = dev
        self.has_transpose = bool(build_transpose)
        # the plan borrows these arrays: keep them alive
        self.rowptr = list(rowptr)
        self.colidx = list(colidx)
        self.vals = list(vals)
        self._handle = C.c_void_p()
        #: let launches use scratch memory for the slice-major copy of X (see h2gcn_spmm_workspace_bytes)
        self.use_workspace = True
        #: narrowest feature chunk a pipeline may cut: any width gives the same bits (one canonical summation tree in every
        #: kernel); below 16 columns a chunk is just not worth its extra pass over the indices
        self.min_chunk_cols = 16

        L = _capi.lib()
        arr_t = C.c_void_p * H
        rp_a = arr_t(*[t.data_ptr() for t in self.rowptr])
        ci_a = arr_t(*[t.data_ptr() for t in self.colidx])
        va_a = arr_t(*[t.data_ptr() for t in self.vals])
        opts = _capi.PlanOpts()
        opts.struct_size = C.sizeof(_capi.PlanOpts)
        opts.flags = ((_capi.PLAN_BUILD_TRANSPOSE if build_transpose else 0) | (0 if validate else _capi.PLAN_SKIP_VALIDATION)
                      | (_capi.PLAN_HOST_TRANSPOSE if host_transpose else 0)
                      | (_capi.PLAN_KEEP_PERMUTATION if keep_permutation and build_transpose else 0))
        opts.long_row_threshold = int(long_row_threshold)
        opts.rows_per_wave = int(rows_per_wave)
        opts.variant = int(variant)
        opts.slice_cols = int(slice_cols)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            st = L.h2gcn_plan_create(H, self.n_rows, self.n_cols, rp_a, ci_a, va_a, C.byref(opts),
                                     C.c_void_p(stream), C.byref(self._handle))
        _capi.check(st)

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_scipy(cls, mats: Iterable, device, **kw) -> "HopPlan":
        """Upload a list of scipy sparse matrices (any format; cast to fp32 CSR with sorted indices -- the
        same cast/ordering ``sparse2Tensor`` applies, reference ``h2gcn/datasets/_dataset.py:528-535``)."""
        import numpy as np
        import scipy.sparse as sp

        rowptr, colidx, vals = [], [], []
        n_cols = None
        n_rows = None
        for m in mats:
            m = sp.csr_matrix(m)
            m.sum_duplicates()
            m.sort_indices()
            _require(n_cols in (None, m.shape[1]) and n_rows in (None, m.shape[0]), "hop matrices differ in shape")
            n_rows, n_cols = m.shape
            rowptr.append(torch.from_numpy(m.indptr.astype(np.int64)).to(device))
            colidx.append(torch.from_numpy(m.indices.astype(np.int32)).to(device))
            vals.append(torch.from_numpy(m.data.astype(np.float32)).to(device))
        _require(n_cols is not None, "empty hop list")
        return cls(rowptr, colidx, vals, n_cols, **kw)

    def set_values(self, hop: int, vals: torch.Tensor) -> None:
        """New values for hop ``hop`` (same pattern); the transposed operand is refreshed too (plans built with
        ``keep_permutation=True``).  ``vals`` is borrowed like the original arrays.  See ``h2gcn_plan_set_values``."""
        _require(vals.dtype == torch.float32 and vals.device == self.device and vals.is_contiguous()
                 and vals.numel() == self.colidx[hop].numel(), "vals must be a contiguous float32 array of the hop's nnz")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _capi.check(_capi.lib().h2gcn_plan_set_values(self._handle, int(hop), C.c_void_p(vals.data_ptr()), C.c_void_p(stream)))
        self.vals[hop] = vals
        self.values_version = getattr(self, "values_version", 0) + 1   # anything cached as a function of the values is stale

    # ------------------------------------------------------------------ introspection
    @property
    def nnz(self) -> list:
        return [int(t.numel()) for t in self.colidx]

    def info(self, hop: int) -> dict:
        L = _capi.lib()
        n_rows, n_cols, nnz, n_long = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        has_t = C.c_int32()
        _capi.check(L.h2gcn_plan_info(self._handle, hop, C.byref(n_rows), C.byref(n_cols), C.byref(nnz),
                                      C.byref(n_long), C.byref(has_t)))
        return dict(n_rows=n_rows.value, n_cols=n_cols.value, nnz=nnz.value, n_long_segments=n_long.value,
                    has_transpose=bool(has_t.value))

    def schedule(self, d: int, ld_src: Optional[int] = None, hops=None, adjoint: bool = False) -> dict:
        """What a launch at feature width ``d`` (source row stride ``ld_src``, default contiguous) would do."""
        L = _capi.lib()
        sc, ns, pf, cp = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        ld = int(ld_src) if ld_src is not None else (self.n_selected(hops) * d if adjoint else d)
        _capi.check(L.h2gcn_plan_schedule(self._handle, self._mask(hops), 1 if adjoint else 0, ld, int(d),
                                          C.byref(sc), C.byref(ns), C.byref(pf), C.byref(cp)))
        return dict(slice_cols=sc.value, n_slices=ns.value,
                    segment_walk={0: "wave per segment", 1: "wave per segment + index prefetch",
                                  2: "lane group per segment (short rows)",
                                  3: "lane group per segment (binned short segments) + wave per segment"}[pf.value],
                    scratch_copy=bool(cp.value) and self.use_workspace)

    def segment_classes(self, d: int, ld_src: Optional[int] = None, hops=None, adjoint: bool = False) -> dict:
        """CSR-adaptive dispatch of a launch at width ``d``: per selected hop the number of short (<= 16 nonzeros) / medium /
        long (>= long_row_threshold) segments and their nonzeros, which walk serves each class, and how many segments the
        launch takes from the binned short list (``listed``; 0 = the wave walk serves the short class too, -1 = in-tile
        short-row mode: rounds of consecutive short rows are grouped on the fly)."""
        L = _capi.lib()
        if not _capi.has("h2gcn_plan_segment_classes"):
            raise RuntimeError(f"{_capi.library_path()} predates h2gcn_plan_segment_classes (ABI 4)")
        h_sel = self.n_selected(hops)
        seg, nnz, listed = (C.c_int64 * (3 * h_sel))(), (C.c_int64 * (3 * h_sel))(), C.c_int64()
        ld = int(ld_src) if ld_src is not None else (h_sel * d if adjoint else d)
        _capi.check(L.h2gcn_plan_segment_classes(self._handle, self._mask(hops), 1 if adjoint else 0, ld, int(d), seg, nnz, C.byref(listed)))
        sel = list(range(self.n_hops)) if hops is None else sorted({int(h) for h in hops})
        short_walk = ("lane group per segment (binned list)" if listed.value > 0 else
                      "lane group per segment (rounds of consecutive short rows)" if listed.value < 0 else "wave per segment")
        per_hop = [dict(hop=sel[s], segments=dict(short=seg[3 * s], medium=seg[3 * s + 1], long=seg[3 * s + 2]),
                        nonzeros=dict(short=nnz[3 * s], medium=nnz[3 * s + 1], long=nnz[3 * s + 2])) for s in range(h_sel)]
        return dict(per_hop=per_hop, listed=listed.value,
                    walks=dict(short=short_walk, medium="wave per segment", long="workgroup per segment (4 waves, LDS-staged)"))

    def _mask(self, hops) -> int:
        if hops is None:
            return 0
        mask = 0
        for h in hops:
            _require(0 <= int(h) < self.n_hops, f"hop index {h} outside 0..{self.n_hops - 1}")
            mask |= 1 << int(h)
        _require(mask != 0, "empty hop selection")
        return mask

    def n_selected(self, hops) -> int:
        return self.n_hops if hops is None else bin(self._mask(hops)).count("1")

    # ------------------------------------------------------------------ launches
    def spmm(self, x: torch.Tensor, hops=None, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
             relu: bool = False) -> torch.Tensor:
        """``out[i, s, :] = act(sum_j A_s[i, j] * x[j, :] + bias)`` for the selected hops -> ``[n_rows, H_sel, d]``
        (``bias`` [d] and ``relu`` are the optional fused epilogue of the store; default: the plain sum).

        ``out`` may be any fp32 tensor view of shape ``[n_rows, H_sel, d]`` whose last dim is contiguous
        (e.g. a column slice of a wider concat buffer)."""
        _require(x.dim() == 2, f"inputs must be [n_cols, d], got shape {tuple(x.shape)}")
        _require(x.dtype == torch.float32, f"inputs must be float32, got {x.dtype}")
        _require(x.device == self.device, f"inputs on {x.device}, plan on {self.device}")
        _require(x.shape[0] == self.n_cols, f"inputs have {x.shape[0]} rows, hop matrices have {self.n_cols} columns")
        d = int(x.shape[1])
        _require(d >= 1, "inputs need at least one column")
        if x.stride(1) != 1:
            x = x.contiguous()
        h_sel = self.n_selected(hops)
        if out is None:
            out = torch.empty((self.n_rows, h_sel, d), dtype=torch.float32, device=self.device)
        else:
            _require(out.dtype == torch.float32 and out.device == self.device, "out must be float32 on the plan's device")
            _require(tuple(out.shape) == (self.n_rows, h_sel, d), f"out has shape {tuple(out.shape)}, expected {(self.n_rows, h_sel, d)}")
            _require(d == 1 or out.stride(2) == 1, "out's last dimension must be contiguous")
        if self.n_rows == 0:
            return out
        L = _capi.lib()
        mask = self._mask(hops)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            # scratch for the slice-major copy of X the library wants when X's row stride is a multiple of 1 KiB, when
            # its rows are wide and not line-aligned, or when d % 4 != 0 / X is not 16-byte addressable (0 bytes
            # otherwise); a torch allocation, so it is stream-ordered and capturable in a hipGraph
            ws_bytes = int(L.h2gcn_spmm_workspace_bytes(self._handle, mask, 0, C.c_void_p(x.data_ptr()), x.stride(0), 0, d)) if self.use_workspace else 0
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device) if ws_bytes else None
            opts = None
            if ws is not None or bias is not None or relu:
                if bias is not None:
                    _require(bias.dtype == torch.float32 and bias.device == self.device and bias.numel() == d and bias.is_contiguous(),
                             f"bias must be a contiguous float32 [{d}] tensor on the plan's device")
                opts = _capi.LaunchOpts(struct_size=C.sizeof(_capi.LaunchOpts), flags=_capi.LAUNCH_RELU if relu else 0,
                                        workspace=ws.data_ptr() if ws is not None else None, workspace_bytes=ws_bytes,
                                        bias=bias.data_ptr() if bias is not None else None)
            st = L.h2gcn_spmm_hops_opts_f32(self._handle, mask, C.c_void_p(x.data_ptr()), x.stride(0), d,
                                            C.c_void_p(out.data_ptr()), out.stride(0), out.stride(1) if h_sel > 1 else d,
                                            C.byref(opts) if opts is not None else None, C.c_void_p(stream))
        _capi.check(st)
        return out

    def spmm_t(self, grad: torch.Tensor, hops=None, out: torch.Tensor = None, accumulate: bool = False) -> torch.Tensor:
        """Adjoint: ``dx[j, :] = sum_s sum_i A_s[i, j] * grad[i, s, :]`` -> ``[n_cols, d]``.  ``out``: write into this
        ``[n_cols, d]`` tensor (unit column stride, any row stride) instead of a new one; ``accumulate=True`` ADDS the
        result to what ``out`` holds (``H2GCN_LAUNCH_ACCUMULATE``: the `+=` of a gradient slot fused into the store)."""
        _require(self.has_transpose, "plan was built without build_transpose=True; backward is unavailable")
        h_sel = self.n_selected(hops)
        _require(grad.dim() == 3 and grad.shape[0] == self.n_rows and grad.shape[1] == h_sel,
                 f"grad must be [{self.n_rows}, {h_sel}, d], got {tuple(grad.shape)}")
        _require(grad.dtype == torch.float32 and grad.device == self.device, "grad must be float32 on the plan's device")
        d = int(grad.shape[2])
        if grad.stride(2) != 1 or grad.stride(0) < d or (h_sel > 1 and grad.stride(1) < d):
            grad = grad.contiguous()  # e.g. an expanded (stride-0) gradient coming out of a reduction
        if out is None:
            _require(not accumulate, "accumulate=True needs the tensor to accumulate into (out=)")
            dx = torch.empty((self.n_cols, d), dtype=torch.float32, device=self.device)
        else:
            _require(out.shape == (self.n_cols, d) and out.dtype == torch.float32 and out.device == self.device
                     and (out.stride(1) == 1 or d == 1) and (out.stride(0) >= d or self.n_cols <= 1),
                     f"out must be float32 [{self.n_cols}, {d}] on the plan's device with unit column stride")
            dx = out
        if self.n_cols == 0:
            return dx
        L = _capi.lib()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            mask = self._mask(hops)
            ld_row, ld_hop = (grad.stride(0) if self.n_rows > 0 else h_sel * d), grad.stride(1)
            # scratch: slice-major copy of the stacked gradient (same rules as the forward launch)
            ws_bytes = int(L.h2gcn_spmm_workspace_bytes(self._handle, mask, 1, C.c_void_p(grad.data_ptr()), ld_row, ld_hop, d)) if self.use_workspace else 0
            opts = None
            if ws_bytes or accumulate:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device) if ws_bytes else None
                opts = _capi.LaunchOpts(struct_size=C.sizeof(_capi.LaunchOpts), flags=_capi.LAUNCH_ACCUMULATE if accumulate else 0,
                                        workspace=ws.data_ptr() if ws is not None else None, workspace_bytes=ws_bytes, bias=None)
            st = L.h2gcn_spmm_hops_T_opts_f32(self._handle, mask, C.c_void_p(grad.data_ptr()), ld_row, ld_hop, d,
                                              C.c_void_p(dx.data_ptr()), dx.stride(0) if self.n_cols > 1 else d,
                                              C.byref(opts) if opts is not None else None, C.c_void_p(stream))
        _capi.check(st)
        return dx

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _capi.lib().h2gcn_plan_destroy(h)
            except Exception:
                pass
            self._handle = C.c_void_p()
