"""``FlowSampler`` — packed device-side handle of one (material, domain) flow sampler.

Thin host object over the C ABI (include/bsdfd.h).  PyTorch is used only for device
memory and the current HIP stream; all arithmetic happens in the HIP kernels of
csrc/bsdfd.hip.  Checks mirror the contiguity/dtype/device checks of the reference's
only native binding (tiny-cuda-nn/bindings/torch/tinycudann/bindings.cpp:54-73).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import weights as W


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _i64(v: int) -> int:
    """uint64 seed / offset -> the int64 the operator schema carries (same 64 bits)."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def default_binding() -> str:
    """'torch' (the operator library torch.ops.bsdfd.*, csrc/torch_ops.cpp) when it has been built, else 'ctypes';
    $BSDFD_HOST_BINDING overrides.  Both are host shims over the same C ABI and the same kernels."""
    import os
    from . import torch_ext
    b = os.environ.get("BSDFD_HOST_BINDING")
    if b:
        if b not in ("ctypes", "torch"):
            raise RuntimeError(f"BSDFD_HOST_BINDING must be 'ctypes' or 'torch', got {b!r}")
        return b
    return "torch" if torch_ext.available() else "ctypes"


class FlowSampler:
    def __init__(self, fw: "W.FlowWeights | str", precision: str = "default", device: Optional[int] = None,
                 binding: Optional[str] = None, tile: int = 0):
        """``binding``: 'ctypes' or 'torch' — which host shim the per-call entry points go through (default:
        ``default_binding()``).  The handle is created through the C ABI either way and is the same object.
        ``tile``: bsdfd_desc.tile — 0 (default), 16 or 32 queries per wave tile.  With 0, ``$BSDFD_TILE=16`` / ``=32`` select the
        16- / 32-query kernels (A/B runs of one build; the LIBRARY reads no environment variable — this host maps it onto the
        field; 32 also selects the opt-in 32-query kernel of the 64 x 6 net and falls back to the default where none exists).
        An explicit ``tile=32`` ARGUMENT for a net / precision without a 32-query kernel raises."""
        if not torch.cuda.is_available():
            raise RuntimeError("FlowSampler needs an MI355X (torch.cuda is unavailable); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if isinstance(fw, str):
            fw = W.load(fw)
        self.weights = fw.validate()
        self.domain = fw.domain
        L = _lib.lib()
        d = _lib.Desc()
        d.domain, d.width, d.n_hidden, d.pe_bands = fw.domain, fw.width, fw.n_hidden, fw.pe_bands
        d.base_hidden, d.base_pe_bands = fw.base_hidden, fw.base_pe_bands
        d.precision = _lib.PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        d.tile = int(tile)
        env_tile = 0
        if d.tile == 0:
            env_tile = {"16": 16, "32": 32}.get(os.environ.get("BSDFD_TILE", "").strip(), 0)
            d.tile = env_tile   # (32 from the ENVIRONMENT is a preference: also selects the opt-in 32-query kernels, falls back below)
        keep = []
        for name in ("w_in", "w_hidden", "w_out", "base_w1", "base_b1", "base_w2", "base_b2"):
            a = np.ascontiguousarray(getattr(fw, name), dtype=np.float32)
            keep.append(a)
            setattr(d, name, a.ctypes.data_as(C.POINTER(C.c_float)))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = L.bsdfd_create(C.byref(d), C.byref(h))
            if rc != 0 and env_tile == 32 and int(tile) == 0 and b"tile = 32" in L.bsdfd_last_error():
                d.tile = 0   # this net / precision has no 32-query kernel: $BSDFD_TILE=32 then means the library's default
                rc = L.bsdfd_create(C.byref(d), C.byref(h))
            _lib.check(rc)
        self._h = h
        self._L = L
        self.binding = binding or default_binding()
        if self.binding not in ("ctypes", "torch"):
            raise RuntimeError(f"binding must be 'ctypes' or 'torch', got {self.binding!r}")
        self._ops = None
        if self.binding == "torch":
            from . import torch_ext
            self._ops = torch_ext.load()
            self._hi = int(h.value)
        p = C.c_int32()
        _lib.check(L.bsdfd_get_info(h, None, None, None, C.byref(p)))
        self.precision = {v: k for k, v in _lib.PRECISIONS.items()}[p.value]
        _lib.check(L.bsdfd_get_tile(h, 0, C.byref(p)))
        self.tile = int(p.value)               # queries per wave tile of the sample / pdf kernels in effect (16 | 32)
        _lib.check(L.bsdfd_get_tile(h, 2, C.byref(p)))
        self.tile_samples_only = int(p.value)  # ... and of the kernel behind flow_samples_only

    def close(self):
        if getattr(self, "_h", None):
            self._L.bsdfd_destroy(self._h)
            self._h = None
            self._hi = 0  # the operator library rejects a null handle (TORCH_CHECK) instead of touching freed memory

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _chk(self, t: Optional[torch.Tensor], cols: int, name: str, n: Optional[int] = None):
        if t is None:
            return None
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA (HIP) tensor")
        if t.device != self.device:
            raise RuntimeError(f"{name} is on {t.device}, the sampler on {self.device}")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be float32, got {t.dtype}")
        if t.dim() != 2 or t.shape[1] != cols:
            raise RuntimeError(f"{name} must have shape [N, {cols}], got {tuple(t.shape)}")
        if n is not None and t.shape[0] != n:
            raise RuntimeError(f"{name} has {t.shape[0]} rows, expected {n}")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        return t

    def _dev_chk(self, t, name: str):
        """torch binding: dtype / shape / contiguity are checked in C++ (csrc/torch_ops.cpp); the one thing the
        operator cannot know is which device the HANDLE lives on."""
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise RuntimeError(f"{name} must be a tensor on {self.device} (the sampler's device)")

    def _chk1(self, t: torch.Tensor, n: int, name: str):
        """A caller-supplied 1-D output (pdf [N]): the kernel writes N floats through its raw pointer."""
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device != self.device:
            raise RuntimeError(f"{name} must be a CUDA (HIP) tensor on {self.device}")
        if t.dtype != torch.float32 or t.dim() != 1 or t.shape[0] != n or not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous float32 tensor of shape [{n}], got {t.dtype} {tuple(t.shape)}")
        return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- per-query context (bsdfd_context_bytes / bsdfd_plugin_{sample,pdf}_ctx) ----
    def context_floats(self, n: int, n_segments: int = 1) -> int:
        """Size (in float32 elements) of the opaque per-query context of an n-query call."""
        if not getattr(self, "_h", None):
            raise RuntimeError("bsdfd: null handle")
        b = int(self._L.bsdfd_context_bytes(self._h, n, n_segments))
        if b < 0:
            raise RuntimeError("bsdfd_context_bytes: bad arguments")
        return b // 4

    def new_context(self, n: int, n_segments: int = 1) -> torch.Tensor:
        """Device buffer a ``plugin_sample(..., ctx_out=)`` call fills and ``plugin_pdf(..., ctx_in=)`` reads."""
        return torch.empty((self.context_floats(n, n_segments),), dtype=torch.float32, device=self.device)

    def _chk_ctx(self, ctx: torch.Tensor, n: int, n_segments: int = 1) -> torch.Tensor:
        need = self.context_floats(n, n_segments)
        if (not isinstance(ctx, torch.Tensor) or ctx.device != self.device or ctx.dtype != torch.float32 or ctx.dim() != 1
                or not ctx.is_contiguous() or ctx.numel() < need or ctx.data_ptr() % 16):
            raise RuntimeError(f"context must be a contiguous, 16-byte aligned float32 tensor of >= {need} elements on "
                               f"{self.device} (FlowSampler.new_context)")
        return ctx

    def flops_per_query(self, T: int) -> int:
        return int(self._L.bsdfd_flops_per_query(self._h, T))

    def set_profiling(self, on: bool):
        _lib.check(self._L.bsdfd_set_profiling(self._h, 1 if on else 0))

    def profile_read(self):
        """(launches, total kernel ms) since ``set_profiling(True)`` — HIP events on the launch stream."""
        n, ms = C.c_int64(), C.c_double()
        _lib.check(self._L.bsdfd_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_read_op(self, op: str):
        """(launches, total kernel ms) of ONE kind of launch since ``set_profiling(True)``: ``op`` is "sample", "pdf",
        "samples_only" or "sample_pdf" (bsdfd_profile_read_op)."""
        n, ms = C.c_int64(), C.c_double()
        code = {"sample": 0, "pdf": 1, "samples_only": 2, "sample_pdf": 3}[op]
        _lib.check(self._L.bsdfd_profile_read_op(self._h, code, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_clock_mhz(self) -> float:
        """Shader clock (MHz) the chip sustained under THIS handle's launches since ``set_profiling(True)``: the waves' own
        shader-cycle / wall-clock counters (bsdfd_profile_clock_mhz); 0 if none were recorded."""
        mhz = C.c_double()
        _lib.check(self._L.bsdfd_profile_clock_mhz(self._h, C.byref(mhz)))
        return mhz.value

    def last_kernel_ms(self) -> float:
        return float(self._L.bsdfd_last_kernel_ms(self._h))

    # ---- operator level (rendering/utils/mlp_brdf_sampling.py) -------
    def network_sampling(self, omega_i, x0=None, T: int = 4, seed: int = 0, offset: int = 0
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._ops is not None:
            self._dev_chk(omega_i, "omega_i")
            return self._ops.network_sampling(self._hi, omega_i, x0, _i64(seed), _i64(offset), T)
        omega_i = self._chk(omega_i, 2, "omega_i")
        n = omega_i.shape[0]
        x0 = self._chk(x0, 2, "x0", n)
        x = torch.empty((n, 2), dtype=torch.float32, device=self.device)
        pdf = torch.empty((n,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_network_sampling(self._h, _ptr(omega_i), _ptr(x0), seed, offset, n, T,
                                                      _ptr(x), _ptr(pdf), self._stream()))
        return x, pdf

    def network_pdf(self, omega_o, omega_i, T: int = 4) -> torch.Tensor:
        if self._ops is not None:
            self._dev_chk(omega_i, "omega_i")
            return self._ops.network_pdf(self._hi, omega_o, omega_i, T)
        omega_i = self._chk(omega_i, 2, "omega_i")
        n = omega_i.shape[0]
        omega_o = self._chk(omega_o, 2, "omega_o", n)
        pdf = torch.empty((n,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_network_pdf(self._h, _ptr(omega_o), _ptr(omega_i), n, T, _ptr(pdf),
                                                 self._stream()))
        return pdf

    def flow_samples_only(self, omega_i, x0, T: int) -> torch.Tensor:
        if self._ops is not None:
            self._dev_chk(omega_i, "omega_i")
            return self._ops.flow_samples_only(self._hi, omega_i, x0, T)
        omega_i = self._chk(omega_i, 2, "omega_i")
        n = omega_i.shape[0]
        x0 = self._chk(x0, 2, "x0", n)
        x = torch.empty((n, 2), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_flow_samples_only(self._h, _ptr(omega_i), _ptr(x0), n, T, _ptr(x),
                                                       self._stream()))
        return x

    # ---- plugin level (tensor core of MyBSDF.sample / MyBSDF.pdf) ----
    def plugin_sample(self, wi, x0=None, T: int = 4, variant: int = _lib.PLUGIN_MEASURED, seed: int = 0,
                      offset: int = 0, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                      ctx_out: Optional[torch.Tensor] = None, rng_index: Optional[torch.Tensor] = None,
                      ctx_in: Optional[torch.Tensor] = None, row_index: Optional[torch.Tensor] = None):
        """``ctx_out`` (``new_context(N)``): also write the per-query context a later ``plugin_pdf(wi, ., ctx_in=)`` /
        ``plugin_sample(wi, ctx_in=)`` for the SAME ``wi`` reads instead of recomputing the prologue (identical results);
        ``ctx_in``: read the context an earlier call (``plugin_pdf(..., ctx_out=)`` or ``plugin_sample(..., ctx_out=)``) wrote
        for this very ``wi``.  At most one of the two.
        ``rng_index`` (int64 [N]): the Philox counter of row i is ``offset + rng_index[i]`` instead of ``offset + i``
        (a bucketed wavefront passes the rows' original lane indices: draws independent of the bucketing).
        ``row_index`` (int64 [n], n <= N, distinct entries < N): the call processes n rows; row i reads ``wi`` / ``x0`` at row
        ``row_index[i]`` and writes ``wo`` / ``pdf`` there (bsdfd_opts.row_index: the gather / scatter of a bucketed wavefront
        inside the kernel's own loads and stores); rows it does not name are left as they are (zeros when ``out`` is None).
        Its Philox counter is ``offset + row_index[i]`` unless ``rng_index`` is given."""
        if ctx_out is not None and ctx_in is not None:
            raise RuntimeError("a call either writes a per-query context (ctx_out) or reads one (ctx_in), not both")
        if ctx_out is not None or ctx_in is not None or rng_index is not None or row_index is not None:
            return self._plugin_sample_ex(wi, x0, T, variant, seed, offset, out, ctx_out if ctx_in is None else ctx_in, rng_index,
                                          ctx_read=ctx_in is not None, row_index=row_index)
        if self._ops is not None:
            self._dev_chk(wi, "wi")
            if out is None:
                return self._ops.plugin_sample(self._hi, variant, wi, x0, _i64(seed), _i64(offset), T)
            self._ops.plugin_sample_out(self._hi, variant, wi, x0, _i64(seed), _i64(offset), T, out[0], out[1])
            return out[0], out[1]
        wi = self._chk(wi, 3, "wi")
        n = wi.shape[0]
        x0 = self._chk(x0, 2, "x0", n)
        if out is None:
            wo = torch.empty((n, 3), dtype=torch.float32, device=self.device)
            pdf = torch.empty((n,), dtype=torch.float32, device=self.device)
        else:
            wo, pdf = self._chk(out[0], 3, "out wo", n), self._chk1(out[1], n, "out pdf")
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_plugin_sample(self._h, variant, _ptr(wi), _ptr(x0), seed, offset, n, T,
                                                   _ptr(wo), _ptr(pdf), self._stream()))
        return wo, pdf

    def _chk_index(self, idx, n, name="rng_index"):
        if idx is None:
            return None
        if (not isinstance(idx, torch.Tensor) or idx.device != self.device or idx.dtype != torch.int64 or idx.dim() != 1
                or (n is not None and idx.shape[0] != n) or not idx.is_contiguous()):
            raise RuntimeError(f"{name} must be a contiguous int64 tensor" + (f" of shape [{n}]" if n is not None else "") +
                               f" on {self.device}")
        return idx

    def _rows(self, row_index, m):
        """Rows a call processes: all m of the arrays, or the rows a row_index names."""
        self._chk_index(row_index, None, "row_index")
        if row_index is not None and row_index.shape[0] > m:
            raise RuntimeError(f"row_index names {row_index.shape[0]} rows, the arrays have {m}")
        if row_index is not None and row_index.numel() and os.environ.get("BSDFD_CHECK_INDEX", "0") not in ("", "0"):
            # debugging aid (one device->host sync per call): the library cannot check this, it does not know the arrays' lengths
            lo, hi = int(row_index.min()), int(row_index.max())
            if lo < 0 or hi >= m:
                raise RuntimeError(f"row_index entries span [{lo}, {hi}], the arrays have rows [0, {m})")
            if int(torch.unique(row_index).numel()) != row_index.numel():
                raise RuntimeError("row_index names a row more than once")
        return m if row_index is None else row_index.shape[0]

    def _plugin_sample_ex(self, wi, x0, T, variant, seed, offset, out, ctx, rng_index, ctx_read=False, row_index=None):
        mk = torch.empty if row_index is None else torch.zeros   # (rows a row_index does not name stay untouched)
        if self._ops is not None:
            self._dev_chk(wi, "wi")
            n = self._rows(row_index, wi.shape[0])
            if ctx is not None:
                self._chk_ctx(ctx, n)
            self._chk_index(rng_index, n)
            if out is None:
                out = (mk((wi.shape[0], 3), dtype=torch.float32, device=self.device),
                       mk((wi.shape[0],), dtype=torch.float32, device=self.device))
            self._ops.plugin_sample_ex_out(self._hi, variant, wi, x0, _i64(seed), _i64(offset), T, out[0], out[1], ctx, rng_index,
                                           ctx_read, row_index)
            return out[0], out[1]
        wi = self._chk(wi, 3, "wi")
        m = wi.shape[0]
        n = self._rows(row_index, m)
        x0 = self._chk(x0, 2, "x0", m)
        if ctx is not None:
            self._chk_ctx(ctx, n)
        self._chk_index(rng_index, n)
        if out is None:
            wo = mk((m, 3), dtype=torch.float32, device=self.device)
            pdf = mk((m,), dtype=torch.float32, device=self.device)
        else:
            wo, pdf = self._chk(out[0], 3, "out wo", m), self._chk1(out[1], m, "out pdf")
        o = (_lib.opts(ctx_in=ctx, rng_index=rng_index, row_index=row_index) if ctx_read
             else _lib.opts(ctx_out=ctx, rng_index=rng_index, row_index=row_index))
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_plugin_sample_ex(self._h, variant, _ptr(wi), _ptr(x0), seed, offset, n, T,
                                                      _ptr(wo), _ptr(pdf), C.byref(o), self._stream()))
        return wo, pdf

    def plugin_sample_pdf(self, wi, wl, x0=None, T: int = 4, variant: int = _lib.PLUGIN_MEASURED, seed: int = 0,
                          offset: int = 0, out: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
        """sample(wi) and pdf(wi, wl) of the same intersections in ONE launch (the per-query prologue is
        shared): -> (wo [N,3], pdf(wo) [N], pdf(wl) [N]), identical to plugin_sample + plugin_pdf(wi, wl)."""
        if self._ops is not None and out is None:
            self._dev_chk(wi, "wi")
            return self._ops.plugin_sample_pdf(self._hi, variant, wi, wl, x0, _i64(seed), _i64(offset), T)
        wi = self._chk(wi, 3, "wi")
        n = wi.shape[0]
        wl = self._chk(wl, 3, "wl", n)
        x0 = self._chk(x0, 2, "x0", n)
        if out is None:
            wo = torch.empty((n, 3), dtype=torch.float32, device=self.device)
            pdf_o = torch.empty((n,), dtype=torch.float32, device=self.device)
            pdf_l = torch.empty((n,), dtype=torch.float32, device=self.device)
        else:
            wo, pdf_o, pdf_l = (self._chk(out[0], 3, "out wo", n), self._chk1(out[1], n, "out pdf(wo)"),
                                self._chk1(out[2], n, "out pdf(wl)"))
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_plugin_sample_pdf(self._h, variant, _ptr(wi), _ptr(x0), _ptr(wl), seed, offset,
                                                       n, T, _ptr(wo), _ptr(pdf_o), _ptr(pdf_l), self._stream()))
        return wo, pdf_o, pdf_l

    def plugin_pdf(self, wi, wo, T: int = 4, variant: int = _lib.PLUGIN_MEASURED,
                   out: Optional[torch.Tensor] = None, ctx_in: Optional[torch.Tensor] = None,
                   ctx_out: Optional[torch.Tensor] = None, row_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``ctx_in``: the context an earlier ``plugin_sample`` / ``plugin_pdf`` call wrote (``ctx_out=``) for this very ``wi``
        array; ``ctx_out`` (``new_context(N)``): write it here (the call order of Mitsuba's path integrator: eval_pdf() for the
        emitter sample first, sample() second — rendering/brdf_measured_disk.py:126,59).  At most one of the two.
        ``row_index``: as in ``plugin_sample`` (row i reads wi / wo at row ``row_index[i]`` and writes pdf there)."""
        if ctx_out is not None and ctx_in is not None:
            raise RuntimeError("a call either writes a per-query context (ctx_out) or reads one (ctx_in), not both")
        if ctx_in is not None or ctx_out is not None or row_index is not None:
            ctx, write = (ctx_in, False) if ctx_out is None else (ctx_out, True)
            mk = torch.empty if row_index is None else torch.zeros
            if self._ops is not None:
                self._dev_chk(wi, "wi")
                n = self._rows(row_index, wi.shape[0])
                if ctx is not None:
                    self._chk_ctx(ctx, n)
                if out is None:
                    out = mk((wi.shape[0],), dtype=torch.float32, device=self.device)
                self._ops.plugin_pdf_ex_out(self._hi, variant, wi, wo, T, out, ctx, write, row_index)
                return out
            wi = self._chk(wi, 3, "wi")
            m = wi.shape[0]
            n = self._rows(row_index, m)
            wo = self._chk(wo, 3, "wo", m)
            if ctx is not None:
                self._chk_ctx(ctx, n)
            pdf = mk((m,), dtype=torch.float32, device=self.device) if out is None else self._chk1(out, m, "out pdf")
            o = _lib.opts(ctx_out=ctx, row_index=row_index) if write else _lib.opts(ctx_in=ctx, row_index=row_index)
            with torch.cuda.device(self.device):
                _lib.check(self._L.bsdfd_plugin_pdf_ex(self._h, variant, _ptr(wi), _ptr(wo), n, T, _ptr(pdf),
                                                       C.byref(o), self._stream()))
            return pdf
        if self._ops is not None:
            self._dev_chk(wi, "wi")
            if out is None:
                return self._ops.plugin_pdf(self._hi, variant, wi, wo, T)
            self._ops.plugin_pdf_out(self._hi, variant, wi, wo, T, out)
            return out
        wi = self._chk(wi, 3, "wi")
        n = wi.shape[0]
        wo = self._chk(wo, 3, "wo", n)
        pdf = torch.empty((n,), dtype=torch.float32, device=self.device) if out is None else self._chk1(out, n, "out pdf")
        with torch.cuda.device(self.device):
            _lib.check(self._L.bsdfd_plugin_pdf(self._h, variant, _ptr(wi), _ptr(wo), n, T, _ptr(pdf),
                                                self._stream()))
        return pdf
