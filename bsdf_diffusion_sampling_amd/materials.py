"""Mixed-material batches (BASELINE.json configs[3]: "all paper measured BSDFs, mixed queries").

The reference binds one plugin instance per material and Mitsuba dispatches each wavefront
lane to its instance (one `sample()` call per material per bounce,
rendering/matpreview/disney_bsdf_array0_envmap.xml: 12 `mybsdf` instances).  Here a
``MaterialTable`` holds one packed device handle per material and serves a batch whose
queries carry a material id: queries are bucketed (stable sort by id) and results are
scattered back to the callers' order.  All disk nets together are ~160 KB of fp16 fragments
— the whole LDS — so keeping every material resident in every workgroup is not an option
(SURVEY.md §7 "mixed-material batches").  Instead ONE segmented launch
(``bsdfd_plugin_sample_multi``) serves all buckets whose handles share a kernel signature
(domain, width, depth, precision; up to 64 per launch): workgroups are dealt to the buckets
in proportion to their sizes and each loads only its own material's 16 KB image.  The 52
measured materials are 2 launches (27 disk + 25 spherical) instead of 52.
``segmented=False`` falls back to one launch per bucket (used by the tests as the
reference for the segmented path).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import ctypes as C

import torch

from . import _lib
from . import weights as W
from .sampler import FlowSampler
from .sharding import bucket_by_material


class MaterialTable:
    def __init__(self, stems: Sequence[str], precision: str = "default", tile: int = 0):
        """``stems`` are shipped weight-set names such as ``aniso_miro_7_rgb_disk`` or
        ``chm_orange_rgb_spherical`` (mixing domains is allowed: the domain, T and plugin
        variant are per material).  ``tile``: bsdfd_desc.tile for every material (0 = library default)."""
        self.stems = list(stems)
        self.samplers: List[FlowSampler] = []
        self.T: List[int] = []
        self.variant: List[int] = []
        for stem in self.stems:
            fw = W.load(W.shipped_path(*self._split(stem)))
            self.samplers.append(FlowSampler(fw, precision=precision, tile=tile))
            disk = fw.domain == W.DOMAIN_DISK
            self.T.append(4 if disk else 8)  # plugin defaults, brdf_measured_disk.py:68 / _spherical.py:78
            self.variant.append(_lib.PLUGIN_FULLSPHERE if stem.startswith("bsdf_") else _lib.PLUGIN_MEASURED)

    @staticmethod
    def _split(stem: str) -> Tuple[str, str]:
        for dom in ("disk", "spherical"):
            if stem.endswith("_" + dom):
                return stem[: -len(dom) - 1], dom
        raise ValueError(f"cannot parse weight-set name {stem!r}")

    @classmethod
    def all_measured(cls, precision: str = "default") -> "MaterialTable":
        """27 disk + 25 spherical measured materials (config 4)."""
        stems = W.list_shipped("disk") + [s for s in W.list_shipped("spherical")
                                          if not s.startswith("bsdf_") and not s.endswith("_complex")]
        return cls([s for s in stems if not s.endswith("_complex")], precision)

    def __len__(self):
        return len(self.samplers)

    def _groups(self):
        """Indices of materials grouped by kernel signature + plugin defaults (T, variant)."""
        groups = {}
        for m, s in enumerate(self.samplers):
            fw = s.weights
            key = (fw.domain, fw.width, fw.n_hidden, s.precision, self.T[m], self.variant[m])
            groups.setdefault(key, []).append(m)
        return groups

    def _multi(self, which, members, seg_end_all, T, variant, wi_s, aux_s, seed, offset, out_wo, out_pdf, ctx=None,
               gkey=None, rng_rows=None, ctx_fill=None, direct_rows=None):
        """``direct_rows`` (int64 [n_mat], the bucket permutation): the arrays are the CALLERS' lane-ordered ones and every launch
        reads and writes its rows through ``bsdfd_opts.row_index`` — no gathered copy, no scatter (``direct=True`` of the public
        calls).  The Philox counter is then ``offset + original lane`` by construction.

        ``ctx`` (a dict, or None): per-query contexts (include/bsdfd.h, bsdfd_context_bytes) of the runs of this
        wavefront — the FILLING call (``ctx_fill``; default: "sample" fills, "pdf" reads) writes one buffer per (kernel
        signature, run), the other call on the same bucketed ``wi`` reads them instead of recomputing the per-query prologue.
        The dict keeps ONE buffer per (kernel signature, run number), grown to the largest wavefront seen, together with the
        bucket layout it was last filled for; a reading call whose layout differs is refused.

        Segmented launch(es) for the materials `members` of one kernel signature.  The bucketed arrays
        are ordered by material id, so a group's buckets may be interleaved with other groups'; each
        maximal run of ADJACENT buckets becomes one `bsdfd_plugin_*_multi` call on the run's row range
        (pointers advanced to the run's first row, Philox offset advanced by the same amount)."""
        L = _lib.lib()
        rc = 0
        stream = C.c_void_p(torch.cuda.current_stream(wi_s.device).cuda_stream)
        # the C ABI takes cumulative ends; non-adjacent buckets are issued as separate runs
        run_h, run_end, base = [], [], None
        run_no = 0
        fill = (which == "sample") if ctx_fill is None else bool(ctx_fill)
        def flush():
            nonlocal run_h, run_end, base, run_no
            if not run_h:
                return 0
            k = len(run_h)
            arr_h = (C.c_void_p * k)(*run_h)
            arr_e = (C.c_int64 * k)(*[e - base for e in run_end])
            off_rows = base
            wi_p = C.c_void_p(wi_s.data_ptr() + off_rows * 12)
            cbuf = None
            if ctx is not None and which in ("sample", "pdf"):
                ckey, layout = (gkey, run_no), (base, tuple(run_end))
                run_no += 1
                ent = ctx.get(ckey)
                if fill:
                    nbytes = int(L.bsdfd_context_bytes(run_h[0], run_end[-1] - base, k))
                    if ent is None or ent["buf"].numel() * 4 < nbytes or ent["buf"].device != wi_s.device:
                        ent = ctx[ckey] = {"buf": torch.empty((nbytes // 4,), dtype=torch.float32, device=wi_s.device)}
                    ent["layout"] = layout
                elif ent is None or ent.get("layout") != layout:
                    raise ValueError("a context-reading call needs the context a filling call of the SAME bucketed wavefront "
                                     "(same plan, same wi array) wrote: sample(ctx=) before pdf(ctx=), or pdf(ctx=, "
                                     "ctx_fill=True) before sample(ctx=, ctx_fill=False)")
                cbuf = ent["buf"]
            c_out, c_in = (cbuf, None) if fill else (None, cbuf)
            if direct_rows is not None:
                # the row range of this run lives in the INDEX array: data pointers stay at row 0 of the lane-ordered arrays
                o = _lib.opts(ctx_out=c_out, ctx_in=c_in, row_index=direct_rows, byte_offset_row=off_rows * 8)
                full = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
                if which == "sample":
                    r = L.bsdfd_plugin_sample_multi_ex(arr_h, k, arr_e, variant, full(wi_s), full(aux_s), seed, offset, T,
                                                       full(out_wo), full(out_pdf), C.byref(o), stream)
                elif which == "pdf":
                    r = L.bsdfd_plugin_pdf_multi_ex(arr_h, k, arr_e, variant, full(wi_s), full(aux_s), T, full(out_pdf),
                                                    C.byref(o), stream)
                else:   # sample_pdf: aux_s = (x0 or None, wl), out_pdf = (pdf_wo, pdf_wl)
                    r = L.bsdfd_plugin_sample_pdf_multi_ex(arr_h, k, arr_e, variant, full(wi_s), full(aux_s[0]), full(aux_s[1]),
                                                           seed, offset, T, full(out_wo), full(out_pdf[0]), full(out_pdf[1]),
                                                           C.byref(o), stream)
            elif which == "sample" and (cbuf is not None or rng_rows is not None):
                x0_p = None if aux_s is None else C.c_void_p(aux_s.data_ptr() + off_rows * 8)
                # rng_rows: counter of a row = offset + its ORIGINAL lane index (not its bucketed position)
                o = _lib.opts(ctx_out=c_out, ctx_in=c_in, rng_index=rng_rows, byte_offset_rng=off_rows * 8)
                r = L.bsdfd_plugin_sample_multi_ex(arr_h, k, arr_e, variant, wi_p, x0_p, seed,
                                                   offset if rng_rows is not None else offset + off_rows, T,
                                                   C.c_void_p(out_wo.data_ptr() + off_rows * 12),
                                                   C.c_void_p(out_pdf.data_ptr() + off_rows * 4), C.byref(o), stream)
            elif which == "pdf" and cbuf is not None:
                o = _lib.opts(ctx_out=c_out, ctx_in=c_in)
                r = L.bsdfd_plugin_pdf_multi_ex(arr_h, k, arr_e, variant, wi_p,
                                                C.c_void_p(aux_s.data_ptr() + off_rows * 12), T,
                                                C.c_void_p(out_pdf.data_ptr() + off_rows * 4), C.byref(o), stream)
            elif which == "sample":
                x0_p = None if aux_s is None else C.c_void_p(aux_s.data_ptr() + off_rows * 8)
                r = L.bsdfd_plugin_sample_multi(arr_h, k, arr_e, variant, wi_p, x0_p, seed, offset + off_rows, T,
                                                C.c_void_p(out_wo.data_ptr() + off_rows * 12),
                                                C.c_void_p(out_pdf.data_ptr() + off_rows * 4), stream)
            elif which == "sample_pdf":  # aux_s = (x0 or None, wl), out_pdf = (pdf_wo, pdf_wl)
                x0_s, wl_s = aux_s
                x0_p = None if x0_s is None else C.c_void_p(x0_s.data_ptr() + off_rows * 8)
                o = _lib.opts(rng_index=rng_rows, byte_offset_rng=off_rows * 8)
                r = L.bsdfd_plugin_sample_pdf_multi_ex(arr_h, k, arr_e, variant, wi_p, x0_p,
                                                       C.c_void_p(wl_s.data_ptr() + off_rows * 12), seed,
                                                       offset if rng_rows is not None else offset + off_rows, T,
                                                       C.c_void_p(out_wo.data_ptr() + off_rows * 12),
                                                       C.c_void_p(out_pdf[0].data_ptr() + off_rows * 4),
                                                       C.c_void_p(out_pdf[1].data_ptr() + off_rows * 4), C.byref(o), stream)
            else:
                r = L.bsdfd_plugin_pdf_multi(arr_h, k, arr_e, variant, wi_p,
                                             C.c_void_p(aux_s.data_ptr() + off_rows * 12), T,
                                             C.c_void_p(out_pdf.data_ptr() + off_rows * 4), stream)
            run_h, run_end, base = [], [], None
            return r
        prev_end = None
        for m in members:
            b, e = seg_end_all[m - 1] if m else 0, seg_end_all[m]
            if prev_end is not None and b != prev_end:
                rc = rc or flush()
            if base is None:
                base = b
            run_h.append(self.samplers[m]._h)
            run_end.append(e)
            prev_end = e
        rc = rc or flush()
        _lib.check(rc)

    def _buckets(self, material_id, extra_bins: int = 0):
        if isinstance(material_id, tuple):  # a plan from bucket(): (perm, counts)
            perm, counts = material_id
            if (not isinstance(perm, torch.Tensor) or perm.dtype != torch.int64 or perm.dim() != 1
                    or len(counts) < len(self) or sum(counts) != perm.shape[0] or min(counts, default=0) < 0):
                raise ValueError("not a bucketing plan of this table: expected (perm int64 [N], counts) from bucket()")
            return perm, list(counts)
        if material_id.dtype != torch.int64:
            material_id = material_id.long()
        perm, counts = bucket_by_material(material_id, len(self) + extra_bins)
        counts = counts.cpu().tolist()
        if sum(counts) != material_id.shape[0]:
            raise ValueError(f"material ids must be in [0, {len(self) + extra_bins})")
        return perm, counts

    def _plan(self, material_id, n_rows: int):
        """(rows, counts, seg_end): `rows` = the bucketed order of the lanes that carry a material (lanes of the
        plan's extra bins sort behind them and are not evaluated), `counts` per material."""
        perm, counts = self._buckets(material_id)
        if perm.shape[0] != n_rows:
            raise ValueError(f"the bucketing plan covers {perm.shape[0]} rows, the batch has {n_rows}")
        counts = counts[: len(self)]
        n_mat = sum(counts)
        rows = perm if n_mat == n_rows else perm[:n_mat]
        return rows, counts, list(__import__("itertools").accumulate(counts))

    def _chk_in(self, t, cols, name, n=None):
        return self.samplers[0]._chk(t, cols, name, n)

    # -- bucketed flow: a renderer keeps a wavefront in bucket order across its sample() and pdf() calls and scatters
    # once at the end, instead of gathering the inputs and scattering the outputs in every call -----------------------
    def _plan_bucketed(self, plan):
        if not isinstance(plan, tuple):
            raise ValueError("bucketed=True needs a plan from bucket()")
        perm, counts = self._buckets(plan)
        counts = counts[: len(self)]
        n_mat = sum(counts)
        return perm[:n_mat] if n_mat != perm.shape[0] else perm, counts, list(__import__("itertools").accumulate(counts))

    def gather(self, plan, *tensors):
        """Rows of the lanes that carry a material, in bucket order (what ``bucketed=True`` calls take and return).
        A single fp32 [N,3] array (the wavefront's ``wi``) goes through the native one-pass kernel."""
        rows, _, _ = self._plan_bucketed(plan)
        if (len(tensors) == 1 and tensors[0].is_cuda and tensors[0].dtype == torch.float32 and tensors[0].dim() == 2
                and tensors[0].shape[1] == 3 and tensors[0].is_contiguous() and rows.is_contiguous()):
            wi = tensors[0]
            out = torch.empty((rows.shape[0], 3), dtype=torch.float32, device=wi.device)
            with torch.cuda.device(wi.device):
                _lib.check(_lib.lib().bsdfd_gather_lanes(C.c_void_p(rows.data_ptr()), rows.shape[0], C.c_void_p(wi.data_ptr()),
                                                         C.c_void_p(out.data_ptr()),
                                                         C.c_void_p(torch.cuda.current_stream(wi.device).cuda_stream)))
            return out
        out = tuple(t[rows].contiguous() for t in tensors)
        return out[0] if len(out) == 1 else out

    def scatter(self, plan, *tensors):
        """Inverse of ``gather``: bucket-ordered results back to the callers' lane order (lanes without a material: 0).
        The usual triple of a wavefront — (wo [n,3], pdf [n], pdf [n]) or (wo, pdf) — is written in ONE native pass."""
        perm, _ = plan
        rows, _, _ = self._plan_bucketed(plan)
        n, k = perm.shape[0], rows.shape[0]
        mk = torch.empty if k == n else torch.zeros

        def f32c(t, *shape):
            return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape
        if (len(tensors) in (2, 3) and f32c(tensors[0], k, 3) and all(f32c(t, k) for t in tensors[1:]) and rows.is_contiguous()):
            dev = tensors[0].device
            full = [mk((n, 3), dtype=torch.float32, device=dev)] + [mk((n,), dtype=torch.float32, device=dev) for _ in tensors[1:]]
            p = [C.c_void_p(t.data_ptr()) for t in tensors] + [None] * (3 - len(tensors))
            q = [C.c_void_p(t.data_ptr()) for t in full] + [None] * (3 - len(tensors))
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().bsdfd_scatter_lanes(C.c_void_p(rows.data_ptr()), k, p[0], p[1], p[2], q[0], q[1], q[2],
                                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            return tuple(full)
        out = []
        for t in tensors:
            full = mk((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            full[rows] = t
            out.append(full)
        return out[0] if len(out) == 1 else tuple(out)

    def bucket(self, material_id: torch.Tensor, extra_bins: int = 0):
        """Bucket a wavefront once and reuse the plan for its sample() and pdf() calls (a renderer asks both
        for the same intersections): pass the returned value in place of ``material_id``.  The stable sort
        of 16 Mi ids costs 0.2 ms natively (1.7 ms with torch.argsort).  ``extra_bins``: ids len(self) ..
        len(self)+extra_bins-1 are lanes that carry no material (a renderer's floor hits and misses); they
        sort behind the materials, are not evaluated, and their outputs are zero."""
        return self._buckets(material_id, extra_bins)

    def sample(self, material_id: torch.Tensor, wi: torch.Tensor, seed: int = 0, offset: int = 0,
               T: Optional[int] = None, x0: Optional[torch.Tensor] = None, segmented: bool = True,
               bucketed: bool = False, ctx: Optional[dict] = None, rng: str = "lane", ctx_fill: bool = True,
               direct: bool = False):
        """wi [N,3], material_id [N] -> (wo [N,3], pdf_sa [N]) in the callers' order.
        ``rng``: ``"lane"`` (default) — the Philox counter of a query is ``offset + its ORIGINAL lane index``, so the base
        draws depend on neither the bucketing nor how the wavefront is sharded over calls / GPUs (a shard passes its
        first lane's global index as ``offset``); ``"bucketed"`` — ``offset + row in the bucketed
        array`` (what a caller that keeps no lane order would use).
        ``ctx``: a dict this call fills with the wavefront's per-query contexts; hand the same dict (and the same
        plan and ``wi``) to ``pdf(..., ctx=)`` and it skips the per-query prologue (identical results).  The other order
        works too: ``pdf(..., ctx=d, ctx_fill=True)`` first, then ``sample(..., ctx=d, ctx_fill=False)``.  Segmented path only.
        Identical for the segmented and the per-bucket path.  ``bucketed=True``: ``material_id`` is a
        plan, ``wi`` / ``x0`` are already in bucket order (``gather(plan, wi)``) and the results stay in it.
        ``direct=True``: no gathered copies at all — the launches read ``wi`` / ``x0`` and write the results in the callers' lane
        order THROUGH the bucket permutation (``bsdfd_opts.row_index``); same results bit for bit (``rng="lane"`` only)."""
        wi = self._chk_in(wi, 3, "wi")
        x0 = self._chk_in(x0, 2, "x0", wi.shape[0])
        if ctx is not None and not segmented:
            raise ValueError("ctx= is a feature of the segmented path (segmented=False issues one plain call per bucket)")
        if direct:
            if bucketed or rng != "lane":
                raise ValueError("direct=True takes lane-ordered arrays and draws with rng='lane'")
            rows, counts, seg_end = self._plan(material_id, wi.shape[0])
            rows = rows.contiguous()
            mk = torch.empty if rows.shape[0] == wi.shape[0] else torch.zeros  # lanes without a material: zeros
            wo = mk(wi.shape, dtype=torch.float32, device=wi.device)
            pdf = mk(wi.shape[0], dtype=torch.float32, device=wi.device)
            with torch.cuda.device(wi.device):
                if segmented:
                    for (dom, w, nh, prec, Tm, var), members in self._groups().items():
                        self._multi("sample", members, seg_end, Tm if T is None else T, var, wi, x0, seed, offset, wo, pdf,
                                    ctx=ctx, gkey=(dom, w, nh, prec, var), ctx_fill=ctx_fill, direct_rows=rows)
                else:
                    lo = 0
                    for m, n in enumerate(counts):
                        if n:
                            self.samplers[m].plugin_sample(wi, x0, T=self.T[m] if T is None else T, variant=self.variant[m],
                                                           seed=seed, offset=offset, out=(wo, pdf), row_index=rows[lo:lo + n])
                        lo += n
            return wo, pdf
        if bucketed:
            rows, counts, seg_end = self._plan_bucketed(material_id)
            if wi.shape[0] != rows.shape[0]:
                raise ValueError(f"bucketed wi has {wi.shape[0]} rows, the plan has {rows.shape[0]} material lanes")
            wi_s, x0_s = wi, x0
        else:
            rows, counts, seg_end = self._plan(material_id, wi.shape[0])
            wi_s = wi[rows].contiguous()
            x0_s = None if x0 is None else x0[rows].contiguous()
        wo_s = torch.empty_like(wi_s)
        pdf_s = torch.empty(wi_s.shape[0], dtype=torch.float32, device=wi.device)
        if rng not in ("lane", "bucketed"):
            raise ValueError("rng must be 'lane' or 'bucketed'")
        rng_rows = rows.contiguous() if rng == "lane" else None
        if segmented:
            with torch.cuda.device(wi.device):
                for (dom, w, nh, prec, Tm, var), members in self._groups().items():
                    self._multi("sample", members, seg_end, Tm if T is None else T, var, wi_s, x0_s, seed, offset,
                                wo_s, pdf_s, ctx=ctx, gkey=(dom, w, nh, prec, var), rng_rows=rng_rows, ctx_fill=ctx_fill)
        else:
            lo = 0
            for m, n in enumerate(counts):
                if n == 0:
                    continue
                sl = slice(lo, lo + n)
                # same Philox keying as the segmented path
                self.samplers[m].plugin_sample(wi_s[sl], None if x0_s is None else x0_s[sl],
                                               T=self.T[m] if T is None else T, variant=self.variant[m],
                                               seed=seed, offset=offset if rng_rows is not None else offset + lo,
                                               out=(wo_s[sl], pdf_s[sl]),
                                               rng_index=None if rng_rows is None else rng_rows[sl])
                lo += n
        if bucketed:
            return wo_s, pdf_s
        mk = torch.empty if rows.shape[0] == wi.shape[0] else torch.zeros  # lanes without a material: zeros
        wo = mk(wi.shape, dtype=torch.float32, device=wi.device)
        pdf = mk(wi.shape[0], dtype=torch.float32, device=wi.device)
        wo[rows] = wo_s
        pdf[rows] = pdf_s
        return wo, pdf

    def sample_pdf(self, material_id, wi: torch.Tensor, wl: torch.Tensor, seed: int = 0, offset: int = 0,
                   T: Optional[int] = None, x0: Optional[torch.Tensor] = None, return_bucketed: bool = False,
                   rng: str = "lane", direct: bool = False):
        """sample(wi) and pdf(wi, wl) for the same material-tagged intersections, one launch per kernel
        signature (``bsdfd_plugin_sample_pdf_multi``) -> (wo [N,3], pdf(wo) [N], pdf(wl) [N]) in the callers' order."""
        wi = self._chk_in(wi, 3, "wi")
        wl = self._chk_in(wl, 3, "wl", wi.shape[0])
        x0 = self._chk_in(x0, 2, "x0", wi.shape[0])
        rows, counts, seg_end = self._plan(material_id, wi.shape[0])  # lanes of the extra bins carry no material
        n_mat = rows.shape[0]
        if direct:   # through the bucket permutation: no gathered copies, results land in lane order
            if return_bucketed or rng != "lane":
                raise ValueError("direct=True returns lane-ordered arrays only and draws with rng='lane'")
            mk = torch.empty if n_mat == wi.shape[0] else torch.zeros
            wo = mk(wi.shape, dtype=torch.float32, device=wi.device)
            po = mk(wi.shape[0], dtype=torch.float32, device=wi.device)
            pl = mk(wi.shape[0], dtype=torch.float32, device=wi.device)
            with torch.cuda.device(wi.device):
                for (dom, w, nh, prec, Tm, var), members in self._groups().items():
                    self._multi("sample_pdf", members, seg_end, Tm if T is None else T, var, wi, (x0, wl), seed, offset,
                                wo, (po, pl), direct_rows=rows.contiguous())
            return wo, po, pl
        wi_s, wl_s = wi[rows].contiguous(), wl[rows].contiguous()
        x0_s = None if x0 is None else x0[rows].contiguous()
        wo_s = torch.empty_like(wi_s)
        po_s = torch.empty(n_mat, dtype=torch.float32, device=wi.device)
        pl_s = torch.empty_like(po_s)
        with torch.cuda.device(wi.device):
            for (dom, w, nh, prec, Tm, var), members in self._groups().items():
                self._multi("sample_pdf", members, seg_end, Tm if T is None else T, var, wi_s, (x0_s, wl_s), seed,
                            offset, wo_s, (po_s, pl_s), rng_rows=rows.contiguous() if rng == "lane" else None)
        full = n_mat == wi.shape[0]
        mk = torch.empty if full else torch.zeros
        wo = mk(wi.shape, dtype=torch.float32, device=wi.device)
        po = mk(wi.shape[0], dtype=torch.float32, device=wi.device)
        pl = mk(wi.shape[0], dtype=torch.float32, device=wi.device)
        wo[rows] = wo_s
        po[rows] = po_s
        pl[rows] = pl_s
        if return_bucketed:  # the bucket-ordered arrays too (material m = rows seg_end[m-1] .. seg_end[m]) and `rows`
            return wo, po, pl, dict(wi=wi_s, wl=wl_s, wo=wo_s, pdf_o=po_s, pdf_l=pl_s, rows=rows, seg_end=seg_end)
        return wo, po, pl

    def pdf(self, material_id: torch.Tensor, wi: torch.Tensor, wo: torch.Tensor, T: Optional[int] = None,
            segmented: bool = True, bucketed: bool = False, ctx: Optional[dict] = None, ctx_fill: bool = False,
            direct: bool = False):
        wi = self._chk_in(wi, 3, "wi")
        wo = self._chk_in(wo, 3, "wo", wi.shape[0])
        if ctx is not None and not segmented:
            raise ValueError("ctx= is a feature of the segmented path (segmented=False issues one plain call per bucket)")
        if direct:   # lane-ordered arrays, read and written through the bucket permutation (see sample())
            if bucketed:
                raise ValueError("direct=True takes lane-ordered arrays")
            rows, counts, seg_end = self._plan(material_id, wi.shape[0])
            rows = rows.contiguous()
            pdf = (torch.empty if rows.shape[0] == wi.shape[0] else torch.zeros)(wi.shape[0], dtype=torch.float32, device=wi.device)
            with torch.cuda.device(wi.device):
                if segmented:
                    for (dom, w, nh, prec, Tm, var), members in self._groups().items():
                        self._multi("pdf", members, seg_end, Tm if T is None else T, var, wi, wo, 0, 0, None, pdf,
                                    ctx=ctx, gkey=(dom, w, nh, prec, var), ctx_fill=ctx_fill, direct_rows=rows)
                else:
                    lo = 0
                    for m, n in enumerate(counts):
                        if n:
                            self.samplers[m].plugin_pdf(wi, wo, T=self.T[m] if T is None else T, variant=self.variant[m],
                                                        out=pdf, row_index=rows[lo:lo + n])
                        lo += n
            return pdf
        if bucketed:
            rows, counts, seg_end = self._plan_bucketed(material_id)
            if wi.shape[0] != rows.shape[0]:
                raise ValueError(f"bucketed wi has {wi.shape[0]} rows, the plan has {rows.shape[0]} material lanes")
            wi_s, wo_s = wi, wo
        else:
            rows, counts, seg_end = self._plan(material_id, wi.shape[0])
            wi_s, wo_s = wi[rows].contiguous(), wo[rows].contiguous()
        pdf_s = torch.empty(wi_s.shape[0], dtype=torch.float32, device=wi.device)
        if segmented:
            with torch.cuda.device(wi.device):
                for (dom, w, nh, prec, Tm, var), members in self._groups().items():
                    self._multi("pdf", members, seg_end, Tm if T is None else T, var, wi_s, wo_s, 0, 0, None, pdf_s,
                                ctx=ctx, gkey=(dom, w, nh, prec, var), ctx_fill=ctx_fill)
        else:
            lo = 0
            for m, n in enumerate(counts):
                if n == 0:
                    continue
                sl = slice(lo, lo + n)
                self.samplers[m].plugin_pdf(wi_s[sl], wo_s[sl], T=self.T[m] if T is None else T,
                                            variant=self.variant[m], out=pdf_s[sl])
                lo += n
        if bucketed:
            return pdf_s
        pdf = (torch.empty if rows.shape[0] == wi.shape[0] else torch.zeros)(wi.shape[0], dtype=torch.float32,
                                                                             device=wi.device)
        pdf[rows] = pdf_s
        return pdf


class _Wavefront:
    """One wavefront in flight through a ``WavefrontPipeline``: ``result()`` orders the calling stream behind its scatter
    and returns (wo [N,3], pdf(wo) by sample() [N], pdf(wo) by pdf() [N]) in the callers' lane order."""

    def __init__(self):
        self.plan = self.wi_b = self.bucketed = self.out = None
        self.prep_done = self.flow_done = self.scatter_done = None

    def result(self):
        cur = torch.cuda.current_stream(self.out[0].device)
        cur.wait_event(self.scatter_done)
        for t in self.out:
            t.record_stream(cur)
        return self.out


class WavefrontPipeline:
    """sample() + pdf() of a STREAM of independent material-tagged wavefronts (the passes of a render, the tiles of a film),
    software-pipelined over three HIP streams: the bucketing + gather of wavefront k run on a side stream under the flow
    kernels of wavefront k-1 (the host's wait for the bucket counts included), its scatter on another under the flow
    kernels of wavefront k+1.  The flow kernels — the only compute-bound part — then run back to back on the calling
    stream; the streaming passes around them use the HBM bandwidth the flow kernels leave idle.  Results are those of
    ``bucket -> gather -> sample(bucketed) -> pdf(bucketed) -> scatter`` issued one after the other, bit for bit.

    ``push()`` returns a ``_Wavefront``; at most two are in flight: the buffers of wavefront k are reused by wavefront k+2,
    so take ``result()`` of a wavefront before pushing the second one after it."""

    # Largest wavefront (lanes) served through the bucket permutation when ``direct`` is left to the pipeline.  Measured, 52 materials,
    # steady state, alternating (profiles/r06_ab/mixed_row_index_vs_gather_by_size.jsonl): wall time of the direct form over the
    # gather form 0.92 (256 Ki lanes), 0.97 (1 Mi), 1.00 (2 Mi), 0.93 (4 Mi), 0.95 (8 Mi), 1.05 (16 Mi).  Why it turns: a row-indexed
    # 12-byte access touches a whole line that ~10 tiles of OTHER buckets touch at other times; while the lane-ordered arrays sit in
    # the 256 MiB Infinity Cache that is free, beyond it every touch is an HBM access — at 16 Mi lanes the direct form moves 8.7 GB
    # per wavefront on the HBM side against 6.1 GB (1.07 GB algorithmic; rocprofv3 FETCH_SIZE / WRITE_SIZE,
    # profiles/r06_ab/mixed_16Mi_row_index_vs_gather_pmc.json): it reads wi through the index twice and writes AND re-reads wo through
    # it, the gather form pays each amplified pass once.
    DIRECT_MAX_LANES = 8 << 20

    def __init__(self, table: MaterialTable, direct: Optional[bool] = None):
        """``direct``: the flow kernels read ``wi`` and write the results in lane order through the bucket permutation
        (``bsdfd_opts.row_index``, round 6) — only the bucketing itself (count, scan, permutation) is left on the side stream, the
        gather of ``wi`` and the scatter of (wo, pdf, pdf) are gone.  ``False``: the round-4 form (gather on the side stream,
        bucket-ordered launches, scatter on a third stream).  ``None`` (default): by wavefront size — direct up to
        ``DIRECT_MAX_LANES`` lanes, the gather form beyond.  Identical results either way, bit for bit."""
        self.tab = table
        self.direct = direct
        self.pre = self.post = None
        self.slots = [_Wavefront(), _Wavefront()]
        self.k = 0

    def push(self, material_id: torch.Tensor, wi: torch.Tensor, seed: int = 0, offset: int = 0,
             ctx: Optional[dict] = None, extra_bins: int = 0, ready=None) -> _Wavefront:
        """``material_id`` / ``wi`` are read on a side stream.  ``ready`` orders that stream behind their producer:

        * ``None`` (default, always safe): an event recorded NOW on the calling stream — whatever the caller enqueued there to
          produce the inputs is waited for.  It is also ordered behind the previous wavefronts' flow kernels on that stream,
          so a caller that produces its inputs on the calling stream gets no overlap of the bucketing with them;
        * a ``torch.cuda.Event`` the producer recorded right behind its last write (e.g. on its own stream): full overlap;
        * ``False``: the inputs are complete already (resident arrays, a host-synchronised producer): no wait at all."""
        tab, dev = self.tab, wi.device
        direct = self.direct if self.direct is not None else wi.shape[0] <= self.DIRECT_MAX_LANES
        main = torch.cuda.current_stream(dev)
        if self.pre is None or self.pre.device != dev:
            self.pre, self.post = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        w = self.slots[self.k % 2]
        self.k += 1
        if ready is None:
            ready = main.record_event()
        with torch.cuda.stream(self.pre):
            if ready is not False:
                self.pre.wait_event(ready)
            if w.flow_done is not None:
                self.pre.wait_event(w.flow_done)       # the flow kernels of wavefront k-2 have read the buffers reused here
            plan = tab.bucket(material_id, extra_bins)  # (the host waits for the counts on THIS stream only)
            wi_b = None if direct else tab.gather(plan, wi)
            for t in (plan[0], wi_b):
                if t is not None:
                    t.record_stream(main)
            w.plan, w.wi_b = plan, wi_b
            w.prep_done = self.pre.record_event()
        main.wait_event(w.prep_done)
        if direct:
            if w.scatter_done is not None:
                main.wait_event(w.scatter_done)   # (a slot last used by the gather form: its scatter reads the slot's buffers)
            wo, pdf = tab.sample(plan, wi, seed=seed, offset=offset, ctx=ctx, direct=True)
            p2 = tab.pdf(plan, wi, wo, ctx=ctx, direct=True)
            w.bucketed = None
            w.out = (wo, pdf, p2)
            w.flow_done = w.scatter_done = main.record_event()
            return w
        if w.scatter_done is not None:
            main.wait_event(w.scatter_done)
        wo_b, pdf_b = tab.sample(plan, wi_b, seed=seed, offset=offset, bucketed=True, ctx=ctx)
        p_b = tab.pdf(plan, wi_b, wo_b, bucketed=True, ctx=ctx)
        for t in (wo_b, pdf_b, p_b, plan[0]):
            t.record_stream(self.post)
        w.bucketed = (wo_b, pdf_b, p_b)
        w.flow_done = main.record_event()
        with torch.cuda.stream(self.post):
            self.post.wait_event(w.flow_done)
            w.out = tab.scatter(plan, wo_b, pdf_b, p_b)
            w.scatter_done = self.post.record_event()
        return w
