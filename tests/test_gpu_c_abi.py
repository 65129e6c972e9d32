"""The C ABI used from plain C++ (no Python host, no torch): build examples/c_abi_demo.cpp against
libbsdfd.so, run it on the GPU (eager + hipGraph capture/replay), and check concurrent streams."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_demo_builds_and_runs(tmp_path):
    from bsdf_diffusion_sampling_amd import _lib, weights as W
    _lib.build()
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_demo.cpp"), "-L", libdir, "-lbsdfd",
                    f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    for mat, dom in (("aniso_miro_7_rgb", "disk"), ("chm_orange_rgb", "spherical")):
        r = subprocess.run([exe, W.shipped_path(mat, dom), "200000"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "graph : 0 of 200000 pdf values differ" in r.stdout, r.stdout
        assert "of per-query context; 0 of 200000 pdf values differ" in r.stdout, r.stdout   # bsdfd_opts / *_ex from plain C++
        assert "row_index; 0 pdf values differ from the plain run, 0 untagged lanes touched" in r.stdout, r.stdout   # ABI 6
    r = subprocess.run([exe, "/nonexistent.bsdfw"], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot open" in r.stderr


def test_concurrent_streams_share_one_handle():
    from conftest import load_case
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    g, fw = load_case("chm_orange_rgb_disk")
    dev = torch.device("cuda", 0)
    s = FlowSampler(fw)
    n = 1 << 18
    wi = torch.from_numpy(np.tile(g["wi"], (n // 2048, 1))).to(dev)
    x0 = torch.from_numpy(np.tile(g["x0"], (n // 2048, 1))).to(dev)
    ref_x, ref_p = s.network_sampling(wi, x0, T=8)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [None] * 4
    torch.cuda.synchronize()
    for rep in range(3):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = s.network_sampling(wi, x0, T=8)
    torch.cuda.synchronize()
    for x, p in outs:
        assert torch.equal(x, ref_x) and torch.equal(p, ref_p)


def test_profile_totals_by_kind_of_launch():
    """bsdfd_profile_read_op: the event totals of a profiling handle split by kind of launch add up to bsdfd_profile_read's."""
    from conftest import load_case
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    g, fw = load_case("aniso_miro_7_rgb_disk")
    dev = torch.device("cuda", 0)
    s = FlowSampler(fw)
    v = np.random.default_rng(5).normal(size=(1 << 16, 3)).astype(np.float32)
    v[:, 2] = np.abs(v[:, 2])
    wi = torch.from_numpy(v / np.linalg.norm(v, axis=1, keepdims=True)).to(dev)
    wo, p = s.plugin_sample(wi, None, T=4, seed=1)
    s.set_profiling(True)
    for k in range(3):
        s.plugin_sample(wi, None, T=4, seed=k)
    for k in range(5):
        s.plugin_pdf(wi, wo, T=8)
    s.plugin_sample_pdf(wi, wo, None, T=4, seed=9)
    n, ms = s.profile_read()
    kinds = {k: s.profile_read_op(k) for k in ("sample", "pdf", "samples_only", "sample_pdf")}
    s.set_profiling(False)
    assert n == 9 and [kinds[k][0] for k in ("sample", "pdf", "samples_only", "sample_pdf")] == [3, 5, 0, 1]
    assert abs(sum(v[1] for v in kinds.values()) - ms) < 1e-6 * max(ms, 1.0)
    assert kinds["pdf"][1] / 5 > kinds["sample"][1] / 3 > 0          # T = 8 launches last longer than T = 4 ones
    assert s.profile_read_op("sample") == (0, 0.0)                    # reset by set_profiling
    with pytest.raises(KeyError):
        s.profile_read_op("warp")


def test_calls_are_hip_graph_capturable():
    """Small wavefronts are launch-bound; the entry points do no allocation, no synchronisation and no
    host read-back, so sample() + pdf() can be captured into a HIP graph and replayed."""
    from conftest import load_case
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    g, fw = load_case("aniso_miro_7_rgb_disk")
    dev = torch.device("cuda", 0)
    s = FlowSampler(fw)
    n = 4096
    gen = torch.Generator().manual_seed(5)
    u = torch.rand(n, 2, generator=gen)
    r, a = 0.9 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    wi2 = torch.stack([r * torch.cos(a), r * torch.sin(a)], 1)
    wi = torch.cat([wi2, torch.sqrt(1 - (wi2 ** 2).sum(1, keepdim=True))], 1).float().to(dev)
    wo = torch.empty_like(wi)
    pdf_s = torch.empty(n, device=dev)
    pdf_p = torch.empty(n, device=dev)
    # eager reference
    s.plugin_sample(wi, None, T=4, seed=9, offset=0, out=(wo, pdf_s))
    s.plugin_pdf(wi, wo, T=4, out=pdf_p)
    torch.cuda.synchronize()
    ref = (wo.clone(), pdf_s.clone(), pdf_p.clone())
    wo.zero_(); pdf_s.zero_(); pdf_p.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            s.plugin_sample(wi, None, T=4, seed=9, offset=0, out=(wo, pdf_s))
            s.plugin_pdf(wi, wo, T=4, out=pdf_p)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        wo.zero_(); pdf_s.zero_(); pdf_p.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(wo, ref[0]) and torch.equal(pdf_s, ref[1]) and torch.equal(pdf_p, ref[2])
    # a new wavefront through the same graph: inputs are read at replay time
    wi.copy_(wi.flip(0))
    graph.replay()
    torch.cuda.synchronize()
    s.plugin_pdf(wi, wo, T=4, out=pdf_s)  # eager pdf of the replayed directions
    torch.cuda.synchronize()
    assert torch.equal(pdf_s, pdf_p)


def test_create_destroy_does_not_leak_device_memory():
    """Handles own their packed weights / tables (hipMalloc) and a ring of HIP events: 300 create-use-destroy
    cycles must give the memory back."""
    import ctypes as C
    import gc
    from conftest import load_case
    from bsdf_diffusion_sampling_amd import _lib
    from bsdf_diffusion_sampling_amd.measured import MeasuredBSDF
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    g, fw = load_case("aniso_miro_7_rgb_spherical_complex")       # the largest weight image (~90 KB)
    dev = torch.device("cuda", 0)
    wi = torch.from_numpy(g["wi"][:256].astype(np.float32)).to(dev)
    fixture = os.path.join(ROOT, "tests", "golden", "chm_orange_rgb.bsdf")

    def cycle(n):
        for _ in range(n):
            s = FlowSampler(fw)
            s.network_sampling(wi, None, T=2)
            m = MeasuredBSDF(fixture)                                  # ~1.3 MB of tables
            del s, m
        gc.collect()
        torch.cuda.synchronize()
    cycle(20)
    free0, _ = torch.cuda.mem_get_info()
    cycle(300)
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)                     # 300 leaked table sets would be ~400 MB


def test_render_demo_in_plain_cpp(tmp_path):
    """examples/render_demo.cpp: a whole render loop (primary -> fused sample+pdf -> ground-truth eval -> shade)
    on the C ABI alone."""
    from bsdf_diffusion_sampling_amd import weights as W
    exe = str(tmp_path / "render_demo")
    lib_dir = os.path.join(ROOT, "bsdf_diffusion_sampling_amd")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "render_demo.cpp"), "-L", lib_dir, "-lbsdfd",
                        f"-Wl,-rpath,{lib_dir}", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    weights = W.shipped_path("chm_orange_rgb", "disk")
    gt = os.path.join(ROOT, "tests", "golden", "chm_orange_rgb.bsdf")
    means = {}
    for tag, arg in (("proxy", "-"), ("gt", gt)):
        out = str(tmp_path / f"{tag}.ppm")
        r = subprocess.run([exe, weights, arg, out, "128", "16"], capture_output=True, text=True)
        assert r.returncode == 0 and "finite 1" in r.stdout, r.stdout + r.stderr
        raw = open(out, "rb").read()
        assert raw.startswith(b"P6\n128 128\n255\n") and len(raw) == 15 + 3 * 128 * 128
        px = np.frombuffer(raw[15:], dtype=np.uint8).reshape(128, 128, 3).astype(np.float64)
        means[tag] = px.mean((0, 1))
    assert means["gt"][0] > 1.15 * means["gt"][2]                 # the measured film is orange
    assert abs(means["proxy"][0] - means["proxy"][2]) < 0.25 * means["proxy"][0]   # the proxy is (nearly) grey


def test_c_weight_loader_rejects_malformed_files(tmp_path):
    """bsdfd_create_from_file: the C reader of the .bsdfw format fails loudly (BSDFD_EIO + message)."""
    import ctypes as C
    from bsdf_diffusion_sampling_amd import _lib
    from bsdf_diffusion_sampling_amd import weights as W
    good = open(W.shipped_path("chm_orange_rgb", "disk"), "rb").read()
    L = _lib.lib()
    for name, blob, msg in (("magic", b"XXXXXXXX" + good[8:], "not a BSDFWT01"), ("cut", good[:-8], "payload size"),
                            ("tail", good + b"\0\0\0\0", "payload size"), ("short", good[:40], "BSDFWT01|header")):
        p = tmp_path / (name + ".bsdfw")
        p.write_bytes(blob)
        h = C.c_void_p()
        rc = L.bsdfd_create_from_file(str(p).encode(), 0, C.byref(h))
        assert rc == 3 and not h.value, (name, rc)
        import re
        assert re.search(msg, L.bsdfd_last_error().decode()), L.bsdfd_last_error()
    h = C.c_void_p()
    assert L.bsdfd_create_from_file(W.shipped_path("chm_orange_rgb", "disk").encode(), 0, C.byref(h)) == 0 and h.value
    L.bsdfd_destroy(h)
