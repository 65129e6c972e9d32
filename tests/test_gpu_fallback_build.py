"""The fallback build of the flow kernels (-DBSDFD_NO_ASYNC_LDS: compiler-managed LDS reads, what `_lib.build()` ships when the
assembly check of the asynchronous reads fails on a toolchain) must be a correct library, not a theoretical one: build it here,
load it in a child process through $BSDFD_LIB_PATH and hold it to the oracle and to the product build bit for bit."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_CHILD = r"""
import json, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np, torch
from conftest import load_case
from bsdf_diffusion_sampling_amd import _lib
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
from oracle import bsdf_oracle as O
dev = torch.device("cuda")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
out = {{"version": _lib.lib().bsdfd_version().decode()}}
for stem in ("aniso_miro_7_rgb_disk", "aniso_miro_7_rgb_spherical", "aniso_miro_7_rgb_spherical_complex"):
    g, fw = load_case(stem)
    T = int(g["meta_T"])
    s = FlowSampler(fw)
    x, p = s.network_sampling(t(g["wi"]), t(g["x0"]), T=T)
    pp = s.network_pdf(x, t(g["wi"]), T=T)
    orc = O.Oracle(fw)
    xo, po = orc.network_sampling(g["wi"], g["x0"], T)
    _, acc = orc.flow(g["x0"], g["wi"], T, False)
    ok = (np.abs(acc) > 1e-3) & (np.abs(acc) < 1e3)
    ok &= np.abs(po) > 1e-6 * np.percentile(np.abs(po[ok]), 99)
    rel = np.abs(p.cpu().numpy() - po)[ok] / np.abs(po[ok])
    out[stem] = {{"x_err": float(np.abs(x.cpu().numpy() - xo).max()), "p99": float(np.percentile(rel, 99)),
                 "x": x.cpu().numpy().tobytes().hex()[:4096], "sum_p": float(p.double().sum()), "sum_pp": float(pp.double().sum())}}
    s.close()
# the f16 samples-only call (packed-fp16 sigmoids: hand-packed SDWA form in the product, compiler-written in the compiler-only build)
g, fw = load_case("aniso_miro_7_rgb_spherical_complex")
for tile in (32, 16):
    s = FlowSampler(fw, precision="f16", tile=tile)
    xs = s.flow_samples_only(t(g["wi"]), t(g["x0"]), T=32)
    out["teacher_f16_tile%d" % tile] = {{"x": xs.cpu().numpy().tobytes().hex()[:4096], "sum": float(xs.double().sum()),
                                        "finite": bool(torch.isfinite(xs).all())}}
    s.close()
print("RESULT" + json.dumps(out))
"""


def _run(lib_path):
    env = dict(os.environ)
    if lib_path:
        env["BSDFD_LIB_PATH"] = lib_path
    else:
        env.pop("BSDFD_LIB_PATH", None)
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][6:])


def test_fallback_build_is_correct_and_agrees_with_the_product_build(tmp_path, monkeypatch):
    if not shutil.which("hipcc"):
        pytest.skip("needs hipcc")
    from bsdf_diffusion_sampling_amd import _lib
    real, calls = _lib._check_asm, []

    def doctored(path):
        calls.append(path)
        return {"flow_kernel<doctored>": ["forced by the test"]} if len(calls) == 1 else real(path)
    monkeypatch.setattr(_lib, "_check_asm", doctored)
    out = str(tmp_path / "libbsdfd_fallback.so")
    _lib.build(force=True, lib_path=out)
    assert _lib.build_info(out)["variant"] == "plain"
    fb, prod = _run(out), _run(None)
    assert "fallback" in fb["version"] and "asynchronous" in prod["version"]
    for stem in ("aniso_miro_7_rgb_disk", "aniso_miro_7_rgb_spherical", "aniso_miro_7_rgb_spherical_complex"):
        assert fb[stem]["x_err"] < 1e-4 and fb[stem]["p99"] < 1e-4, (stem, fb[stem])
        # the same arithmetic in the same order: only WHEN the fragments are fetched differs
        assert fb[stem]["x"] == prod[stem]["x"] and fb[stem]["sum_p"] == prod[stem]["sum_p"] and fb[stem]["sum_pp"] == prod[stem]["sum_pp"], stem


def test_compiler_only_build_is_correct_and_agrees_with_the_product_build(tmp_path, monkeypatch):
    """BSDFD_COMPILER_ONLY_BUILD=1 (no inline-asm LDS reads, no SDWA sigmoids, nothing parsed — the build for a toolchain the
    assembly parser does not know): the same arithmetic in the same order everywhere, so every result equals the product build's bit
    for bit — the split3 operators AND the f16 samples-only call whose sigmoids the compiler now writes."""
    if not shutil.which("hipcc"):
        pytest.skip("needs hipcc")
    from bsdf_diffusion_sampling_amd import _lib
    monkeypatch.setenv("BSDFD_COMPILER_ONLY_BUILD", "1")
    out = str(tmp_path / "libbsdfd_compiler_only.so")
    _lib.build(force=True, lib_path=out)
    monkeypatch.delenv("BSDFD_COMPILER_ONLY_BUILD")
    assert _lib.build_info(out)["compiler_only"] is True
    co, prod = _run(out), _run(None)
    assert "COMPILER-ONLY BUILD" in co["version"] and "COMPILER-ONLY" not in prod["version"]
    for stem in ("aniso_miro_7_rgb_disk", "aniso_miro_7_rgb_spherical", "aniso_miro_7_rgb_spherical_complex"):
        assert co[stem]["x_err"] < 1e-4 and co[stem]["p99"] < 1e-4, (stem, co[stem])
        assert co[stem]["x"] == prod[stem]["x"] and co[stem]["sum_p"] == prod[stem]["sum_p"] and co[stem]["sum_pp"] == prod[stem]["sum_pp"], stem
    for key in ("teacher_f16_tile32", "teacher_f16_tile16"):
        assert co[key]["finite"] and co[key]["x"] == prod[key]["x"] and co[key]["sum"] == prod[key]["sum"], key
