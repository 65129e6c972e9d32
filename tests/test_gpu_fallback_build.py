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
