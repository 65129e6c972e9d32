"""The 1e-4 parity claim as a STATISTIC on every shipped material, at plugin level (north_star: "PDF
rel-err <= 1e-4 on the paper's measured BSDFs").  77 weight sets x 65 536 queries x {sample, pdf at produced directions, pdf at
fresh directions} x both tilings through the C ABI against the pinned fp64 oracle; the assertion is on the UPPER end of the
bootstrap 95 % interval of each p99 (tests/parity77.py).  The record of the run goes to gpurun_out/plugin_parity_77sets.json;
the committed copy is profiles/r06_plugin_parity_77sets.json (tools/plugin_parity_sweep.py writes the same record).

Lines matched: rendering/brdf_measured_disk.py:59-82,112-124, brdf_measured_spherical.py:35-39,69-91,122-137,
bsdf_myresult.py:59-84,115-133, rendering/utils/mlp_brdf_sampling.py:17-181."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_every_shipped_material_plugin_level_p99_interval_below_1e_4():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import parity77 as P
    from conftest import ROOT
    n = int(os.environ.get("BSDFD_PARITY77_N", "65536"))
    rec = P.run(n=n)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "plugin_parity_77sets.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass
    s = rec["summary"]
    print(json.dumps({k: s[k] for k in ("queries_per_set", "sets", "worst_det_not_exempt", "median_of_p99_det", "seconds")}))
    assert s["sets"] == 77
    assert s["queries_per_set"] >= 65536 or "BSDFD_PARITY77_N" in os.environ
    assert not s["failures"], s["failures"]
    # materials where the reference's OWN fp32 evaluation is above 1e-4 on the same rows (and the kernel no worse than 1.25 x it):
    # named here so that the list cannot grow unnoticed — bsdf_23 pdf() at fresh directions: reference fp32 1.5e-3, kernel 5.6e-4
    assert set(s["exempt_reference_fp32_also_above_bound"]) <= {"bsdf_23_spherical"}, s["exempt_reference_fp32_also_above_bound"]
    # ... and nothing else above the bound (until round 6's scaled output-layer lo rows the 16-query tiling held one (set, call) at
    # 1.05e-4: cc_amber_citrine_rgb_disk, pdf() at fresh directions; tests/parity77.py, KNOWN_ABOVE_BOUND)
    assert not s["known_above_bound_under_their_cap"], s["known_above_bound_under_their_cap"]
    # disk rows whose fp64 r^2 is within 1e-5 of the 0.995 guard and that kernel and oracle decide differently: a handful at most
    assert s["threshold_rows"] <= 20, s["threshold_rows"]
