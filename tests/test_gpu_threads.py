"""include/bsdfd.h: "calls are re-entrant across host threads and streams".  Held to it here: several host threads, each on
its own HIP stream, drive ONE handle (and, second, a handle each) at once — sample, pdf, the fused call and the samples-only
call interleaved, with launch timing switched on in one pass (its counters are the library's only mutable state) — and every
result must equal, bit for bit, what the same call returned alone.  ctypes releases the GIL for the duration of a foreign call,
so the C entry points really do overlap."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import load_case  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda", 0)


def _dirs(rng, n, lo):
    z, ph = rng.uniform(lo, 1.0, size=n), rng.uniform(0, 2 * np.pi, size=n)
    r = np.sqrt(1 - z * z)
    return torch.from_numpy(np.stack([r * np.cos(ph), r * np.sin(ph), z], 1).astype(np.float32)).to(_dev())


def _work(s, T, wi, wl, seed):
    """One thread's round: four kinds of call on the caller's current stream."""
    wo, p = s.plugin_sample(wi, None, T=T, seed=seed, offset=5)
    pl = s.plugin_pdf(wi, wl, T=T)
    wo2, p2, pl2 = s.plugin_sample_pdf(wi, wl, None, T=T, seed=seed, offset=5)
    xs = s.flow_samples_only(wi[:, :2].contiguous(), wl[:, :2].contiguous(), T=T)
    return [wo, p, pl, wo2, p2, pl2, xs]


@pytest.mark.parametrize("shared_handle", [True, False])
@pytest.mark.parametrize("profiling", [False, True])
@pytest.mark.parametrize("stem", ["chm_orange_rgb_disk", "aniso_miro_7_rgb_spherical"])
def test_host_threads_on_their_own_streams(stem, profiling, shared_handle, monkeypatch):
    monkeypatch.setenv("BSDFD_HOST_BINDING", "ctypes")
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    _, fw = load_case(stem)
    T = 4 if fw.domain == 0 else 8
    n_threads, rounds = 4, 6
    samplers = [FlowSampler(fw, precision="split3") for _ in range(1 if shared_handle else n_threads)]
    rng = np.random.default_rng(11)
    sizes = [30001, 4097, 65536, 1000]
    inputs = [(_dirs(rng, sizes[i], 0.05), _dirs(rng, sizes[i], 0.02)) for i in range(n_threads)]
    # the reference results: each thread's calls alone, on the default stream
    ref = [[t.clone() for t in _work(samplers[i % len(samplers)], T, *inputs[i], seed=100 + i)] for i in range(n_threads)]
    torch.cuda.synchronize()
    for s in samplers:
        s.set_profiling(profiling)
    errors, start = [], threading.Barrier(n_threads)

    def body(i):
        try:
            s = samplers[i % len(samplers)]
            stream = torch.cuda.Stream(device=_dev())
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(rounds):
                    out = _work(s, T, *inputs[i], seed=100 + i)
                    stream.synchronize()
                    for k, (a, b) in enumerate(zip(out, ref[i])):
                        if not torch.equal(a, b):
                            bad = int((a != b).sum())
                            raise AssertionError(f"thread {i} output {k}: {bad} elements differ from the call made alone")
        except BaseException as exc:  # noqa: BLE001 - reported in the main thread
            errors.append(exc)
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=body, args=(i,)) for i in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), "a thread hung"
    assert not errors, errors
    if profiling:
        # every launch of the concurrent phase was counted exactly once: 4 calls per round
        # (the fused call is one launch), `rounds` rounds per thread
        total = sum(s.profile_read()[0] for s in samplers)
        assert total == 4 * rounds * n_threads, total
    for s in samplers:
        s.set_profiling(False)
        s.close()
