"""Stand-alone positional encoding (csrc/encoding.hip) vs the oracle and vs the reference's formula
restated with torch CPU ops (rendering/utils/model.py:9-57)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import bsdf_oracle as O  # noqa: E402


def _ref(t, P, include_input=True, log_sampling=True):
    enc = [t] if include_input else []
    if log_sampling:
        f = 2.0 ** torch.linspace(0.0, P - 1, P, dtype=t.dtype)
    else:
        f = torch.linspace(2.0 ** 0.0, 2.0 ** (P - 1), P, dtype=t.dtype)
    for freq in f:
        for fn in (torch.sin, torch.cos):
            enc.append(fn(t * freq))
    return enc[0] if len(enc) == 1 else torch.cat(enc, dim=-1)


@pytest.mark.parametrize("n", [0, 1, 127, 128, 129, 1000, 65537])
def test_matches_oracle_disk_shapes(n):
    from bsdf_diffusion_sampling_amd.encoding import positional_encoding_1
    g = torch.Generator().manual_seed(n)
    x = (torch.rand(n, 2, generator=g) * 2 - 1) * 1.6
    for P in (5, 3):
        got = positional_encoding_1(x.cuda(), P).cpu()
        assert got.shape == (n, 2 + 4 * P)
        if n:
            want = O.positional_encoding(x.numpy().astype(np.float64), P)
            assert np.abs(got.numpy() - want).max() < 5e-7
            assert torch.equal(got[:, :2], x)


@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("P", [0, 1, 4, 6])
@pytest.mark.parametrize("include_input,log_sampling", [(True, True), (False, True), (True, False), (False, False)])
def test_matches_reference_formula(dim, P, include_input, log_sampling):
    from bsdf_diffusion_sampling_amd.encoding import positional_encoding_1
    if P == 0 and not include_input:
        with pytest.raises(RuntimeError, match="empty encoding"):
            positional_encoding_1(torch.zeros(4, dim).cuda(), P, include_input, log_sampling)
        return
    x = (torch.rand(3, 77, dim, generator=torch.Generator().manual_seed(1)) * 2 - 1) * 3.2
    got = positional_encoding_1(x.cuda(), P, include_input, log_sampling).cpu()
    want = _ref(x, P, include_input, log_sampling)
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-6   # sincosf vs torch's CPU sin/cos: <= 2 ulp at |arg| <= 100


def test_errors_are_loud():
    from bsdf_diffusion_sampling_amd.encoding import positional_encoding_1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        positional_encoding_1(torch.zeros(4, 2), 5)
    with pytest.raises(TypeError):
        positional_encoding_1(torch.zeros(4, 2, dtype=torch.float64).cuda(), 5)
    with pytest.raises(RuntimeError, match="bands"):
        positional_encoding_1(torch.zeros(4, 2).cuda(), 17)


def test_longest_rows():
    """dim 8 x 16 bands + input = 264 floats per row: the workgroup tile shrinks to stay within LDS."""
    from bsdf_diffusion_sampling_amd.encoding import positional_encoding_1
    x = (torch.rand(1000, 8, generator=torch.Generator().manual_seed(2)) * 2 - 1) * 1e-3   # 2^15 x: keep |arg| modest
    got = positional_encoding_1(x.cuda(), 16).cpu()
    want = _ref(x.double(), 16).float()
    assert got.shape == (1000, 8 * 33) and (got - want).abs().max() < 5e-6
