"""NA2D pin against the REAL natten (0.17.1 is what the reference pins: .github/workflows/ci.yml:53), whenever that
package is importable. It is not installed in the build image or on the GPU box, so this test SKIPS there and NA2D
parity stays "unpinned" (two independent restatements + window-property tests, tests/test_na2d_oracle.py); the day a
natten wheel is present this test tightens the pin without further work (oracle/refimport.py then also prefers the
real package when generating fixtures)."""
import pytest
import torch

natten = pytest.importorskip("natten")


@pytest.mark.parametrize("H,W,heads,D,dil", [(12, 12, 2, 8, 1), (14, 15, 4, 8, 2), (25, 25, 8, 16, 1), (20, 21, 4, 32, 2)])
def test_restatement_matches_real_natten(H, W, heads, D, dil):
    from natten.functional import na2d_av, na2d_qk

    from oracle import na2d_ref

    g = torch.Generator().manual_seed(H * 100 + W)
    q, k, v = (torch.randn(2, heads, H, W, D, generator=g) for _ in range(3))
    a_ref = na2d_ref.na2d_qk(q, k, 3, dil)
    a_nat = na2d_qk(q, k, kernel_size=3, dilation=dil)
    assert (a_ref - a_nat).abs().max() <= 1e-5
    p = a_ref.softmax(dim=-1)
    o_ref = na2d_ref.na2d_av(p, v, 3, dil)
    o_nat = na2d_av(p, v, kernel_size=3, dilation=dil)
    assert (o_ref - o_nat).abs().max() <= 1e-5


def test_module_matches_real_natten():
    from oracle import na2d_ref

    torch.manual_seed(0)
    real = natten.NeighborhoodAttention2D(dim=32, num_heads=4, kernel_size=3, dilation=2)
    mine = na2d_ref.NeighborhoodAttention2D(dim=32, num_heads=4, kernel_size=3, dilation=2)
    mine.load_state_dict(real.state_dict())
    x = torch.randn(2, 14, 15, 32)
    assert (mine(x) - real(x)).abs().max() <= 1e-5
