"""NA2D restatements: the vectorised gather and the scalar loops must agree, and the
clamped/dilated window rule must have the published properties."""
import pytest
import torch

from oracle import na2d_ref as N


@pytest.mark.parametrize("L,K,d", [(100, 3, 1), (100, 3, 2), (101, 3, 2), (50, 3, 1), (25, 3, 1), (13, 3, 1), (28, 3, 2), (7, 3, 2), (9, 5, 1), (13, 5, 2)])
def test_window_properties(L, K, d):
    for i in range(L):
        s = N.window_start(i, L, K, d)
        taps = [s + j * d for j in range(K)]
        assert 0 <= taps[0] and taps[-1] < L  # no zero padding: all in bounds
        assert all(t % d == i % d for t in taps)  # same residue class
        assert i in taps  # the query attends to itself
        if i - (K // 2) * d >= 0 and i + (K // 2) * d < L:
            assert taps[K // 2] == i  # centred in the interior


def test_published_examples():
    w = lambda i, L: [N.window_start(i, L, 3, 2) + 2 * j for j in range(3)]
    assert w(0, 100) == [0, 2, 4] and w(1, 100) == [1, 3, 5]
    assert w(98, 100) == [94, 96, 98] and w(99, 100) == [95, 97, 99]
    assert w(100, 101) == [96, 98, 100] and w(99, 101) == [95, 97, 99]


@pytest.mark.parametrize("H,W,K,d", [(9, 11, 3, 1), (12, 10, 3, 2), (7, 7, 3, 2), (8, 13, 5, 1)])
def test_two_restatements_agree(H, W, K, d):
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(2, 3, H, W, 4, generator=g) for _ in range(3))
    a = N.na2d_av(N.na2d_qk(q, k, K, d).softmax(-1), v, K, d)
    b = N.na2d_scalar(q, k, v, K, d)
    assert (a - b).abs().max() < 1e-5


def test_interior_equals_unfold_attention():
    """With d=1 interior pixels see the plain centred 3x3 window."""
    g = torch.Generator().manual_seed(5)
    B, h, H, W, D = 1, 2, 8, 9, 4
    q, k, v = (torch.randn(B, h, H, W, D, generator=g) for _ in range(3))
    out = N.na2d_av(N.na2d_qk(q, k, 3, 1).softmax(-1), v, 3, 1)
    ku = torch.nn.functional.unfold(k.permute(0, 1, 4, 2, 3).reshape(B, h * D, H, W), 3).reshape(B, h, D, 9, H - 2, W - 2)
    vu = torch.nn.functional.unfold(v.permute(0, 1, 4, 2, 3).reshape(B, h * D, H, W), 3).reshape(B, h, D, 9, H - 2, W - 2)
    qi = q[:, :, 1:-1, 1:-1]
    logits = torch.einsum("bhxyd,bhdkxy->bhxyk", qi, ku)
    ref = torch.einsum("bhxyk,bhdkxy->bhxyd", logits.softmax(-1), vu)
    assert (out[:, :, 1:-1, 1:-1] - ref).abs().max() < 1e-5


def _na_subgrid_unfold(q, k, v, K, d):
    """A THIRD formulation that shares no code (and no window rule) with oracle/na2d_ref.py: dilated neighborhood
    attention as published (DiNAT): split the plane into its d x d residue-class sub-grids (pixel-unshuffle), run plain
    (dilation 1) neighborhood attention inside every sub-grid, where the window of query j on an axis of length n is the
    K-wide window CENTRED at clamp(j, K//2, n-1-K//2). All centred windows of a sub-grid come from torch's own
    F.unfold; edge queries pick the unfold column of their clamped centre. q is already scaled."""
    import torch.nn.functional as F

    B, h, H, W, D = q.shape
    out = torch.empty_like(q)
    n = K // 2
    for ry in range(d):
        for rx in range(d):
            qs, ks, vs = (t[:, :, ry::d, rx::d] for t in (q, k, v))
            Hs, Ws = qs.shape[2], qs.shape[3]
            unf = lambda t: F.unfold(t.permute(0, 1, 4, 2, 3).reshape(B, h * D, Hs, Ws), K).reshape(
                B, h, D, K * K, Hs - 2 * n, Ws - 2 * n)
            ku, vu = unf(ks), unf(vs)
            cy = torch.arange(Hs).clamp(n, Hs - 1 - n) - n   # unfold row of each query's clamped centre
            cx = torch.arange(Ws).clamp(n, Ws - 1 - n) - n
            kq = ku[:, :, :, :, cy][:, :, :, :, :, cx]        # [B,h,D,KK,Hs,Ws]
            vq = vu[:, :, :, :, cy][:, :, :, :, :, cx]
            logits = torch.einsum("bhxyd,bhdkxy->bhxyk", qs, kq)
            out[:, :, ry::d, rx::d] = torch.einsum("bhxyk,bhdkxy->bhxyd", logits.softmax(-1), vq)
    return out


@pytest.mark.parametrize("heads,H,W,d", [(4, 100, 100, 2), (4, 50, 50, 1), (8, 25, 25, 1),   # unet_parts.py:19-40 at
                                         (4, 28, 28, 2), (4, 14, 14, 1), (8, 7, 7, 1),        # 100^2 and at 28^2
                                         (2, 11, 13, 3), (2, 9, 101, 2)])                     # ragged residue classes
def test_subgrid_unfold_formulation_matches_everywhere(heads, H, W, d):
    """oracle/na2d_ref.py (what every default-configuration fixture flows through, natten being absent) against the
    sub-grid / F.unfold formulation on ALL pixels -- edges included -- at the reference's three call-site shapes
    (C = 128: head dims 32 / 32 / 16). This does not pin natten; it removes the shared-author risk of the edge rule:
    the two formulations have no line in common (window_start is not used by the third one)."""
    g = torch.Generator().manual_seed(17)
    D = 128 // heads
    q, k, v = (torch.randn(1, heads, H, W, D, generator=g) for _ in range(3))
    q = q * D ** -0.5
    a = N.na2d_av(N.na2d_qk(q, k, 3, d).softmax(-1), v, 3, d)
    b = _na_subgrid_unfold(q, k, v, 3, d)
    assert (a - b).abs().max() < 2e-5
