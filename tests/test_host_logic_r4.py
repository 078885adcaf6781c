"""Host-side logic added in round 4 that needs no GPU: the flat parameter store's contiguity groups and the inline
fallback of engine.spawn / join."""
import torch
import torch.nn as nn


class _Heads(nn.Module):
    def __init__(self, c=8):
        super().__init__()
        self.a = nn.Conv2d(c, 3, 3, bias=False)
        self.mid = nn.BatchNorm2d(3)
        self.b = nn.Conv2d(c, 3, 3, bias=False)
        self.tail = nn.Conv2d(3, 1, 3)
        self.c = nn.Conv2d(c, 3, 3, bias=False)

    def cn_contiguous_params(self):
        return [[self.a.weight, self.b.weight, self.c.weight]]


def test_param_store_makes_declared_groups_adjacent():
    from cultionet_amd.engine import ParamStore

    m = nn.Sequential(nn.Conv2d(4, 8, 3), _Heads(8), nn.Conv2d(3, 1, 1))
    params = list(m.parameters())
    out = ParamStore._with_contiguous_groups(m, params)
    assert len(out) == len(params) and {id(p) for p in out} == {id(p) for p in params}
    h = m[1]
    i = [id(p) for p in out].index(id(h.a.weight))
    assert out[i + 1] is h.b.weight and out[i + 2] is h.c.weight  # the group, at the place of its first member
    rest = [p for p in out if all(p is not q for q in (h.a.weight, h.b.weight, h.c.weight))]
    assert [id(p) for p in rest] == [id(p) for p in params if all(p is not q for q in (h.a.weight, h.b.weight, h.c.weight))]


def test_groups_with_members_that_would_be_padded_are_ignored():
    from cultionet_amd.engine import ParamStore

    h = _Heads(8)
    h.b = nn.Conv2d(8, 3, 1, bias=False)  # 24 elements: a multiple of 4 -> fine; make one that is not
    h.c = nn.Conv2d(3, 1, 1, bias=False)  # 3 elements: the store pads its slice to 16 bytes, adjacency would break
    params = list(h.parameters())
    assert [id(p) for p in ParamStore._with_contiguous_groups(h, params)] == [id(p) for p in params]


def test_spawn_runs_inline_without_a_gpu():
    from cultionet_amd import engine as E

    x = E.Var(torch.zeros(1, 2, 3, 3))
    seen = []
    br, res = E.spawn(lambda v: seen.append(v) or 7, [x], 0)
    assert br is None and res == 7 and seen[0] is x  # no alias, no stream: the plain call
    E.join([br])  # a no-op
