"""CPU: the statistics seam (imagefolder_amd/lazy.py).  SURVEY §8b: `usages: list[float]` — json.dumps / isinstance(u, float) in a caller
must work unless the caller opted into lazy reads."""
import json

import torch

from imagefolder_amd.lazy import LazyFloat, lazy_list, materialise, mean_lazy


def test_materialise_gives_python_floats_unless_lazy():
    vals = [LazyFloat(torch.tensor(2.5)), LazyFloat(torch.tensor(7.0), scale=2.0)]
    out = materialise(vals, lazy=False)
    assert out == [2.5, 14.0] and all(type(v) is float for v in out)
    assert json.dumps({"usages": out}) == '{"usages": [2.5, 14.0]}'
    lazy = materialise(vals, lazy=True)
    assert all(isinstance(v, LazyFloat) for v in lazy) and float(lazy[1]) == 14.0 and f"{lazy[0]:.1f}" == "2.5"


def test_mean_lazy_averages_branches_without_reading_lazy_values():
    a = [LazyFloat(torch.tensor(1.0)), LazyFloat(torch.tensor(3.0))]
    b = [LazyFloat(torch.tensor(5.0)), LazyFloat(torch.tensor(9.0))]
    m = mean_lazy([a, b])                          # xqgan_model.py:287: sum(us) / product_quant per scale
    assert [float(v) for v in m] == [3.0, 6.0]
    assert all(v._value is None for v in a + b), "averaging must not read the inputs on the host"
    assert mean_lazy([[1.0, 3.0], [5.0, 9.0]]) == [3.0, 6.0]           # plain floats: host mean
    assert [float(v) for v in lazy_list(torch.tensor([1.0, 2.0]), scale=100.0)] == [100.0, 200.0]
