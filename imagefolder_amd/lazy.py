"""Statistics that the reference returns as Python floats via .item() (codebook usage: xqgan_model.py:788, quant.py:140) without
stalling the stream: the value is copied to pinned host memory asynchronously and only waited for when somebody reads it
(formatting, float(), arithmetic, numpy conversion).  A train loop that logs every k steps pays one wait every k steps instead of
one device synchronisation in the middle of every forward.

The API seam keeps the reference's contract — `usages: list[float]` (json.dumps / isinstance(u, float) in a caller) —
unless the caller opts in: `materialise(values, lazy)` turns the statistics into real Python floats (one synchronisation, what
upstream's .item() costs) when `lazy` is false, which is the default of the quantizer modules; train.TokenizerTrainStep (and a
hipGraph capture, where a host read is impossible) switch the modules to lazy=True."""
import numpy as np
import torch


class LazyFloat:
    __slots__ = ("_host", "_event", "_scale", "_value", "_dev", "_t")

    def __init__(self, device_scalar: torch.Tensor, scale: float = 1.0):
        self._scale = scale
        self._value = None
        t = device_scalar.detach().reshape(())
        self._dev = None
        self._t = t          # the device scalar itself: lets a caller combine statistics on the device (mean_lazy) without reading them
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture (train.TokenizerTrainStep.capture): no host copy, no event — the scalar lives in the graph's
            # memory pool and is rewritten by every replay; reading it synchronises (a logging-time cost, not a per-step one)
            self._host, self._event, self._dev = None, None, t
        elif t.is_cuda:
            self._host = torch.empty((), dtype=t.dtype, pin_memory=True)
            self._host.copy_(t, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream(t.device))
        else:
            self._host, self._event = t.clone(), None

    def _get(self) -> float:
        if self._dev is not None:          # captured: always re-read (the graph may have been replayed since)
            return float(self._dev.item()) * self._scale
        if self._value is None:
            if self._event is not None:
                self._event.synchronize()
            self._value = float(self._host) * self._scale
        return self._value

    def item(self):
        return self._get()

    def __float__(self):
        return self._get()

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._get(), dtype=dtype or np.float64)

    def __repr__(self):
        return repr(self._get())

    def __format__(self, spec):
        return format(self._get(), spec)

    def __bool__(self):
        return bool(self._get())

    def __eq__(self, o):
        return self._get() == float(o)

    def __lt__(self, o):
        return self._get() < float(o)

    def __le__(self, o):
        return self._get() <= float(o)

    def __gt__(self, o):
        return self._get() > float(o)

    def __ge__(self, o):
        return self._get() >= float(o)

    def __hash__(self):
        return hash(self._get())

    def __add__(self, o):
        return self._get() + float(o)

    __radd__ = __add__

    def __sub__(self, o):
        return self._get() - float(o)

    def __rsub__(self, o):
        return float(o) - self._get()

    def __mul__(self, o):
        return self._get() * float(o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._get() / float(o)

    def __rtruediv__(self, o):
        return float(o) / self._get()

    def __neg__(self):
        return -self._get()

    def __round__(self, n=None):
        return round(self._get(), n)


def lazy_list(device_vector: torch.Tensor, scale: float = 1.0):
    """one asynchronous copy for a whole vector of statistics (per-scale usages) -> list of LazyFloat views"""
    v = device_vector.detach().reshape(-1)
    if not v.is_cuda:
        return [float(x) * scale for x in v.tolist()]
    if torch.cuda.is_current_stream_capturing():
        return [LazyFloat(v[i], scale) for i in range(v.numel())]
    host = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
    host.copy_(v, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(v.device))
    out = []
    for i in range(v.numel()):
        lf = LazyFloat.__new__(LazyFloat)
        lf._host, lf._event, lf._scale, lf._value, lf._dev, lf._t = host[i], ev, scale, None, None, v[i]
        out.append(lf)
    return out


def materialise(values, lazy: bool):
    """list of statistics -> what the API returns: the LazyFloat objects themselves (lazy) or plain floats (the reference's contract)"""
    return list(values) if lazy else [float(v) for v in values]


def mean_lazy(groups):
    """element-wise mean over several lists of statistics (the product branches: xqgan_model.py:287 `sum(us) / product_quant`).
    LazyFloat inputs are averaged ON THE DEVICE and stay lazy — adding them on the host would read each one, i.e. synchronise in
    the middle of the forward (and is impossible inside a hipGraph capture); plain floats are averaged on the host."""
    groups = [list(g) for g in groups]
    n = len(groups)
    if all(isinstance(u, LazyFloat) and u._t is not None for g in groups for u in g):
        dev = torch.stack([torch.stack([u._t.float() * u._scale for u in g]) for g in groups]).mean(dim=0)
        return lazy_list(dev)
    return [sum(float(u) for u in us) / n for us in zip(*groups)]
