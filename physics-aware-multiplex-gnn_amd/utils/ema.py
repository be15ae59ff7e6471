"""Exponential moving average of the trainable parameters, API of the reference's utils/ema.py:3-32:

    ema = EMA(model, decay=0.999); ...; ema(model) after every optimiser step;
    ema.assign(model) / ema.resume(model) around evaluation.

Semantics kept: shadow starts as a copy of the parameters; update `shadow = (1-d)*param + d*shadow` with
`d = min(decay, (1+num_updates)/(10+num_updates))` (num_updates defaults to 99999, i.e. d = decay for decay <= 0.99991);
`assign` swaps the shadow in and remembers the live weights, `resume` puts them back.

Implementation: all tensors are updated by two multi-tensor (`torch._foreach_*`) launches instead of one launch per
parameter; with `pamnet_amd.train.Trainer` the same update runs on the flat parameter buffer in a single kernel and this
class is not needed."""
import torch


class EMA(object):
    def __init__(self, model, decay):
        self.decay = decay
        self._names = [n for n, p in model.named_parameters() if p.requires_grad]
        self._shadow = [p.detach().clone() for n, p in model.named_parameters() if p.requires_grad]
        self._live = None

    # the reference exposes dicts; keep them readable for code that inspects `ema.shadow[name]`
    @property
    def shadow(self):
        return dict(zip(self._names, self._shadow))

    @property
    def original(self):
        return {} if self._live is None else dict(zip(self._names, self._live))

    def _params(self, model):
        ps = [p for n, p in model.named_parameters() if p.requires_grad]
        assert len(ps) == len(self._shadow), 'model does not match the one this EMA was built from'
        return ps
    # (models.PAMNet caches its parameter walk, so this is a list copy per call, not a walk of ~250 sub-modules)

    @torch.no_grad()
    def __call__(self, model, num_updates=99999):
        d = min(self.decay, (1.0 + num_updates) / (10.0 + num_updates))
        torch._foreach_mul_(self._shadow, d)
        torch._foreach_add_(self._shadow, [p.detach() for p in self._params(model)], alpha=1.0 - d)

    @torch.no_grad()
    def assign(self, model):
        ps = self._params(model)
        self._live = [p.detach().clone() for p in ps]
        for p, s in zip(ps, self._shadow):
            p.copy_(s)              # in place: parameters may be views of a flat buffer (train.FlatParams)

    @torch.no_grad()
    def resume(self, model):
        assert self._live is not None, 'resume() without assign()'
        for p, w in zip(self._params(model), self._live):
            p.copy_(w)
        self._live = None
