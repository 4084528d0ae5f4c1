"""Single-pass Adam over the model's parameters, complex ones included.

The reference patches torch's Adam for complex parameters (``makani/third_party/torch``) and
steps outside the captured graph (``makani/utils/trainer.py:762-763``); torch >= 2.1 handles
complex parameters natively but only in the multi-pass ``foreach`` form (7 HBM passes over the
2.3 GB of spectral weights).  Here every parameter is handed to ``torch.optim.Adam(fused=True)``
as a real leaf view sharing its storage (complex64 -> ``view_as_real``), which is the same
arithmetic (Adam treats a complex number as two reals) in one multi-tensor kernel.
"""
import torch


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.views = []
        for p in self.params:
            v = (torch.view_as_real(p.data) if p.is_complex() else p.data).detach()
            self.views.append(v)
        self.opt = torch.optim.Adam(self.views, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)

    @property
    def param_groups(self):
        return self.opt.param_groups

    def zero_grad(self, set_to_none=True):
        for p, v in zip(self.params, self.views):
            v.grad = None
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.grad = None
                continue
            if g.stride() != p.stride():      # the fused kernel walks storage linearly: layouts must agree
                g2 = torch.empty_like(p.data)  # preserve_format -> the parameter's strides
                g2.copy_(g)
                g = g2
            v.grad = torch.view_as_real(g) if g.is_complex() else g
        self.opt.step()

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd)
