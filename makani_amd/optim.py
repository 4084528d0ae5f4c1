"""Single-pass Adam over the model's parameters, complex ones included.

The reference patches torch's Adam for complex parameters (``makani/third_party/torch``) and
steps outside the captured graph (``makani/utils/trainer.py:762-763``).  Adam treats a complex
number as two reals, so every parameter is stepped through a real view of its storage
(complex64 -> ``view_as_real``):

* large dense tensors (the spectral weights: 283 M of the net's 289 M parameters) by
  ``mk_adam_step`` -- one HIP streaming pass per tensor over its storage in memory order;
* the many small ones by ``torch.optim.Adam(fused=True)`` in one multi-tensor launch.

``overlap_backward=k`` (opt-in, 0 = off): the update of a large tensor is launched on a side stream as soon as its gradient
has been accumulated ``k`` times in this step (``k`` = backward passes per step: 1, or the number of micro-batches), so the
HBM-bound streaming pass runs under the remaining backward kernels (which leave most of the HBM bandwidth idle) instead
of after them; ``step()`` launches whatever is left and joins the side stream.  The arithmetic and the result are the same
as without it.  (On one MI355X the step time does not change: the backward kernels that run beside an Adam pass slow
down by what the pass would have cost alone -- ``bench.py`` leaves it off.)  Only valid when nothing is done to these gradients between backward and ``step()``: no gradient clipping
or scaling, no data-parallel averaging (the spectral weights are sharded, not shared, over the model-parallel groups, so
``reduce_shared_gradients`` does not touch them).  Under stream capture the hook does nothing (the captured graph ends with
backward; the optimizer steps outside it, ``trainer.py:762-763``).
"""
import torch

from . import _lib

_BIG = 1 << 20      # elements: below this the per-tensor launch is not worth it


def _flat_storage_view(t):
    """1-D view over a dense tensor's elements in memory order (works for permuted-contiguous layouts), or None."""
    if t.is_contiguous():
        return t.view(-1)
    order = sorted(range(t.dim()), key=lambda d: -t.stride(d))
    tp = t.permute(order)
    return tp.reshape(-1) if tp.is_contiguous() else None


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, overlap_backward=0):
        self.params = [p for p in params if p.requires_grad]
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.views, self._small, self._big = [], [], []
        self.overlap_backward = int(overlap_backward)
        self._side = None           # side stream of the overlapped updates
        self._hooks = []
        for p in self.params:
            v = (torch.view_as_real(p.data) if p.is_complex() else p.data).detach()
            self.views.append(v)
            flat = _flat_storage_view(v) if (v.is_cuda and v.dtype == torch.float32 and v.numel() >= _BIG) else None
            if flat is not None and flat.data_ptr() % 16 == 0:
                st = {"p": p, "flat": flat, "m": torch.zeros_like(flat), "v": torch.zeros_like(flat), "step": 0,
                      "acc": 0, "done": False}
                self._big.append(st)
                if self.overlap_backward > 0:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, st=st: self._grad_ready(st)))
            else:
                self._small.append((p, v))
        self.opt = None
        self._own_groups = [dict(self.defaults, params=[])]
        if self._small:
            self.opt = torch.optim.Adam([v for _, v in self._small], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                        fused=all(v.is_cuda for _, v in self._small))

    @property
    def param_groups(self):
        """ONE persistent list (torch's for the small tensors, else the optimizer's own): a scheduler that writes
        ``group["lr"]`` reaches both the torch optimizer and the streaming passes, which read group 0."""
        return self.opt.param_groups if self.opt is not None else self._own_groups

    def zero_grad(self, set_to_none=True):
        for st in self._big:
            st["acc"] = 0
        for p, v in zip(self.params, self.views):
            v.grad = None
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @staticmethod
    def _grad_like_param(p):
        g = p.grad
        if g is not None and g.stride() != p.stride():      # the kernels walk storage linearly: layouts must agree
            g2 = torch.empty_like(p.data)                    # preserve_format -> the parameter's strides
            g2.copy_(g)
            g = g2
        return g

    def _step_big(self, st, stream):
        """One streaming pass over a large tensor on ``stream`` (a raw HIP stream handle)."""
        g = self._grad_like_param(st["p"])
        if g is None:
            return
        hp = self.param_groups[0]
        lr, (b1, b2), eps, wd = hp["lr"], hp["betas"], hp["eps"], hp["weight_decay"]
        gf = _flat_storage_view(torch.view_as_real(g) if g.is_complex() else g)
        assert gf is not None and gf.dtype == torch.float32 and gf.numel() == st["flat"].numel()
        st["step"] += 1
        _lib.check(_lib.load().mk_adam_step(st["flat"].data_ptr(), gf.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(),
                                            gf.numel(), float(lr), float(b1), float(b2), float(eps), float(wd), st["step"],
                                            stream), "mk_adam_step")

    def _grad_ready(self, st):
        """Post-accumulate hook of a large tensor (``overlap_backward``): after the last accumulation of the step its update
        goes to the side stream, ordered behind everything the accumulating stream has queued so far (the kernels that read
        the old weights and wrote the gradient)."""
        st["acc"] += 1
        if st["acc"] != self.overlap_backward or st["done"] or not st["p"].is_cuda:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=st["p"].device)    # gfx950 offers priorities (0, -1): 0 is the lowest
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self._step_big(st, self._side.cuda_stream)
        st["done"] = True

    def step(self):
        for p, v in self._small:
            g = self._grad_like_param(p)
            v.grad = None if g is None else (torch.view_as_real(g) if g.is_complex() else g)
        if self.opt is not None:
            self.opt.step()
        if not self._big:
            return
        stream = torch.cuda.current_stream().cuda_stream
        joined = False
        for st in self._big:
            if st["done"]:              # already under way on the side stream: the gradient stays alive until the join below
                st["done"] = False
                joined = True
            else:
                self._step_big(st, stream)
            st["acc"] = 0
        if joined:
            torch.cuda.current_stream().wait_stream(self._side)

    def state_dict(self):
        return {"small": None if self.opt is None else self.opt.state_dict(),
                "big": [{"m": st["m"], "v": st["v"], "step": st["step"]} for st in self._big]}

    def load_state_dict(self, sd):
        if not isinstance(sd, dict) or set(sd) != {"small", "big"}:
            raise ValueError("FusedAdam.load_state_dict: unrecognised layout (expected keys {'small', 'big'}; a plain "
                             "torch.optim.Adam state dict cannot be mapped onto the split small / large tensor state)")
        if len(sd["big"]) != len(self._big):
            raise ValueError(f"FusedAdam.load_state_dict: {len(sd['big'])} large-tensor states for {len(self._big)} tensors")
        if self.opt is not None and sd.get("small") is not None:
            self.opt.load_state_dict(sd["small"])
        for st, src in zip(self._big, sd.get("big", [])):
            st["m"].copy_(src["m"])
            st["v"].copy_(src["v"])
            st["step"] = int(src["step"])
