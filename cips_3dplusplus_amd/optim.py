"""Adam on the HIP path (csrc/optim.hip): torch.optim.Adam's update rule -- the one the reference's inversion loop runs on its
three parameter sets every step (/root/reference/exp/cips3d/models/projector_v10.py:279-390, 1210-1216) -- as one bandwidth-bound
launch per 48 parameter tensors.  A torch.optim.Optimizer: param_groups, `lr` / `initial_lr` handling, zero_grad and
state_dict behave as usual; `state[p]` holds `step`, `exp_avg`, `exp_avg_sq` like torch's (`step` is kept as a Python int; a
tensor-valued `step` from a torch.optim.Adam state_dict is accepted and converted).

The update is written through raw device pointers, which autograd's version counters do not see; every updated parameter's
version is therefore bumped explicitly after the launch -- the caches keyed on (data_ptr, _version) (the renderer's packed weights
and stacked biases, `renderer.py:_weights_key`; the forward plan's noise bound, `plan.py:_noise_bound`; `hip.tag_amax`) would
otherwise keep serving the initial values.  Options of torch.optim.Adam that the kernel does not implement (weight_decay,
amsgrad, maximize, ...) are refused, in the constructor and in loaded state, rather than ignored."""
import ctypes as C

import torch

from . import _lib


_UNSUPPORTED = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                    fused=None, decoupled_weight_decay=False)


def _bump_version(p):
    """Tell autograd (and everything keyed on tensor._version) that p's storage was rewritten behind its back."""
    try:
        torch.autograd.graph.increment_version(p)
    except AttributeError:          # older torch: an in-place no-op does the same
        p.add_(0)


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **unsupported):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        self._check_options(unsupported)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @staticmethod
    def _check_options(opts):
        for k, v in opts.items():
            if k not in _UNSUPPORTED:
                raise TypeError(f"HipAdam: unknown option {k!r}")
            if v not in (_UNSUPPORTED[k], None, False, 0):
                raise NotImplementedError(f"HipAdam implements plain Adam only: {k}={v!r} is not supported")

    def _collect(self, entries, group_of, hypers, keep, updated):
        """Append this optimiser's work to the shared lists of one cips3d_adam_step_groups call."""
        for group in self.param_groups:
            # (param_groups of a loaded torch.optim.Adam state_dict carry its other options)
            self._check_options({k: v for k, v in group.items() if k in _UNSUPPORTED})
            slot_of_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("HipAdam updates contiguous fp32 CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1          # (a tensor step of torch's own Adam state is accepted)
                updated.append(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype != torch.float32:
                    g = g.float()
                keep.append(g)
                hi = slot_of_step.get(st["step"])
                if hi is None:                            # one hyper-parameter set per (group, step count)
                    hi = slot_of_step[st["step"]] = len(hypers)
                    hypers.append(_lib.AdamHyper(float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                                 float(group["eps"]), int(st["step"])))
                entries.append(_lib.AdamEntry(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
                group_of.append(hi)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        step_many([self])
        return loss


@torch.no_grad()
def step_many(optimizers):
    """One step of several HipAdam optimisers in SHARED launches (cips3d_adam_step_groups: 48 tensors per launch whatever group or
    optimiser they belong to) -- the three optimisers of an inversion step are 3 launches instead of 6."""
    entries, group_of, hypers, keep, updated = [], [], [], [], []
    for o in optimizers:
        if not isinstance(o, HipAdam):
            raise TypeError("step_many takes HipAdam optimisers")
        o._collect(entries, group_of, hypers, keep, updated)
    if entries:
        arr = (_lib.AdamEntry * len(entries))(*entries)
        gof = (C.c_int * len(group_of))(*group_of)
        hyp = (_lib.AdamHyper * len(hypers))(*hypers)
        _lib.check(_lib.load().cips3d_adam_step_groups(arr, gof, len(entries), hyp, len(hypers), _lib.stream_ptr()),
                   "cips3d_adam_step_groups")
    for p in updated:
        _bump_version(p)
