"""Adam on the HIP path (csrc/optim.hip): torch.optim.Adam's update rule -- the one the reference's inversion loop runs on its
three parameter sets every step (/root/reference/exp/cips3d/models/projector_v10.py:279-390, 1210-1216) -- as one bandwidth-bound
launch per 48 parameter tensors.  A torch.optim.Optimizer: param_groups, `lr` / `initial_lr` handling, zero_grad and
state_dict behave as usual; `state[p]` holds `step` (int), `exp_avg`, `exp_avg_sq` like torch's."""
import ctypes as C

import torch

from . import _lib


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            by_step = {}
            keep = []
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
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype != torch.float32:
                    g = g.float()
                keep.append(g)
                e = _lib.AdamEntry(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                by_step.setdefault(st["step"], []).append(e)
            for step, entries in by_step.items():
                arr = (_lib.AdamEntry * len(entries))(*entries)
                _lib.check(lib.cips3d_adam_step(arr, len(entries), float(group["lr"]), float(group["betas"][0]),
                                                float(group["betas"][1]), float(group["eps"]), int(step), _lib.stream_ptr()),
                           "cips3d_adam_step")
        return loss
