"""Optimizer for the flat-parameter mode (TGAT.flatten_parameters): torch.optim.Adam's update rule, one HIP kernel per step.

The reference's trainers build `torch.optim.Adam(model.parameters(), lr, weight_decay)` (utils/utils.py create_optimizer); with
the backbone's 24 tensors re-homed in one flat parameter the update is a single element-wise pass (tg_adam_f32)."""
import torch

from . import ops
from ._lib import check, lib


class FlatAdam(torch.optim.Optimizer):
    """Adam (no amsgrad) over flat fp32 device parameters; state keys as torch.optim.Adam ('step', 'exp_avg', 'exp_avg_sq')."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("FlatAdam: contiguous fp32 device parameters only (TGAT.flatten_parameters())")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                check(lib().tg_adam_f32(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                        float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                        int(st["step"]), ops._stream()), "tg_adam_f32")
        return loss

    def native_args(self, p):
        """the update of parameter `p` as a tg_adam_args struct for a native step (flid_amd.stepper: the library issues the kernel behind
        its backward); counts the step as step() does"""
        from ._lib import AdamArgs
        for group in self.param_groups:
            if any(q is p for q in group["params"]):
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                b1, b2 = group["betas"]
                return AdamArgs(st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]), float(b1), float(b2),
                                float(group["eps"]), float(group["weight_decay"]), int(st["step"]))
        raise RuntimeError("FlatAdam.native_args: not a parameter of this optimizer")

    def native_rollback(self, p):
        """the native call that was to issue native_args()'s update failed: the step was not taken"""
        st = self.state.get(p)
        if st and st.get("step", 0) > 0:
            st["step"] -= 1

    def zero_grad(self, set_to_none: bool = True):
        """as torch.optim.Optimizer.zero_grad, without its profiler context (~10 us of host time per step)"""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.requires_grad_(False)
                        p.grad.zero_()

    # torch.optim.Optimizer wraps a subclass's step() in a profiler record_function context unless the function says it is hooked
    # already: that wrapper is ~25 us of host time between the backward's last launch and this kernel -- measured as GPU idle time
    # in front of adam_kernel in every step (the launches before it are short, the host has no lead left there)
    step.hooked = True
