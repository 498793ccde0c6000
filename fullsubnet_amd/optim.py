"""Gradient clipping + Adam of the training step on the HIP library.

Mirror of the two calls at recipes/dns_interspeech_2020/fullsubnet/trainer.py:65-69
(`torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad_norm_value)` then
`self.optimizer.step()`, optimizer built at train.py:55-59 as `torch.optim.Adam(lr, betas=(beta1, 0.999))`):
`ClipAdam.step()` runs both in two multi-tensor launches (`fsn_clip_adam_step`).  It subclasses
`torch.optim.Optimizer`, keeps torch.optim.Adam's state layout (`step`, `exp_avg`, `exp_avg_sq`), so
`state_dict()` / `load_state_dict()` interoperate with the reference's checkpoints
(base_trainer.py:134,185 save / restore `optimizer.state_dict()`): a loaded torch.optim.Adam group lacks
`clip_grad_norm_value` (read with a default of 0 = no clipping until train_step sets it) and must have
`weight_decay == 0`, `amsgrad == False`, `maximize == False` - the recipe's settings (train.py:55-59); anything
else is rejected instead of being silently ignored.

The parameters are updated through raw device pointers, which autograd's version counters do not see; `step()`
therefore bumps every updated parameter's version itself, so that the packed-weight caches of the inference
path (keyed on data_ptr / _version) are rebuilt after a training step.
"""
import ctypes

import torch

from . import _lib


class ClipAdam(torch.optim.Optimizer):
    # torch.amp.GradScaler.step() then sets `self.grad_scale` / `self.found_inf` (device tensors) and calls step()
    # directly: the kernel unscales (g / scale before the norm), and skips on a non-finite norm without a host sync
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip_grad_norm_value=0.0):
        if lr <= 0 or eps <= 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("ClipAdam: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, clip_grad_norm_value=clip_grad_norm_value))
        self.total_norm = None  # device scalar: gradient 2-norm before clipping (of the last group stepped)
        # device counter {updates skipped because the gradient norm was not finite, scratch}: the kernel decides and
        # counts without a host sync (GradScaler.step()'s inf-skip, fullsubnet/trainer.py:69); `skipped_steps()` reads it
        self._skipped = {}  # one counter per parameter group (index -> int32[2] on the group's device)

    def skipped_steps(self, group=0):
        """Updates of parameter group `group` skipped so far because the gradient norm was not finite (one host sync)."""
        return int(self._skipped[group][0].item()) if group in self._skipped else 0

    def state_dict(self):
        """torch.optim.Adam's layout; `step` counts the updates APPLIED (skipped ones do not advance Adam's step)."""
        sd = super().state_dict()
        skipped = {gi: self.skipped_steps(gi) for gi in self._skipped}
        if any(skipped.values()):
            sd = {"state": {i: dict(st) for i, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
            for gi, g in enumerate(sd["param_groups"]):
                for i in g["params"]:
                    if i in sd["state"] and skipped.get(gi, 0):
                        sd["state"][i]["step"] = sd["state"][i]["step"] - skipped[gi]
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for c in self._skipped.values():
            c.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise _lib.FsnError("ClipAdam implements the recipe's Adam (train.py:55-59): weight_decay = 0, "
                                    "amsgrad = False, maximize = False")
            if len(ps) > _lib.ADAM_MAX_TENSORS:
                raise _lib.FsnError(f"ClipAdam: at most {_lib.ADAM_MAX_TENSORS} tensors per parameter group")
            step = None
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                k = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
                if step is not None and k != step:
                    raise _lib.FsnError("ClipAdam: parameters of one group must share the step count")
                step = k
            n = len(ps)
            arr = ctypes.c_void_p * n
            P = arr(*[_lib.dev_ptr(p.data, "param").value for p in ps])
            G = arr(*[_lib.dev_ptr(p.grad, "grad").value for p in ps])
            M = arr(*[_lib.dev_ptr(self.state[p]["exp_avg"], "exp_avg").value for p in ps])
            V = arr(*[_lib.dev_ptr(self.state[p]["exp_avg_sq"], "exp_avg_sq").value for p in ps])
            numel = (ctypes.c_size_t * n)(*[p.numel() for p in ps])
            dev = ps[0].device
            cfg = _lib.AdamCfg(group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                               float(group.get("clip_grad_norm_value", 0.0) or 0.0), step)
            ws = _lib.workspace(L.fsn_clip_adam_workspace_bytes(n, numel), dev)
            self.total_norm = torch.empty(1, dtype=torch.float32, device=dev)
            if gi not in self._skipped or self._skipped[gi].device != dev:
                self._skipped[gi] = torch.zeros(2, dtype=torch.int32, device=dev)
            scale = getattr(self, "grad_scale", None)  # set by GradScaler.step for the duration of this call
            if scale is not None:
                scale = scale.to(device=dev, dtype=torch.float32).reshape(1)
            # GradScaler's inf flag covers every group of this optimizer: all of them skip together (scaler.step skips
            # the whole optimizer.step in the reference), not each on its own gradients
            found = getattr(self, "found_inf", None)
            if found is not None:
                found = found.to(device=dev, dtype=torch.float32).reshape(1)
            _lib.check(L.fsn_clip_adam_step(n, P, G, M, V, numel, ctypes.byref(cfg), _lib.dev_ptr(self.total_norm),
                                            _lib.dev_ptr(scale, "grad_scale", allow_none=True),
                                            _lib.dev_ptr(found, "found_inf", allow_none=True),
                                            ctypes.c_void_p(self._skipped[gi].data_ptr()),
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr(dev)))
            torch._C._increment_version(ps)  # the raw-pointer update above is invisible to autograd's counters
        return loss
