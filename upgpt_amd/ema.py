"""LitEma — shadow weights as buffers named name.replace('.', '') (ldm/modules/ema.py:15-20),
needed to LOAD `model_ema.*` checkpoint keys and honour ema_scope(); the update step
(training) is kept for completeness, it is a handful of torch ops outside the hot path."""
import torch
from torch import nn


class LitEma(nn.Module):
    def __init__(self, model, decay=0.9999, use_num_upates=True):
        super().__init__()
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.m_name2s_name = {}
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0 if use_num_upates else -1, dtype=torch.int))
        for name, p in model.named_parameters():
            if p.requires_grad:
                s_name = name.replace(".", "")  # '.' is not allowed in buffer names
                self.m_name2s_name[name] = s_name
                self.register_buffer(s_name, p.clone().detach().data)
        self.collected_params = []

    def shadow(self, name):
        return getattr(self, self.m_name2s_name[name])

    def forward(self, model):
        decay = self.decay
        if self.num_updates >= 0:
            self.num_updates += 1
            decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.requires_grad:
                    s = self.shadow(name)
                    s.sub_((1.0 - decay) * (s - p))

    def copy_to(self, model):
        # in-place copy_ under no_grad (not `.data.copy_`): it bumps Tensor._version, which is what the packed-weight
        # caches watch (upgpt_amd/params.py weights_fingerprint) — a `.data` write would leave stale fp16 packs
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.requires_grad:
                    p.copy_(self.shadow(name))

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            for c, p in zip(self.collected_params, parameters):
                p.copy_(c)
