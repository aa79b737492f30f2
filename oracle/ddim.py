"""CPU restatement of the DDIM sampler loop (test oracle).

Follows ldm/models/diffusion/ddim.py: sample 56-111, ddim_sampling 113-163,
p_sample_ddim 165-204.  Noise is injected (a [S, B, C, H, W] tensor in loop order) so
that eta > 0 runs are reproducible across devices (SURVEY.md §7 hard part 5): the
reference draws torch.randn on its own device every step, even when sigma = 0
(ddim.py:200).
"""
import numpy as np
import torch

from .schedule import ddim_step_coefficients


@torch.no_grad()
def ddim_sample(eps_fn, alphas_cumprod_f32, shape, S, eta, x_T, noise=None, temperature=1.0,
                cond=None, uncond=None, guidance_scale=1.0, log_every_t=100, num_ddpm=1000,
                mask=None, x0=None, q_sample=None):
    """eps_fn(x, t_long[B], cond) -> eps.  Returns (x_0, intermediates) like ddim.py:163."""
    b = shape[0]
    ts, a, ap, sig, sq1m = ddim_step_coefficients(alphas_cumprod_f32, S, eta, num_ddpm)
    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    total = ts.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:  # ddim.py:144-147
            img = q_sample(x0, t) * mask + (1.0 - mask) * img
        if uncond is None or guidance_scale == 1.0:  # ddim.py:171-172
            e_t = eps_fn(img, t, cond)
        else:  # ddim.py:174-178 (tensor or dict conditioning, batch doubled)
            e_u = eps_fn(img, t, uncond)
            e_c = eps_fn(img, t, cond)
            e_t = e_u + guidance_scale * (e_c - e_u)
        a_t = torch.full((b, 1, 1, 1), float(a[index]))
        a_prev = torch.full((b, 1, 1, 1), float(ap[index]))
        sigma_t = torch.full((b, 1, 1, 1), float(sig[index]))
        sq = torch.full((b, 1, 1, 1), float(sq1m[index]))
        pred_x0 = (img - sq * e_t) / a_t.sqrt()
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        nz = noise[i] if noise is not None else torch.zeros_like(img)
        img = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * nz * temperature
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter
