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


@torch.no_grad()
def ddim_decode(eps_fn, alphas_cumprod_f32, x_latent, S, t_start, cond, uncond=None, guidance_scale=1.0,
                num_ddpm=1000):
    """DDIMSampler.decode (ddim.py:222-241): the last t_start steps of an S-step eta = 0 schedule applied to
    x_latent (the img2img entry: stochastic_encode -> decode)."""
    b = x_latent.shape[0]
    ts, a, ap, sig, sq1m = ddim_step_coefficients(alphas_cumprod_f32, S, 0.0, num_ddpm)
    steps = np.flip(ts[:t_start])
    total = steps.shape[0]
    x = x_latent
    for i, step in enumerate(steps):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if uncond is None or guidance_scale == 1.0:
            e_t = eps_fn(x, t, cond)
        else:
            e_u = eps_fn(x, t, uncond)
            e_t = e_u + guidance_scale * (eps_fn(x, t, cond) - e_u)
        pred_x0 = (x - float(sq1m[index]) * e_t) / float(np.sqrt(np.float32(a[index])))
        x = float(np.sqrt(np.float32(ap[index]))) * pred_x0 + float(np.sqrt(np.float32(1.0 - ap[index]))) * e_t
    return x


@torch.no_grad()
def plms_sample(eps_fn, alphas_cumprod_f32, shape, S, x_T, cond=None, log_every_t=100, num_ddpm=1000):
    """ldm/models/diffusion/plms.py:114-236 (eta = 0): the DDIM update applied to an
    Adams-Bashforth combination of the current and up to three previous eps; the first step
    is a pseudo improved Euler step with one extra model evaluation at (x_prev, t_next)."""
    b = shape[0]
    ts, a, ap, sig, sq1m = ddim_step_coefficients(alphas_cumprod_f32, S, 0.0, num_ddpm)
    time_range = np.flip(ts)
    total = ts.shape[0]
    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    old = []

    def update(x, e, index):
        a_t = torch.full((b, 1, 1, 1), float(a[index]))
        a_prev = torch.full((b, 1, 1, 1), float(ap[index]))
        sq = torch.full((b, 1, 1, 1), float(sq1m[index]))
        pred_x0 = (x - sq * e) / a_t.sqrt()
        return a_prev.sqrt() * pred_x0 + (1.0 - a_prev).sqrt() * e, pred_x0

    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        t_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        e_t = eps_fn(img, t, cond)
        if len(old) == 0:
            x_prev, _ = update(img, e_t, index)
            e_p = (e_t + eps_fn(x_prev, t_next, cond)) / 2
        elif len(old) == 1:
            e_p = (3 * e_t - old[-1]) / 2
        elif len(old) == 2:
            e_p = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            e_p = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        img, pred_x0 = update(img, e_p, index)
        old = (old + [e_t])[-3:]
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter
