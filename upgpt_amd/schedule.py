"""Host-side noise schedule tables (once per model / once per sample() call).

Numerics follow the reference exactly because the tables are tiny and precision-critical:
betas / alphas_cumprod in float64 numpy then stored fp32 (util.py:21-43, ddpm.py:125-146);
DDIM selection and sigmas per util.py:46-74.  The per-step scalars that the reference
re-materialises with torch.full every step (ddim.py:189-192) are folded here into ONE
device table of 4 fp32 coefficients per step for upk_ddim_step_f32.
"""
import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """Only "linear" — what bbox.yaml, the upscale config and every checkpoint of the path use (ddpm.py:85 default).  The
    reference's cosine / sqrt_linear / sqrt variants (util.py:29-40) serve no config on the path and are refused loudly."""
    if schedule != "linear":
        raise NotImplementedError("beta schedule %r is not on the UPGPT inference path (only 'linear' is)" % (schedule,))
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.arange(0, num_ddpm_timesteps, stride)
    elif ddim_discr_method == "quad":
        steps = (np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps = steps + 1  # the +1 gets the final alpha values right (util.py:56-57)
    if verbose:
        print("ddim timesteps:", steps)
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphacums: fp32 tensor (the model buffer).  Returns (sigmas f64 tensor, alphas fp32
    tensor, alphas_prev f64 ndarray) — the same types the reference hands back."""
    acp = torch.as_tensor(alphacums).detach().cpu().float()
    idx = torch.as_tensor(np.asarray(ddim_timesteps), dtype=torch.long)
    alphas = acp[idx]
    alphas_prev = np.asarray([float(acp[0])] + acp[idx[:-1]].tolist())
    ap = torch.from_numpy(alphas_prev)
    # fp32 reciprocal first, as torch's mixed ndarray/tensor dispatch does in the reference
    ratio = (1 - alphas).reciprocal().double() * (1 - ap)
    sigmas = eta * torch.sqrt(ratio * (1 - alphas.double() / ap))
    if verbose:
        print("ddim alphas:", alphas, "alphas_prev:", alphas_prev, "sigmas (eta %s):" % eta, sigmas)
    return sigmas, alphas, alphas_prev


def ddim_coefficient_table(alphas, alphas_prev, sigmas, sqrt_one_minus_alphas, order):
    """[len(order), 4] fp32: {sqrt(1-a_t), 1/sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2)}
    for DDIM indices `order` (loop order = descending index).  Scalars are rounded to
    fp32 first, like torch.full((b,1,1,1), value) does in ddim.py:189-192."""
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64)).float()
    a, ap, sg, sq = f32(alphas), f32(alphas_prev), f32(sigmas), f32(sqrt_one_minus_alphas)
    idx = torch.as_tensor(np.asarray(order), dtype=torch.long)
    a, ap, sg, sq = a[idx], ap[idx], sg[idx], sq[idx]
    return torch.stack([sq, 1.0 / a.sqrt(), ap.sqrt(), (1.0 - ap - sg ** 2).sqrt()], dim=1).contiguous()


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))
