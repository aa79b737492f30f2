"""PLMSSampler — drop-in for ldm.models.diffusion.plms.PLMSSampler: pseudo
linear multistep (Adams-Bashforth on eps) over the same schedule tables as DDIM (eta must be 0).

Fast path (same conditions as DDIMSampler's): ONE captured HIP graph = UNet body + upk_plms_step_f32 +
upk_advance_step, replayed S + 1 times (the first step evaluates the model twice); the eps history ring,
the Adams-Bashforth combination and classifier-free guidance live in the update kernel, the timestep
embeddings of all evaluations are precomputed like DDIM's.  Anything else (masks, score correctors, noise
dropout) runs on the general path: one UNetModel.forward per model evaluation driven from Python.
"""
import contextlib

import numpy as np
import torch

from .ddim import DDIMSampler
from ._check import require
from ._lib import host_io

# Adams-Bashforth weights on [e_t, e_{t-1}, e_{t-2}, e_{t-3}] by available history (plms.py:224-234)
_AB = {1: (3 / 2, -1 / 2), 2: (23 / 12, -16 / 12, 5 / 12), 3: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}


class PLMSSampler(DDIMSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        super().make_schedule(ddim_num_steps, ddim_discretize, ddim_eta, verbose)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        acp = self.model.alphas_cumprod  # (tables of an unchanged (S, eta) are kept between calls, as in DDIMSampler.sample)
        key = (int(S), float(eta), id(self.model), acp.data_ptr(), int(getattr(acp, "_version", 0)), str(acp.device))
        if verbose or getattr(self, "_sched_key", None) != key:
            with host_io():
                self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
            self._sched_key = key
        C, H, W = shape
        size = (batch_size, C, H, W)
        print(f"Data shape for PLMS sampling is {size}")
        return self.plms_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, noise_dropout=noise_dropout,
                                  temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None):
        if ddim_use_original_steps or quantize_denoised:
            raise NotImplementedError("PLMS with the original 1000 steps / quantized x0 is not on the UPGPT path")
        device = self.model.betas.device
        b = shape[0]
        if timesteps is None:
            timesteps = self.ddim_timesteps
        else:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        if self._fast_ok(cond, ddim_use_original_steps, quantize_denoised, mask, noise_dropout, score_corrector,
                         unconditional_guidance_scale, unconditional_conditioning) \
                and len(timesteps) == len(self.ddim_timesteps) and temperature == 1.:
            return self._fast_plms(cond, shape, x_T, timesteps, callback, img_callback, log_every_t,
                                   unconditional_guidance_scale, unconditional_conditioning)
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device)
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total = timesteps.shape[0]
        print(f"Running PLMS Sampling with {total} timesteps")
        history = []  # newest last, at most 3 entries
        for i, step in enumerate(time_range):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), device=device,
                                 dtype=torch.long)
            if mask is not None:
                require(x0 is not None, "mask given without x0", ValueError)
                img = self.model.q_sample(x0, ts) * mask + (1. - mask) * img
            img, pred_x0, e_t = self.p_sample_plms(img, cond, ts, index=index, temperature=temperature,
                                                   noise_dropout=noise_dropout, score_corrector=score_corrector,
                                                   corrector_kwargs=corrector_kwargs,
                                                   unconditional_guidance_scale=unconditional_guidance_scale,
                                                   unconditional_conditioning=unconditional_conditioning,
                                                   old_eps=history, t_next=ts_next)
            history = (history + [e_t])[-3:]
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    def _fast_plms(self, cond, shape, x_T, timesteps, callback, img_callback, log_every_t, cfg_scale, uc):
        from .ddim import ddim_coefficient_table
        from .engine import SamplerState
        model = self.model
        unet = model.model.diffusion_model
        b, C, H, W = shape
        cfg = uc is not None and cfg_scale != 1.
        if cfg:
            cond = self._cat_cond(uc, cond)
        c_concat, c_cross = model._split_cond(cond)
        S = int(timesteps.shape[0])
        plan = unet.plan(2 * b if cfg else b, H, W, c_cross.shape[1], S + 1, "sampler")  # rows = model evaluations
        dev = plan.dev
        with torch.cuda.device(dev):
            attr = "_plms_state_cfg" if cfg else "_plms_state"
            st = getattr(plan, attr, None)
            if st is None:
                st = SamplerState(plan, C, cfg=cfg, plms=True)
                setattr(plan, attr, st)
            order = np.arange(S)[::-1].copy()
            t_desc = np.asarray(timesteps)[order].astype(np.float32)
            # evaluation k runs at: t_0, then (x~, t_next = t_1) for the Euler corrector, then t_1, t_2, ...
            t_eval = np.concatenate([t_desc[:1], t_desc[min(1, S - 1):min(1, S - 1) + 1], t_desc[1:]])
            require(t_eval.shape[0] == S + 1, "PLMS evaluates S + 1 times", RuntimeError)
            f64b = lambda v: np.asarray(v, dtype=np.float64).tobytes()
            rkey = ("plms", t_eval.tobytes())  # (the rows belong to the plan, the coefficients to this state: ddim.py)
            ckey = (f64b(self.ddim_alphas), f64b(self.ddim_alphas_prev), f64b(self.ddim_sigmas))
            fresh_rows = getattr(plan, "_t_rows_key", None) != rkey
            fresh_coefs = getattr(st, "_coef_key", None) != ckey
            fresh = fresh_rows or fresh_coefs
            on_host = any(t is not None and torch.is_tensor(t) and not t.is_cuda for t in (x_T, c_concat, c_cross))
            with (host_io() if (fresh or on_host) else contextlib.nullcontext()):  # (ddim.py _fast_sampling: same rule)
                img = torch.randn(shape, device=dev) if x_T is None else x_T.to(dev, torch.float32)
                st.x.copy_(img)
                plan.load_x_nchw(torch.cat([st.x, st.x]) if cfg else st.x, 0, 0)
                if c_concat is not None:
                    plan.load_x_nchw(c_concat, C, plan.cin_pad)
                require(C + (0 if c_concat is None else c_concat.shape[1]) == unet.in_channels, "latent + concat channels != UNet in_channels", ValueError)
                plan.load_context(c_cross)
                if fresh_rows:
                    plan.t_rows.copy_(torch.as_tensor(t_eval))
                    plan._t_rows_key = rkey
                if fresh_coefs:
                    st.coefs[:S].copy_(ddim_coefficient_table(self.ddim_alphas, self.ddim_alphas_prev, self.ddim_sigmas,
                                                              self.ddim_sqrt_one_minus_alphas, order))
                    st._coef_key = ckey
                for _ in range(S + 1):  # the reference draws (and, eta being 0, discards) one noise tensor per update:
                    torch.randn(shape, device=dev)  # plms.py get_x_prev_and_pred_x0 — same generator state afterwards
                plan.step.zero_()
                plan.prep.run()
            intermediates = {"x_inter": [st.x.clone()], "pred_x0": [st.x.clone()]}
            print(f"Running PLMS Sampling with {S} timesteps")
            for k in range(S + 1):
                st.launch(False, cfg_scale)
                if k == 0:
                    continue  # predictor evaluation of the first step
                i = k - 1
                index = S - i - 1
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(st.pred_x0.clone(), i)
                if index % log_every_t == 0 or index == S - 1:
                    intermediates["x_inter"].append(st.x.clone())
                    intermediates["pred_x0"].append(st.pred_x0.clone())
            return st.x.clone(), intermediates

    @torch.no_grad()
    def p_sample_plms(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, old_eps=None, t_next=None):
        def model_eps(xx, tt):
            if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
                e = self.model.apply_model(xx, tt, c)
            else:
                e_u, e = self.model.apply_model(torch.cat([xx] * 2), torch.cat([tt] * 2),
                                                self._cat_cond(unconditional_conditioning, c)).chunk(2)
                e = e_u + unconditional_guidance_scale * (e - e_u)
            if score_corrector is not None:
                require(self.model.parameterization == "eps", "classifier-free guidance / score correction needs an eps-parameterised model", NotImplementedError)
                e = score_corrector.modify_score(self.model, e, xx, tt, c, **corrector_kwargs)
            return e

        def update(e):  # the DDIM update with sigma = 0 (eta is forced to 0), fused kernel
            return self._ddim_update(x, e, index, temperature=temperature, noise_dropout=noise_dropout,
                                     repeat_noise=repeat_noise)

        old_eps = old_eps or []
        e_t = model_eps(x, t)
        if len(old_eps) == 0:  # pseudo improved Euler: one extra evaluation at (x_prev, t_next)
            x_prev, _ = update(e_t)
            e_prime = (e_t + model_eps(x_prev, t_next)) / 2
        else:
            w = _AB[min(len(old_eps), 3)]
            e_prime = w[0] * e_t
            for k in range(1, len(w)):
                e_prime = e_prime + w[k] * old_eps[-k]
        x_prev, pred_x0 = update(e_prime)
        return x_prev, pred_x0, e_t
