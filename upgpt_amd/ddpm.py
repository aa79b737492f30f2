"""DDPM / LatentDiffusion / DiffusionWrapper — the inference call surface of
ldm/models/diffusion/ddpm.py.

What is CONTRACT here (and therefore kept name for name): the constructor keyword arguments, so that the `model:`
block of configs/deepfashion/bbox.yaml instantiates unchanged; the module / buffer names that make reference
checkpoints load (model.diffusion_model.*, model_ema.*, first_stage_model.*, cond_stage_model.*, extra_cond_models.*,
the schedule buffers); the methods the named callers use (apply_model, decode_first_stage, get_learned_conditioning,
ema_scope, q_sample, sample_log, log_images, get_input) with their argument order and return shapes.  Everything
else is this package's own: the schedule buffers come out of one table, conditioning assembly lives in one place
(`_conditioning`), EMA evaluation packs the shadow weights instead of copying them over the live ones, and the
training-side state of the reference (loss weights, log-variance, ELBO terms, LR scheduler config) does not exist —
those keyword arguments are accepted and ignored, the training entry points raise.

Plain torch.nn.Modules (no pytorch_lightning); every FLOP of the denoiser / first stage runs in the HIP engine.
"""
import threading
from contextlib import contextmanager

import numpy as np
import torch
from torch import nn

from .config import count_params, instantiate_from_config, to_plain
from .ddim import DDIMSampler
from .ema import LitEma
from .schedule import extract_into_tensor, make_beta_schedule
from ._check import require


_EMA_LOCK = threading.Lock()


def disabled_train(self, mode=True):
    return self


class DiffusionWrapper(nn.Module):
    """ddpm.py:1550-1577."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        require(self.conditioning_key in [None, "concat", "crossattn", "hybrid", "adm"], "unknown conditioning_key %r" % (self.conditioning_key,), ValueError)

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None):
        key = self.conditioning_key
        if key is None:
            return self.diffusion_model(x, t)
        if key == "concat":
            return self.diffusion_model(torch.cat([x] + c_concat, dim=1), t)
        if key == "crossattn":
            return self.diffusion_model(x, t, context=torch.cat(c_crossattn, 1))
        if key == "hybrid":
            # c_crossattn is a TENSOR in the hybrid branch (ddpm.py:1569): a None c_concat
            # raises TypeError exactly like the reference
            return self.diffusion_model(torch.cat([x] + c_concat, dim=1), t, context=torch.cat([c_crossattn], 1))
        if key == "adm":
            return self.diffusion_model(x, t, y=c_crossattn[0])
        raise NotImplementedError()


# Persistent schedule buffers (checkpoint keys, ddpm.py:125-146) as functions of (betas, alphas_cumprod,
# alphas_cumprod_prev), evaluated in float64 and stored as fp32 like the reference does.
_SCHEDULE = (
    ("betas", lambda b, a, p: b),
    ("alphas_cumprod", lambda b, a, p: a),
    ("alphas_cumprod_prev", lambda b, a, p: p),
    ("sqrt_alphas_cumprod", lambda b, a, p: np.sqrt(a)),
    ("sqrt_one_minus_alphas_cumprod", lambda b, a, p: np.sqrt(1.0 - a)),
    ("log_one_minus_alphas_cumprod", lambda b, a, p: np.log(1.0 - a)),
    ("sqrt_recip_alphas_cumprod", lambda b, a, p: np.sqrt(1.0 / a)),
    ("sqrt_recipm1_alphas_cumprod", lambda b, a, p: np.sqrt(1.0 / a - 1)),
    ("posterior_variance", lambda b, a, p: b * (1.0 - p) / (1.0 - a)),
    ("posterior_log_variance_clipped", lambda b, a, p: np.log(np.maximum(b * (1.0 - p) / (1.0 - a), 1e-20))),
    ("posterior_mean_coef1", lambda b, a, p: b * np.sqrt(p) / (1.0 - a)),
    ("posterior_mean_coef2", lambda b, a, p: (1.0 - p) * np.sqrt(1.0 - b) / (1.0 - a)),
)


def _load_weights(module, path, drop_prefixes=(), what="checkpoint"):
    """Reads a (Lightning) checkpoint, drops keys by prefix, loads non-strictly and reports what did not match.
    The files are trusted pickles with hparams / callback objects inside, hence weights_only=False."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    state = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    kept = {}
    for name, tensor in state.items():
        if name.startswith(tuple(drop_prefixes)) and drop_prefixes:
            print("Deleting key {} from state_dict.".format(name))
        else:
            kept[name] = tensor
    result = module.load_state_dict(kept, strict=False)
    print(f"Restored {what} from {path} with {len(result.missing_keys)} missing and "
          f"{len(result.unexpected_keys)} unexpected keys")
    for label, keys in (("Missing", result.missing_keys), ("Unexpected", result.unexpected_keys)):
        if keys:
            print(f"{label} Keys: {list(keys)}")
    return result


class DDPM(nn.Module):
    """ddpm.py:50-210: denoiser wrapper + EMA shadow + noise schedule."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=[], load_only_unet=False, monitor="val/loss", use_ema=True, first_stage_key="image",
                 image_size=256, crop_size=[256, 176], channels=3, log_every_t=100, clip_denoised=True,
                 linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, given_betas=None, original_elbo_weight=0.,
                 v_posterior=0., l_simple_weight=1., conditioning_key=None, parameterization="eps",
                 scheduler_config=None, use_positional_encodings=False, learn_logvar=False, logvar_init=0.):
        super().__init__()
        if parameterization not in ("eps", "x0"):
            raise AssertionError('currently only supporting "eps" and "x0"')
        if v_posterior:
            raise NotImplementedError("v_posterior != 0 changes the posterior variance buffers; not used by UPGPT")
        # loss_type, monitor, log_every_t, original_elbo_weight, l_simple_weight, scheduler_config, learn_logvar and
        # logvar_init configure the training loop of the reference and do nothing here; the plain attributes its
        # scripts read (main.py:653 model.monitor, classifier.py:204 diffusion_model.log_every_t) are kept
        self.loss_type, self.monitor, self.log_every_t = loss_type, monitor, log_every_t
        self.v_posterior, self.original_elbo_weight, self.l_simple_weight = v_posterior, original_elbo_weight, l_simple_weight
        self.use_scheduler, self.learn_logvar = scheduler_config is not None, learn_logvar
        if self.use_scheduler:
            self.scheduler_config = scheduler_config
        self.parameterization, self.first_stage_key = parameterization, first_stage_key
        self.channels, self.crop_size = channels, crop_size
        self.clip_denoised, self.use_positional_encodings = clip_denoised, use_positional_encodings
        self.image_size = [int(v) for v in image_size] if np.ndim(image_size) else [int(image_size)] * 2
        self.cond_stage_model, self.use_ema = None, bool(use_ema)
        print(f"{type(self).__name__}: Running in {parameterization}-prediction mode")
        self.model = denoiser = DiffusionWrapper(unet_config, conditioning_key)
        count_params(denoiser, verbose=True)
        if use_ema:
            self.model_ema = LitEma(denoiser)  # shadow buffers `model_ema.*` of the checkpoints
            print(f"Keeping EMAs of {sum(1 for _ in self.model_ema.buffers())}.")
        self.register_schedule(given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s)
        if ckpt_path is not None and type(self) is DDPM:
            self.init_from_ckpt(ckpt_path, ignore_keys, only_model=load_only_unet)

    @property
    def device(self):
        return next(self.parameters()).device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = np.asarray(given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s),
            dtype=np.float64)
        acp = np.cumprod(1.0 - betas, axis=0)
        acp_prev = np.concatenate([[1.0], acp[:-1]])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        for name, fn in _SCHEDULE:
            self.register_buffer(name, torch.tensor(fn(betas, acp, acp_prev), dtype=torch.float32))

    @contextmanager
    def ema_scope(self, context=None):
        """ddpm.py:179-192.  Inside the scope the denoiser COMPUTES with the LitEma shadow
        weights; they are packed straight from the shadow buffers instead of being copied
        over the live parameters (same results, no 1.7 GB copy + repack per call)."""
        if not self.use_ema:
            yield None
            return
        # the scope nests and may be entered from several host threads at once (one per execution lane): the override
        # goes in with the first entrant and out with the last
        unet, ema = self.model.diffusion_model, self.model_ema
        with _EMA_LOCK:
            n = self.__dict__.get("_ema_depth", 0)
            if n == 0:
                unet.set_weight_override("ema", lambda n: ema.shadow("diffusion_model." + n).data,
                                         lambda: (sum(b._version for b in ema.buffers()), ema.decay.data_ptr()))
            self.__dict__["_ema_depth"] = n + 1
        if context is not None:
            print(f"{context}: Switched to EMA weights")
        try:
            yield None
        finally:
            with _EMA_LOCK:
                n = self.__dict__["_ema_depth"] = self.__dict__["_ema_depth"] - 1
                if n == 0:
                    unet.set_weight_override(None)
            if context is not None:
                print(f"{context}: Restored training weights")

    def init_from_ckpt(self, path, ignore_keys=list(), only_model=False):
        """ddpm.py:194-210: the whole module, or only the denoiser wrapper."""
        _load_weights(self.model if only_model else self, path, tuple(ignore_keys))

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:271-274: sqrt(a_t) x0 + sqrt(1 - a_t) noise."""
        if noise is None:
            noise = torch.randn_like(x_start)
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def get_input(self, batch, k):
        """ddpm.py:331-338: NHWC (or HW-only) float image -> contiguous NCHW fp32."""
        x = batch[k]
        if x.dim() == 3:
            x = x.unsqueeze(-1)
        return x.movedim(-1, 1).contiguous().float()

    def training_step(self, *a, **k):
        raise NotImplementedError("training is out of scope of upgpt_amd (inference hot path only)")

    p_losses = configure_optimizers = validation_step = training_step


class LatentDiffusion(DDPM):
    """ddpm.py:433-1547, inference surface."""

    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="image",
                 cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None, conditioning_key=None,
                 scale_factor=1.0, scale_by_std=False, concat_key=None, *args, **kwargs):
        if (num_timesteps_cond or 1) > 1:
            raise NotImplementedError("num_timesteps_cond > 1 (shortened cond schedule) is not on the UPGPT path")
        self.num_timesteps_cond = 1
        require(self.num_timesteps_cond <= kwargs["timesteps"], "num_timesteps_cond > timesteps", ValueError)
        opts = {k: to_plain(v) for k, v in kwargs.items()}
        ckpt_path = opts.pop("ckpt_path", None)
        ignore_keys = opts.pop("ignore_keys", [])
        extra = to_plain(opts.pop("extra_cond_stages", None)) or {}
        key2 = opts.pop("cond_stage_key_2", None)
        first_stage_config, cond_stage_config = to_plain(first_stage_config), to_plain(cond_stage_config)
        if cond_stage_config == "__is_unconditional__":
            conditioning_key = None
        elif conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        super().__init__(conditioning_key=conditioning_key, *args, **opts)
        self.cond_stage_key, self.cond_stage_key_2 = cond_stage_key, key2
        self.cond_stage_trainable, self.cond_stage_forward = cond_stage_trainable, cond_stage_forward
        self.concat_mode, self.concat_key = concat_mode, concat_key
        self.scale_by_std = scale_by_std
        self.clip_denoised = False
        self.bbox_tokenizer = None  # (ddpm.py:489; only the patch-split first stage, out of scope, would set it)
        try:  # (ddpm.py:476-479)
            self.num_downs = len(first_stage_config["params"]["ddconfig"]["ch_mult"]) - 1
        except (KeyError, TypeError):
            self.num_downs = 0
        if scale_by_std:  # (a buffer then: it is a checkpoint key)
            self.register_buffer("scale_factor", torch.tensor(scale_factor))
        if not scale_by_std:
            self.scale_factor = scale_factor
        # extra conditioning encoders (SMPL / pose projections): module list + the batch key each one reads
        self.extra_cond_keys = [cfg["cond_stage_key"] for cfg in extra.values()]
        self.extra_cond_models = nn.ModuleList(instantiate_from_config(cfg) for cfg in extra.values()) if extra else []
        for build, cfg in ((self.instantiate_first_stage, first_stage_config),
                           (self.instantiate_cond_stage, cond_stage_config)):
            build(cfg)
        self.restarted_from_ckpt = ckpt_path is not None
        if self.restarted_from_ckpt:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    # ---- construction
    @staticmethod
    def _frozen(module, freeze_params):
        module = module.eval()
        module.train = disabled_train
        if freeze_params:
            for p in module.parameters():
                p.requires_grad = False
        return module

    def instantiate_first_stage(self, config):
        self.first_stage_model = self._frozen(instantiate_from_config(config), False)

    def instantiate_cond_stage(self, config):
        """ddpm.py:531-549: a config node, or one of the two sentinels."""
        if config == "__is_unconditional__":
            print(f"Training {type(self).__name__} as an unconditional model.")
            self.cond_stage_model = None
        elif config == "__is_first_stage__":
            print("Using first stage also as cond stage.")
            self.cond_stage_model = self.first_stage_model
        elif self.cond_stage_trainable:
            self.cond_stage_model = instantiate_from_config(config).eval()
        else:
            self.cond_stage_model = self._frozen(instantiate_from_config(config), True)

    # ---- conditioning
    def get_learned_conditioning(self, c):
        """ddpm.py:577-592: run the conditioning stage on `c` — a named method when cond_stage_forward is set,
        else .encode() when the stage has one (a returned distribution is replaced by its mode), else a plain call
        (keyword call for dict input)."""
        stage = self.cond_stage_model
        if self.cond_stage_forward is not None:
            return getattr(stage, self.cond_stage_forward)(c)
        encode = getattr(stage, "encode", None)
        if callable(encode):
            out = encode(c)
            mode = getattr(out, "mode", None)
            return mode() if callable(mode) and not torch.is_tensor(out) else out
        return stage(**c) if isinstance(c, dict) else stage(c)

    def _as_cond_dict(self, cond):
        """tensor / list / dict conditioning -> {'c_concat' | 'c_crossattn': ...} (ddpm.py:962-966)."""
        if isinstance(cond, dict):
            return cond
        slot = "c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"
        return {slot: cond if isinstance(cond, list) else [cond]}

    def _split_cond(self, cond):
        """Any conditioning the callers pass -> (c_concat tensor | None, c_crossattn tensor) for the fused sampler
        path — the rules of apply_model + DiffusionWrapper (ddpm.py:962-971, 1557-1570)."""
        key = self.model.conditioning_key
        cond = self._as_cond_dict(cond)
        cc, ca = cond.get("c_concat"), cond.get("c_crossattn")
        if key == "hybrid":
            if cc is None or any(v is None for v in cc):
                raise TypeError('can only concatenate list (not "NoneType") to list')  # as the reference
            return torch.cat(list(cc), 1), (ca if torch.is_tensor(ca) else torch.cat(ca, 1))
        if key == "crossattn":
            return None, (ca if torch.is_tensor(ca) else torch.cat(ca, 1))
        raise NotImplementedError("fused sampler path for conditioning_key=%r" % key)

    # ---- first stage
    def get_first_stage_encoding(self, encoder_posterior):
        """ddpm.py:566-575: a posterior is sampled, a tensor is taken as is; both scaled by scale_factor."""
        if torch.is_tensor(encoder_posterior):
            z = encoder_posterior
        elif callable(getattr(encoder_posterior, "sample", None)):
            z = encoder_posterior.sample()
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    @torch.no_grad()
    def encode_first_stage(self, x):
        if hasattr(self, "split_input_params"):
            raise NotImplementedError("patch-split first stage (split_input_params) is not used by UPGPT configs")
        return self.first_stage_model.encode(x)

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:771-829, plain branch: (1/scale_factor) * z -> first_stage_model.decode."""
        if predict_cids or hasattr(self, "split_input_params"):
            raise NotImplementedError("predict_cids / split_input_params are not used by UPGPT configs")
        sf = float(self.scale_factor)
        if hasattr(self.first_stage_model, "_decode_plan"):
            return self.first_stage_model.decode(z, scale_factor=sf)  # scaling fused into the input kernel
        return self.first_stage_model.decode(1. / sf * z)

    # ---- denoiser
    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """ddpm.py:962-966, 1057-1063 (non-split branch)."""
        if hasattr(self, "split_input_params"):
            raise NotImplementedError("split_input_params")
        out = self.model(x_noisy, t, **self._as_cond_dict(cond))
        return out[0] if isinstance(out, tuple) and not return_ids else out

    # ---- sampling / logging
    def _conditioning(self, batch, x, cond_key, force_c_encode, bs):
        """(c, original conditioning input): the cross-attention sequence of ddpm.py:718-752 — the main stage on
        batch[cond_key] (text, optionally a {key, key_2} pair), then every extra stage's tokens appended along the
        token axis (styles | smpl for bbox.yaml)."""
        dev = self.device
        on_dev = lambda v: v.to(dev) if torch.is_tensor(v) else v
        cond_key = cond_key or self.cond_stage_key
        if cond_key == self.first_stage_key:
            xc = x
        elif cond_key == "class_label":
            xc = batch
        elif cond_key in ("caption", "coordinates_bbox", "txt"):
            xc = batch[cond_key]
            if self.cond_stage_key_2:
                xc = {cond_key: xc, self.cond_stage_key_2: on_dev(batch[self.cond_stage_key_2])}
        else:
            xc = DDPM.get_input(self, batch, cond_key).to(dev)
        if self.cond_stage_trainable and not force_c_encode:
            c = self.cond_stage_model(**xc)
        else:
            c = self.get_learned_conditioning(on_dev(xc))
        for key, stage in zip(self.extra_cond_keys, self.extra_cond_models):
            c = torch.cat((c, stage.forward(on_dev(batch.get(key)))), 1)
        return (c if bs is None else c[:bs]), xc

    def get_input(self, batch, k, return_first_stage_outputs=False, force_c_encode=False, cond_key=None,
                  return_original_cond=False, bs=None, return_loss_w=False, encode_image=True):
        """ddpm.py:684-769 -> [z, {'c_crossattn': c, 'c_concat': [mask]}, (x, xrec)?, (xc)?, (loss_w)?].
        The image is encoded only when an encoder is available."""
        dev = self.device
        head = (lambda v: v) if bs is None else (lambda v: v[:bs])
        x = head(DDPM.get_input(self, batch, k)).to(dev)
        z = None
        if encode_image:
            try:
                z = self.get_first_stage_encoding(self.encode_first_stage(x)).detach()
            except NotImplementedError:
                pass
        c = xc = mask = None
        if self.model.conditioning_key is not None:
            if self.concat_key:
                mask = head(batch[self.concat_key]).to(dev)
            c, xc = self._conditioning(batch, x, cond_key, force_c_encode, bs)
        out = [z, {"c_crossattn": c, "c_concat": [mask]}]
        if return_first_stage_outputs:
            out += [x, None if z is None else self.decode_first_stage(z)]
        if return_original_cond:
            out.append(xc)
        if return_loss_w:
            out.append(batch.get("loss_w", None))
        return out

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        """ddpm.py:1312-1325."""
        if not ddim:
            raise NotImplementedError("DDPM ancestral sampling (ddim=False) is not used by the UPGPT callers")
        shape = (self.channels, *self.image_size)
        return DDIMSampler(self).sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)

    @torch.no_grad()
    def log_images(self, batch, N=8, n_row=4, sample=True, ddim_steps=200, ddim_eta=1., return_keys=None,
                   quantize_denoised=False, inpaint=False, plot_denoise_rows=False, plot_progressive_rows=False,
                   plot_diffusion_rows=False, seed=None, **kwargs):
        """ddpm.py:1380-1499 — the path InferenceModel.generate drives (generate_utils.py:159-169):
        conditioning assembly -> EMA scope -> DDIM -> decode.  Returns {'reconstruction'?, 'samples'}."""
        if inpaint or plot_denoise_rows or plot_progressive_rows or plot_diffusion_rows or quantize_denoised:
            raise NotImplementedError("only the sampling branch of log_images is implemented")
        if ddim_steps is None:
            raise NotImplementedError("ddim_steps=None (DDPM sampler)")
        _, cond, x, xrec, _ = self.get_input(batch, self.first_stage_key, return_first_stage_outputs=True,
                                             force_c_encode=True, return_original_cond=True, bs=N)
        n = min(x.shape[0], N)
        log = {} if xrec is None else {"reconstruction": xrec}
        if sample:
            x_T = None
            if seed:  # one seeded latent shared by the batch (ddpm.py:1422-1426)
                torch.manual_seed(seed)
                x_T = torch.randn((1, self.channels, *self.image_size), device=self.device).repeat(n, 1, 1, 1)
            with self.ema_scope("Plotting"):
                z, _ = self.sample_log(cond=cond, batch_size=n, ddim=True, ddim_steps=ddim_steps, eta=ddim_eta,
                                       x_T=x_T, **kwargs)
            log["samples"] = self.decode_first_stage(z)
        if return_keys and any(k in log for k in return_keys):
            return {k: log[k] for k in return_keys}
        return log
