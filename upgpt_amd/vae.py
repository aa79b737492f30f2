"""AutoencoderKL — drop-in for ldm.models.autoencoder.AutoencoderKL (ctor 286-306,
encode 324-328, decode 330-333) with the Encoder / Decoder (model.py:368-568) running on
libupk.so: decode is on the images/sec path, encode is the first "next" row of SURVEY.md
§8f (reconstruction output of log_images, img2img entry)."""
import os

import torch

from .arch import VAEArch
from .config import instantiate_from_config
from .params import ParamTree, weights_fingerprint
from ._check import require
from ._lib import host_io


class DiagonalGaussianDistribution(object):
    """Posterior q(z|x) from the encoder's moments (ldm/modules/distributions/distributions.py:24-65):
    mean | logvar split along channels, logvar clamped to [-30, 20].  A handful of element-wise
    ops on a [B, 2z, h, w] tensor once per batch — plain tensor arithmetic, not a kernel."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, device=self.parameters.device)
        return self.mean + self.std * noise.to(self.mean.device)

    def mode(self):
        return self.mean

    def kl(self, other=None):
        if self.deterministic:
            return torch.zeros(1)
        if other is None:
            return 0.5 * torch.sum(self.mean ** 2 + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum((self.mean - other.mean) ** 2 / other.var + self.var / other.var - 1.0 -
                               self.logvar + other.logvar, dim=[1, 2, 3])


class AutoencoderKL(ParamTree):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=None, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        require(ddconfig["double_z"], "AutoencoderKL needs ddconfig.double_z", NotImplementedError)
        self.image_key = image_key
        self.arch = VAEArch(ddconfig, embed_dim)
        self.embed_dim = embed_dim
        self.add_params(self.arch.param_shapes())
        try:  # training-only object; kept when cheap (bbox.yaml uses torch.nn.Identity)
            self.loss = instantiate_from_config(lossconfig) if lossconfig else None
        except Exception:
            self.loss = None
        if colorize_nlabels is not None:
            require(type(colorize_nlabels) == int, "colorize_nlabels must be an int", TypeError)
            self.register_buffer("colorize", torch.randn(3, colorize_nlabels, 1, 1))
        if monitor is not None:
            self.monitor = monitor
        self._packed = None
        self._plans = {}
        self._packed_enc = None
        self._enc_plans = {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]  # (trusted Lightning pickle)
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def _decode_plan(self, B, h, w, scale_factor):
        from ._lib import PLAN_LOCK
        with PLAN_LOCK:
            return self._decode_plan_locked(B, h, w, scale_factor)

    def _decode_plan_locked(self, B, h, w, scale_factor):
        from ._lib import get_context
        from .engine import PackedVAEDecoder, VAEDecodePlan
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("upgpt_amd.AutoencoderKL.decode runs only on the MI355X HIP path (parameters are on "
                               "%s); there is no CPU fallback" % p.device)
        ctx = get_context(p.device)
        fp = weights_fingerprint(self)
        if self._packed is None or self._packed[0] != fp:
            params = dict(self.named_parameters())
            with torch.cuda.device(p.device), host_io():
                self._packed = (fp, PackedVAEDecoder(ctx, self.arch, lambda n: params[n].data))
                torch.cuda.current_stream(p.device).synchronize()  # (packed on this lane's stream, read from every lane's)
            self._plans = {}
        from ._lib import concurrency, current_lane
        # (the tuning table the plan is built from is part of its identity, as in UNetModel.plan)
        key = (B, h, w, float(scale_factor), concurrency() > 1, current_lane())
        if key not in self._plans:
            mine = [k for k in self._plans if k[-1] == key[-1]]
            if len(mine) >= 4:  # (per lane: another lane's plans may be executing)
                self._plans.pop(mine[0])
            with torch.cuda.device(p.device), host_io():
                self._plans[key] = VAEDecodePlan(ctx, self._packed[1], B, h, w, scale_factor)
                self._plans[key].apply_tuning(tune_missing=os.environ.get("UPGPT_AUTOTUNE", "0") == "1")
        return self._plans[key]

    @torch.no_grad()
    def decode(self, z, scale_factor=1.0):
        """z [B, embed_dim, h, w] -> image [B, out_ch, h*f, w*f] fp32.  `scale_factor`
        (LatentDiffusion's z / 0.18215, ddpm.py:779) is folded into the input conversion."""
        B, c, h, w = z.shape
        pl = self._decode_plan(B, h, w, scale_factor)
        with torch.cuda.device(pl.dev):
            return pl.run(z).clone()

    def _encode_plan(self, B, H, W):
        from ._lib import PLAN_LOCK
        with PLAN_LOCK:
            return self._encode_plan_locked(B, H, W)

    def _encode_plan_locked(self, B, H, W):
        from ._lib import get_context
        from .engine import PackedVAEEncoder, VAEEncodePlan
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("upgpt_amd.AutoencoderKL.encode runs only on the MI355X HIP path (parameters are on "
                               "%s); there is no CPU fallback" % p.device)
        ctx = get_context(p.device)
        fp = weights_fingerprint(self)
        if self._packed_enc is None or self._packed_enc[0] != fp:
            params = dict(self.named_parameters())
            with torch.cuda.device(p.device), host_io():
                self._packed_enc = (fp, PackedVAEEncoder(ctx, self.arch, lambda n: params[n].data))
                torch.cuda.current_stream(p.device).synchronize()
            self._enc_plans = {}
        from ._lib import concurrency, current_lane
        key = (B, H, W, concurrency() > 1, current_lane())
        if key not in self._enc_plans:
            mine = [k for k in self._enc_plans if k[-1] == key[-1]]
            if len(mine) >= 4:
                self._enc_plans.pop(mine[0])
            with torch.cuda.device(p.device), host_io():
                self._enc_plans[key] = VAEEncodePlan(ctx, self._packed_enc[1], B, H, W)
                self._enc_plans[key].apply_tuning(tune_missing=os.environ.get("UPGPT_AUTOTUNE", "0") == "1")
        return self._enc_plans[key]

    @torch.no_grad()
    def encode(self, x):
        """x [B, 3, H, W] in [-1, 1] -> DiagonalGaussianDistribution over z (autoencoder.py:324-328)."""
        B, c, H, W = x.shape
        pl = self._encode_plan(B, H, W)
        with torch.cuda.device(pl.dev):
            return DiagonalGaussianDistribution(pl.run(x).clone())

    @torch.no_grad()
    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior

    def get_last_layer(self):
        return self.decoder.conv_out.weight
