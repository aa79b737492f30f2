"""AutoencoderKL — drop-in for ldm.models.autoencoder.AutoencoderKL (ctor 286-306,
decode 330-333) with the Decoder (model.py:462-568) running on libupk.so.  The encoder's
parameters are held (checkpoint keys first_stage_model.encoder.*) but encode() is a
"next" row of SURVEY.md §8f and raises until it is built on the same kernels."""
import os

import torch

from .arch import VAEArch
from .config import instantiate_from_config
from .params import ParamTree, weights_fingerprint


class AutoencoderKL(ParamTree):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=None, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        assert ddconfig["double_z"]
        self.image_key = image_key
        self.arch = VAEArch(ddconfig, embed_dim)
        self.embed_dim = embed_dim
        self.add_params(self.arch.param_shapes())
        try:  # training-only object; kept when cheap (bbox.yaml uses torch.nn.Identity)
            self.loss = instantiate_from_config(lossconfig) if lossconfig else None
        except Exception:
            self.loss = None
        if colorize_nlabels is not None:
            assert type(colorize_nlabels) == int
            self.register_buffer("colorize", torch.randn(3, colorize_nlabels, 1, 1))
        if monitor is not None:
            self.monitor = monitor
        self._packed = None
        self._plans = {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def _decode_plan(self, B, h, w, scale_factor):
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
            with torch.cuda.device(p.device):
                self._packed = (fp, PackedVAEDecoder(ctx, self.arch, lambda n: params[n].data))
            self._plans = {}
        key = (B, h, w, float(scale_factor))
        if key not in self._plans:
            if len(self._plans) >= 4:
                self._plans.pop(next(iter(self._plans)))
            with torch.cuda.device(p.device):
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

    def encode(self, x):
        raise NotImplementedError("AutoencoderKL.encode (VAE encoder) is scheduled after the denoising hot path "
                                  "(SURVEY.md §8f-2); only decode() is implemented on the HIP kernels")

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("AutoencoderKL.forward needs encode(); see encode()")

    def get_last_layer(self):
        return self.decoder.conv_out.weight
