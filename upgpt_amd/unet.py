"""UNetModel — drop-in for ldm.modules.diffusionmodules.openaimodel.UNetModel
(constructor kwargs openaimodel.py:443-469, forward :710-742) whose compute runs in the
hand-written HIP kernels of libupk.so.  Weights live in a ParamTree with the reference's
state-dict keys (time_embed.0.weight, input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight,
out.2.weight, ...)."""
import os

import torch

from .arch import UNetArch
from .params import ParamTree, weights_fingerprint
from ._check import require


class UNetModel(ParamTree):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.arch = UNetArch(*args, **kwargs)
        a = self.arch
        self.image_size = a.image_size
        self.in_channels = a.in_channels
        self.model_channels = a.model_channels
        self.out_channels = a.out_channels
        self.num_res_blocks = a.num_res_blocks
        self.attention_resolutions = a.attention_resolutions
        self.channel_mult = a.channel_mult
        self.num_classes = None
        self.dtype = torch.float32  # public boundary dtype (openaimodel.py:493); fp16 is internal
        self.num_heads = a.num_heads
        self.num_head_channels = a.num_head_channels
        self.predict_codebook_ids = False
        self.add_params(a.param_shapes())
        self._packed = {}     # tag -> (fingerprint, PackedUNet)
        self._plans = {}      # (tag, B, H, W, n_ctx, rows, mode, several batches in flight, lane) -> UNetPlan
        self._weight_override = None  # (tag, callable name -> tensor): EMA weights without copying

    # ---- engine plumbing
    def _device(self):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError(
                "upgpt_amd.UNetModel computes only through the HIP kernels on an MI355X: move the model to "
                "'cuda' first (parameters are on %s). There is no CPU fallback." % p.device)
        return p.device

    def set_weight_override(self, tag=None, getter=None, fingerprint=None):
        """Compute with an alternative weight set (LitEma shadows) without touching the
        parameters; `fingerprint` is a callable used to invalidate the packed cache."""
        self._weight_override = None if tag is None else (tag, getter, fingerprint)

    def packed(self):
        from ._lib import PLAN_LOCK
        with PLAN_LOCK:
            return self._packed_locked()

    def _packed_locked(self):
        from ._lib import get_context
        from .engine import PackedUNet
        dev = self._device()
        ctx = get_context(dev)
        if self._weight_override is None:
            tag, fp = "live", weights_fingerprint(self)
            params = dict(self.named_parameters())
            get = lambda n: params[n].data
        else:
            tag, get, fpf = self._weight_override
            fp = fpf()
        ent = self._packed.get(tag)
        if ent is None or ent[0] != fp:
            from ._lib import host_io
            with torch.cuda.device(dev), host_io():
                pk = PackedUNet(ctx, self.arch, get)
                # the pack kernels ran on THIS thread's stream; other lanes read the packed weights from theirs
                torch.cuda.current_stream(dev).synchronize()
            self._packed[tag] = (fp, pk)
            stale = [k for k in self._plans if k[0] == tag]
            if stale:
                torch.cuda.synchronize(dev)  # (another lane may still be replaying graphs of the old weight set)
            for k in stale:
                self._plans.pop(k).close()  # (plans of the old weight set: their captured graphs go with them)
        return ctx, tag, self._packed[tag][1]

    def plan(self, B, H, W, n_ctx, rows, mode):
        from ._lib import PLAN_LOCK
        with PLAN_LOCK:
            return self._plan_locked(B, H, W, n_ctx, rows, mode)

    def _plan_locked(self, B, H, W, n_ctx, rows, mode):
        from ._lib import concurrency, current_lane
        from .engine import UNetPlan
        ctx, tag, pk = self.packed()  # (ctx = the calling thread's lane: own workspace, own buffers; weights shared)
        key = (tag, B, H, W, n_ctx, rows, mode, concurrency() > 1, current_lane())
        pl = self._plans.get(key)
        if pl is None:
            mine = [k for k in self._plans if k[-1] == key[-1]]
            if len(mine) >= 8:  # bound device memory held by stale shapes — per lane: another lane's plans may be executing
                self._plans.pop(mine[0]).close()
            from ._lib import host_io
            with torch.cuda.device(ctx.device), host_io():  # (descriptor tables are uploaded from the host)
                pl = UNetPlan(ctx, pk, B, H, W, n_ctx, rows, mode)
                pl.apply_tuning(tune_missing=os.environ.get("UPGPT_AUTOTUNE", "0") == "1")
            self._plans[key] = pl
        return pl

    def invalidate(self):
        """Drops the packed weights and every plan (call after editing parameters through `.data`)."""
        self._packed.clear()
        for pl in self._plans.values():
            pl.close()
        self._plans.clear()

    # ---- reference surface
    def convert_to_fp16(self):
        """No-op like the reference's stub (openaimodel.py:23-28): fp16 is internal to the kernels."""

    def convert_to_fp32(self):
        pass

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """x [N, in_channels, H, W], timesteps [N], context [N, n_ctx, context_dim] -> eps [N, out, H, W]."""
        require(y is None, "must specify y if and only if the model is class-conditional", NotImplementedError)
        if context is None:
            raise NotImplementedError("UNetModel without context (cross-attn defaults to self-attn) is not on the "
                                      "UPGPT path")
        B, Cin, H, W = x.shape
        require(Cin == self.in_channels, lambda: "expected %d input channels, got %d" % (self.in_channels, Cin), ValueError)
        require(timesteps is not None and timesteps.shape[0] == B, "timesteps must be given, one per sample", ValueError)
        pl = self.plan(B, H, W, context.shape[1], B, "forward")
        with torch.cuda.device(pl.dev):
            pl.load_x_nchw(x, 0, pl.cin_pad)
            pl.t_rows.copy_(timesteps.to(pl.dev, torch.float32))
            pl._t_rows_key = None  # (a sampler sharing this plan re-uploads its rows)
            pl.load_context(context)
            pl.prep.run()
            pl.body.run()
            return pl.eps.clone().to(x.dtype)
