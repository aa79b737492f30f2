"""upgpt_amd — MI355X-native implementation of UPGPT's denoising hot path
(UNetModel.forward x DDIMSampler loop -> VAE decode) behind the reference's Python call
surface.  Compute = hand-written HIP kernels for gfx950 in libupk.so (include/upk.h);
host = Python on PyTorch-ROCm (allocator, streams, torch.distributed only)."""
from .config import instantiate_from_config, load_config  # noqa: F401

__version__ = "0.1.0"


def lane(i, stream=None):
    """Execution lane `i` for the calling thread (context manager; upgpt_amd._lib.lane): batches sampled in different
    lanes may be in flight on one GPU at the same time — own scratch, own buffers, shared weights."""
    from ._lib import lane as _lane
    return _lane(i, stream)


def model_params(kind="bbox", overrides=None):
    """Constructor kwargs of LatentDiffusion for the restated reference configs in synth.py (`bbox`, `upscale`, `tiny`),
    conditioning stages replaced by DummyModel (embeddings fed directly)."""
    import copy

    from . import synth
    unet = {"bbox": synth.BBOX_UNET, "tiny": synth.TINY_UNET, "upscale": synth.UPSCALE_UNET}[kind]
    dd = {"bbox": synth.BBOX_DDCONFIG, "tiny": synth.TINY_DDCONFIG, "upscale": synth.UPSCALE_DDCONFIG}[kind]
    up = kind == "upscale"
    dummy = {"target": "upgpt_amd.poses.DummyModel"}
    p = dict(
        linear_start=0.0001 if up else 0.00085, linear_end=0.02 if up else 0.012, num_timesteps_cond=1,
        log_every_t=1000, timesteps=1000, first_stage_key="image", cond_stage_key="txt",
        concat_key="lr" if up else "person_mask", image_size=[128, 96] if up else [32, 24],
        crop_size=[512, 352] if up else [256, 176], channels=3 if up else 4, cond_stage_trainable=False,
        conditioning_key="hybrid", scale_factor=0.18215, use_ema=not up,
        unet_config={"target": "upgpt_amd.unet.UNetModel", "params": copy.deepcopy(unet)},
        first_stage_config={"target": "upgpt_amd.vae.AutoencoderKL",
                            "params": {"embed_dim": dd["z_channels"], "ddconfig": copy.deepcopy(dd),
                                       "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config=dummy,
        extra_cond_stages={"style_cond": dict(dummy, cond_stage_key="styles")} if up else {
            "style_cond": dict(dummy, cond_stage_key="styles"),
            "pose_cond": {"target": "upgpt_amd.poses.LinearProject", "cond_stage_key": "smpl",
                          "params": {"input_dim": 85, "output_dim": 768}}},
    )
    if overrides:
        p.update(overrides)
    return p


def model_config(kind="bbox", overrides=None):
    """The same as a config tree shaped like configs/deepfashion/bbox.yaml ({'model': {'target', 'params'}}), with the
    reference's dotted path as target: what load_model_from_config / instantiate_from_config take."""
    return {"model": {"target": "ldm.models.diffusion.ddpm.LatentDiffusion", "params": model_params(kind, overrides)}}


def build_model(kind="bbox", overrides=None):
    """LatentDiffusion(**model_params(kind, overrides)).eval()"""
    from .ddpm import LatentDiffusion
    return LatentDiffusion(**model_params(kind, overrides)).eval()
