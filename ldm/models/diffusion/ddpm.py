"""ldm.models.diffusion.ddpm -> upgpt_amd.ddpm."""
from upgpt_amd.ddpm import DDPM, DiffusionWrapper, LatentDiffusion, disabled_train  # noqa: F401
