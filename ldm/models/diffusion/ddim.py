"""ldm.models.diffusion.ddim -> upgpt_amd.ddim."""
from upgpt_amd.ddim import DDIMSampler, noise_like  # noqa: F401
