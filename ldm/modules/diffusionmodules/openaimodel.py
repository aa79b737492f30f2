"""ldm.modules.diffusionmodules.openaimodel -> upgpt_amd.unet."""
from upgpt_amd.unet import UNetModel  # noqa: F401
