"""ldm.models.diffusion.plms -> upgpt_amd.plms."""
from upgpt_amd.plms import PLMSSampler  # noqa: F401
