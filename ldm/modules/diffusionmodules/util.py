"""ldm.modules.diffusionmodules.util -> upgpt_amd.schedule / ddim."""
from upgpt_amd.ddim import noise_like  # noqa: F401
from upgpt_amd.schedule import (extract_into_tensor, make_beta_schedule, make_ddim_sampling_parameters,  # noqa: F401
                                make_ddim_timesteps)
