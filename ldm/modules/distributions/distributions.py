"""ldm.modules.distributions.distributions -> upgpt_amd.vae."""
from upgpt_amd.vae import DiagonalGaussianDistribution  # noqa: F401
