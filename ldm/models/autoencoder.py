"""ldm.models.autoencoder -> upgpt_amd.vae."""
from upgpt_amd.vae import AutoencoderKL, DiagonalGaussianDistribution  # noqa: F401
