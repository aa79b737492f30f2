"""ldm.modules.ema -> upgpt_amd.ema."""
from upgpt_amd.ema import LitEma  # noqa: F401
