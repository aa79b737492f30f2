"""ldm.modules.poses.poses -> upgpt_amd.poses."""
from upgpt_amd.poses import DummyModel, LinearProject  # noqa: F401
