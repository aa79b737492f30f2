"""ldm.util -> upgpt_amd.config (instantiate_from_config & friends)."""
from upgpt_amd.config import (count_params, default, exists, get_obj_from_str,  # noqa: F401
                              instantiate_from_config)
