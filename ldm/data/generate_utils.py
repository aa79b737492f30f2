"""ldm.data.generate_utils — import path of the demo's caller glue (app.py:19, the inference notebooks); the
implementation is upgpt_amd/inference.py."""
from upgpt_amd.inference import (InferenceModel, clip_normalize, convert_fname, draw_styles, get_coord,  # noqa: F401
                                 get_empty_style, get_mask, interp_mask, load_clip_weights, load_model_from_config,
                                 style_names)
