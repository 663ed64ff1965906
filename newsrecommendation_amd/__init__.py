"""MI355X-native NRMS / NAML encoder + scorer (hot path of patngnw/NewsRecommendation).

`newsrecommendation_amd.model.{NRMS,NAML,model_utils}` mirror the reference's `src/model/` module
surface (class names, constructor signatures, forward signatures, state_dict keys) on top of
hand-written HIP kernels behind the C ABI of include/nrhip.h (libnrhip.so).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "model", "data", "train"]
