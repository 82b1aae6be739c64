"""videorenderer_amd — MI355X-native shader video processor (the CDX11VideoProcessor::Process path of
MPC Video Renderer: convert -> resize -> PQ/HLG->SDR -> dither) as hand-written HIP behind a C-ABI.

Only what that path needs lives here: csrc/ (HIP kernels + C-ABI, built into libmpcvr.so), api.py
(ctypes mirror of the reference's CVideoProcessor interface), synth.py (synthetic frames of SURVEY.md
§8d), dist.py (frame sharding + parameter-blob broadcast over torch.distributed/RCCL).
"""
from .api import *  # noqa: F401,F403
from .api import VideoProcessor, default_settings, load_library, make_extfmt  # noqa: F401
