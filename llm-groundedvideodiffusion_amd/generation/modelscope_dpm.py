"""Baseline run-model `modelscope` (reference: generation/modelscope_dpm.py) on the HIP denoiser."""
from ._common import Method, configure  # noqa: F401

_m = Method("modelscope", use_guidance=False, use_gligen=False)
version = _m.version


def init(option=""):
    return _m.init("modelscope256" if option == "256" else "modelscope512")


def run(parsed_layout, seed, **kwargs):
    return _m.run(parsed_layout, seed, **kwargs)


def run_many(jobs, **kwargs):
    """V (layout, seed) samples through one denoising loop (generate.py --videos-per-gpu V; _common.Method.run_many)."""
    return _m.run_many(jobs, **kwargs)
