"""Baseline run-model `zeroscope` (reference: generation/zeroscope_dpm.py — stock text-to-video sampling, no layout
conditioning) on the HIP denoiser.  The reference builds its pipeline at import time on "cuda" in fp16 and `generate.py`
refuses < 24 frames; both restrictions are lifted here (BASELINE config 0 runs 256x144x8)."""
from ._common import BASE_MODELS, Method, configure  # noqa: F401

_m = Method("zeroscope", use_guidance=False, use_gligen=False)
version = _m.version


def init(option=""):
    if option not in ("", None):
        raise ValueError(f"zeroscope option {option!r} (the XL upsampler) is out of scope")
    return _m.init("zeroscope")


def run(parsed_layout, seed, **kwargs):
    return _m.run(parsed_layout, seed, **kwargs)


def run_many(jobs, **kwargs):
    """V (layout, seed) samples through one denoising loop (generate.py --videos-per-gpu V; _common.Method.run_many)."""
    return _m.run_many(jobs, **kwargs)
