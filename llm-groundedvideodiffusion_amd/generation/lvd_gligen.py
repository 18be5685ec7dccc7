"""Run-model `lvd-gligen` (reference: generation/lvd_gligen.py) on the HIP denoiser; the shared runner is _common.Method."""
from ._common import Method, configure  # noqa: F401

_m = Method("lvd-gligen", use_guidance=False, use_gligen=True)
version = _m.version


def init(base_model):
    """base_model in {"zeroscope", "modelscope256", "modelscope512"} -> (H, W)."""
    return _m.init(base_model)


def run(parsed_layout, seed, **kwargs):
    return _m.run(parsed_layout, seed, **kwargs)


def run_many(jobs, **kwargs):
    """V (layout, seed) samples through one denoising loop (generate.py --videos-per-gpu V; _common.Method.run_many)."""
    return _m.run_many(jobs, **kwargs)
