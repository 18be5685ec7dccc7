"""Mirror of /root/reference/models/pipelines.py:21-150 — same function name and call contract, HIP arithmetic."""
from ..guidance import DEFAULT_GUIDANCE_ATTN_KEYS, hip_latent_backward_guidance  # noqa: F401

latent_backward_guidance = hip_latent_backward_guidance
