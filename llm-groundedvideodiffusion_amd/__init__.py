"""MI355X-native denoise step of LLM-grounded Video Diffusion (hot path only).

Sub-modules
  hip        ctypes binding of the C ABI in include/lvdhip.h (liblvdhip.so, built in-tree)
  ops        torch-tensor level wrappers of the kernels (shape checks, workspace handling)
  weights    seeded synthetic weights + repacking of reference-named state_dicts
  engine     the token-matrix UNet3D executor (forward, recorded forward, hand-scheduled backward)
  guidance   fused cross-attention-energy loss + latent_backward_guidance drop-in
"""
__version__ = "0.1.0"
