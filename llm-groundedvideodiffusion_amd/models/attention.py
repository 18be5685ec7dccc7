"""Mirror of the pieces of /root/reference/models/attention.py the pipeline touches by name."""
import torch.nn as nn


class GatedSelfAttentionDense(nn.Module):
    """Handle for one GLIGEN fuser (models/attention.py:26-60).  The pipeline toggles `.enabled` on every module whose class
    is *named* GatedSelfAttentionDense (controllable_pipeline_text_to_video_synth.py:535-539); the arithmetic lives in
    HipUNet3D._fuser.  Parameters stay in the owning UNet's packed weight store."""

    def __init__(self, prefix):
        super().__init__()
        self.prefix = prefix
        self.enabled = True

    def forward(self, x, objs, fuser_attn_kwargs=None):  # pragma: no cover - never called, the engine runs the fuser
        raise RuntimeError("GatedSelfAttentionDense is executed inside the HIP engine")
