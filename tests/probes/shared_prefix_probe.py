import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lvd_amd
from lvd_amd.engine import HipUNet3D
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict
for gated in (False, True):
    cfg = UNetConfig(attention_type="gated" if gated else "default", **TINY)
    net = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0))
    gen = torch.Generator().manual_seed(3)
    for V in (1, 2):
        Fr = 4
        lat = torch.randn(V, 4, Fr, 16, 16, generator=gen).cuda()
        ehs = torch.randn(2 * V, 77, cfg.cross_attention_dim, generator=gen).cuda()
        text = net.encode_text(ehs)
        gl = None
        if gated:
            gl = {"boxes": torch.rand(2 * V * Fr, 30, 4, generator=gen), "masks": (torch.rand(2 * V * Fr, 30, generator=gen) > 0.5).float(),
                  "positive_embeddings": torch.randn(2 * V * Fr, 30, cfg.cross_attention_dim, generator=gen)}
        full = net.forward(lat.repeat_interleave(2, 0).contiguous(), 500, text=text, gligen=gl)
        shared = net.forward(lat, 500, text=text, gligen=gl, cfg_pairs=True)
        print(f"gated={gated} V={V}: rel {((shared-full).norm()/full.norm()).item():.3e}  bit-equal {torch.equal(shared, full)}  differing {(shared!=full).sum().item()} of {full.numel()}")
