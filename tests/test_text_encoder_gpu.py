"""CLIP text encoder on the HIP kernels against `transformers.CLIPTextModel` itself — the third-party
implementation the reference calls (`controllable_pipeline_text_to_video_synth.py:251-286,751-763`) is importable in
this image, so this row's parity is pinned against the real thing (random-init weights, no hub access needed)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvd_amd  # noqa: E402
from lvd_amd import ops  # noqa: E402
from lvd_amd.text_encoder import CLIPTextConfig, HipCLIPTextEncoder  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_gelu_kernel_and_causal_attention():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn(77, 512, device=DEV, generator=g) * 2).bfloat16()
    assert rel(ops.gelu(x), torch.nn.functional.gelu(x.float())) < 4e-3
    assert rel(ops.gelu(x, "quick_gelu"), x.float() * torch.sigmoid(1.702 * x.float())) < 4e-3
    B, L, H = 3, 77, 2
    qkv = (torch.randn(B * L, 3 * H * 64, device=DEV, generator=g)).bfloat16()
    o = torch.empty(B * L, H * 64, device=DEV, dtype=torch.bfloat16)
    rows = ops.RowMap(ninner=1, os=L, is_=0, step=1)
    ops.attention_fwd(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], o, samples=B, heads=H, sq=L, skv=L, qmap=rows, kvmap=rows, scale=0.125, causal=True)
    q, k, v = [t.float().reshape(B, L, H, 64).permute(0, 2, 1, 3) for t in (qkv[:, :128], qkv[:, 128:256], qkv[:, 256:])]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True).permute(0, 2, 1, 3).reshape(B * L, H * 64)
    assert rel(o, ref) < 6e-3


@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_text_encoder_matches_transformers(act):
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                                         max_position_embeddings=77, hidden_act=act, eos_token_id=2, bos_token_id=0, pad_token_id=1)
    torch.manual_seed(0)
    model = transformers.CLIPTextModel(hf_cfg).eval()
    sd = {k: v.to(torch.bfloat16).float() for k, v in model.state_dict().items()}  # bf16-representable weights on both sides
    model.load_state_dict(sd)
    ids = torch.randint(3, 999, (2, 77), generator=torch.Generator().manual_seed(1))
    ids[:, 0] = 0
    ids[0, 9:] = 1
    ids[0, 9] = 999   # the highest id marks EOS under the legacy rule
    ids[1, 30] = 999
    with torch.no_grad():
        ref = model(input_ids=ids)
    cfg = CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2, hidden_act=act)
    enc = HipCLIPTextEncoder(cfg, {"text_model." + k if not k.startswith("text_model.") else k: v for k, v in sd.items()}, device=DEV)
    out = enc(ids.to(DEV), attention_mask=None)
    assert out[0].shape == (2, 77, 128) and out.pooler_output.shape == (2, 128)
    assert rel(out[0], ref.last_hidden_state) < 2e-2, rel(out[0], ref.last_hidden_state)
    assert rel(out.pooler_output, ref.pooler_output) < 2e-2
