"""Cross-check of the two restatements of the six diffusers==0.27.2 classes the reference imports but does not vendor
(parity unpinned — see oracle/diffusers_restated.py): the nn.Module transcription that oracle/make_golden.py hands to the REAL reference
code (and through which every tests/golden/*.npz fixture was produced) against the functional transcription in oracle/unet_ref.py
that the GPU parity tests use — op by op, on independently seeded random parameters, NON-zero-initialised (diffusers zero-inits
TemporalConvLayer.conv4; a slip there would otherwise be invisible).  CPU only."""
import torch

from oracle import unet_ref
from oracle.diffusers_restated import Downsample2D, ResnetBlock2D, TemporalConvLayer, TimestepEmbedding, Timesteps, Upsample2D


def _randomise(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.dim() > 1 else 0.5))
    return module.eval()


def _sd(module, prefix):
    return {f"{prefix}.{k}": v.detach() for k, v in module.state_dict().items()}


def _close(a, b, what):
    err = ((a - b).norm() / b.norm()).item()
    assert err < 1e-5, f"{what}: the two restatements differ (rel-L2 {err:.2e})"


def test_resnet_block_with_and_without_shortcut():
    for cin, cout, seed in ((64, 64, 1), (96, 64, 2)):
        m = _randomise(ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=128, eps=1e-5, groups=32, dropout=0.0,
                                     time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True), seed)
        g = torch.Generator().manual_seed(10 + seed)
        x, temb = torch.randn(6, cin, 9, 7, generator=g), torch.randn(6, 128, generator=g)
        with torch.no_grad():
            _close(unet_ref.resnet(_sd(m, "r"), "r", x, temb, 32, 1e-5), m(x, temb), f"ResnetBlock2D {cin}->{cout}")


def test_temporal_conv_layer():
    m = _randomise(TemporalConvLayer(64, 64, dropout=0.1, norm_num_groups=32), 3)   # dropout 0.1 as the reference passes it: identity in eval
    assert float(m.conv4[-1].weight.abs().sum()) > 0
    g = torch.Generator().manual_seed(13)
    frames = 5
    x = torch.randn(2 * frames, 64, 4, 6, generator=g)
    with torch.no_grad():
        _close(unet_ref.temporal_conv(_sd(m, "t"), "t", x, frames, 32), m(x, num_frames=frames), "TemporalConvLayer")


def test_down_and_upsample():
    g = torch.Generator().manual_seed(14)
    x = torch.randn(3, 32, 10, 9, generator=g)  # odd width: stride-2 pad-1 output is ceil(n/2)
    d = _randomise(Downsample2D(32, use_conv=True, out_channels=48, padding=1, name="op"), 4)
    u = _randomise(Upsample2D(32, use_conv=True, out_channels=32), 5)
    with torch.no_grad():
        sd = _sd(d, "d")
        _close(torch.nn.functional.conv2d(x, sd["d.conv.weight"], sd["d.conv.bias"], stride=2, padding=1), d(x), "Downsample2D")  # unet_ref.unet_forward's call
        sd = _sd(u, "u")
        up = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
        _close(torch.nn.functional.conv2d(up, sd["u.conv.weight"], sd["u.conv.bias"], padding=1), u(x), "Upsample2D x2")
        up = torch.nn.functional.interpolate(x, size=(19, 17), mode="nearest")
        _close(torch.nn.functional.conv2d(up, sd["u.conv.weight"], sd["u.conv.bias"], padding=1), u(x, output_size=(19, 17)), "Upsample2D to size")


def test_timestep_features_and_embedding():
    t = torch.tensor([0.0, 1.0, 25.0, 500.0, 961.0, 999.0])
    with torch.no_grad():
        _close(unet_ref.timestep_embedding(t, 320), Timesteps(320, True, 0)(t), "Timesteps(flip_sin_to_cos=True, shift=0)")
        m = _randomise(TimestepEmbedding(64, 256, "silu"), 6)
        sd = _sd(m, "time_embedding")
        x = torch.randn(4, 64, generator=torch.Generator().manual_seed(15))
        ref = unet_ref._lin(sd, "time_embedding.linear_2", torch.nn.functional.silu(unet_ref._lin(sd, "time_embedding.linear_1", x)))
        _close(ref, m(x), "TimestepEmbedding")
