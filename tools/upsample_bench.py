"""Time the zeroscope-XL video-to-video pass at the reference's geometry: 24 frames 320x576 -> 576x1024 (latent 72x128),
strength 0.35 of 50 steps = 17 denoising steps at CFG scale 15, random-init weights of the real topologies."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvd_amd  # noqa: E402,F401
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.upsample import HipVideoToVideo  # noqa: E402
from lvd_amd.vae import HipVAEDecoder, HipVAEEncoder  # noqa: E402
from lvd_amd.weights import UNetConfig, VAEConfig, synthetic_state_dict, synthetic_vae_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ucfg, vcfg = UNetConfig(), VAEConfig()
vsd = synthetic_vae_state_dict(vcfg, seed=0, device="cuda", encoder=True)
pipe = HipVideoToVideo(HipUNet3D(ucfg, synthetic_state_dict(ucfg, seed=0, device="cuda"), device="cuda"), HipVAEEncoder(vcfg, vsd), HipVAEDecoder(vcfg, vsd))
video = np.random.RandomState(0).randint(0, 256, (24, 320, 576, 3)).astype(np.uint8)
g = torch.Generator().manual_seed(0)
pe, ne = torch.randn(1, 77, 1024, generator=g), torch.randn(1, 77, 1024, generator=g)
for rep in range(2):  # first pass autotunes the GEMM shapes of this geometry
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z = pipe.enc.encode(torch.from_numpy(video), eps=torch.randn(24, 4, 72, 128, generator=g), size=(576, 1024))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    frames = pipe(video=video, strength=0.35, num_inference_steps=steps, prompt_embeds=pe, negative_prompt_embeds=ne, size=(576, 1024), generator=g)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = steps - HipVideoToVideo.get_timesteps(steps, 0.35)
    print(f"pass {rep}: encode 24x576x1024 {1e3 * (t1 - t0):.0f} ms; whole video-to-video ({n} steps incl. encode + decode) {t2 - t1:.2f} s; "
          f"frames {tuple(frames.shape)} finite={bool(torch.isfinite(frames).all())} peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
