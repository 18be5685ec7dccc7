#!/usr/bin/env python
"""Upsample generated videos with the zeroscope-XL video-to-video pass on the HIP kernels.

    python scripts/upsample.py --videos img_generations/.../video_0.joblib --prompts "A bear walks ..." --use_zsxl --horizontal \
        [--checkpoint-dir /path/to/zeroscope_v2_XL | --synthetic-weights]

Command line of /root/reference/scripts/upsample.py:131-158.  `--use_zsxl` is provided (`lvd_amd.upsample`); `--use_sdxl` and
`--use_zssdxl` need the SDXL refiner, a different UNet family that this build does not contain, and exit with that message.
`--checkpoint-dir` holds `unet.pt`, `vae.pt` (torch-saved state_dicts with the diffusers names), `text_encoder.pt` and the CLIP
tokenizer files; there is no hub access.  `--output-mp4` needs OpenCV, which this image lacks: gif + joblib are written."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NEGATIVE = ("dull, gray, unrealistic, colorless, drawing, painting, crayon, sketch, graphite, impressionist, noisy, blurry, soft, "
            "deformed, ugly")


def build(args, device):
    import torch
    from lvd_amd.engine import HipUNet3D
    from lvd_amd.text_encoder import CLIPTextConfig, HipCLIPTextEncoder
    from lvd_amd.upsample import HipVideoToVideo
    from lvd_amd.vae import HipVAEDecoder, HipVAEEncoder
    from lvd_amd.weights import UNetConfig, VAEConfig, synthetic_state_dict, synthetic_vae_state_dict
    ucfg, vcfg = UNetConfig(), VAEConfig()
    if args.synthetic_weights:
        unet_sd = synthetic_state_dict(ucfg, seed=0, device=device)
        vae_sd = synthetic_vae_state_dict(vcfg, seed=0, device=device, encoder=True)
        g = torch.Generator().manual_seed(0)
        encode_prompt = lambda texts: torch.randn(len(texts), 77, ucfg.cross_attention_dim, generator=g)  # stand-in conditioning
    else:
        d = args.checkpoint_dir
        if d is None:
            raise SystemExit("--checkpoint-dir (zeroscope_v2_XL state_dicts) or --synthetic-weights is required")
        unet_sd = torch.load(os.path.join(d, "unet.pt"), map_location="cpu")
        vae_sd = torch.load(os.path.join(d, "vae.pt"), map_location="cpu")
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(os.path.join(d, "tokenizer"))
        clip = HipCLIPTextEncoder(CLIPTextConfig(), torch.load(os.path.join(d, "text_encoder.pt"), map_location="cpu"), device=device)
        encode_prompt = lambda texts: clip(tok(texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt")["input_ids"])[0]
    return HipVideoToVideo(HipUNet3D(ucfg, unet_sd, device=device), HipVAEEncoder(vcfg, vae_sd, device=device),
                           HipVAEDecoder(vcfg, vae_sd, device=device), encode_prompt=encode_prompt)


def _local_device():
    """cuda:<LOCAL_RANK>, made the current device (kernels launch on the current device's stream: one process per GPU)."""
    import torch
    idx = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(idx)
    return f"cuda:{idx}"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", nargs="+", required=True, type=str, help="path to videos in joblib format")
    ap.add_argument("--prompts", nargs="+", required=True, type=str, help="prompts")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--strength", type=float, default=0.35)
    ap.add_argument("--negative_prompt", type=str, default=NEGATIVE)
    ap.add_argument("--use_zsxl", action="store_true")
    ap.add_argument("--use_sdxl", action="store_true")
    ap.add_argument("--use_zssdxl", action="store_true")
    ap.add_argument("--horizontal", action="store_true", help="576x320 -> 1024x576; otherwise square -> 1024x1024")
    ap.add_argument("--output-mp4", action="store_true")
    ap.add_argument("--checkpoint-dir", default=None)
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--size", nargs=2, type=int, default=None, help="override the target (height width), multiples of 8")
    ap.add_argument("--num_inference_steps", type=int, default=50)
    args = ap.parse_args(argv)
    if args.use_sdxl or args.use_zssdxl:
        raise SystemExit("--use_sdxl / --use_zssdxl need stable-diffusion-xl-refiner (a different UNet family): not part of this build")
    if args.output_mp4:
        raise SystemExit("--output-mp4 needs OpenCV, which is not installed here; the gif and joblib outputs are always written")
    if not args.use_zsxl:
        raise SystemExit("nothing to do: pass --use_zsxl")
    import joblib
    import numpy as np
    import torch
    import lvd_amd  # noqa: F401
    from lvd_amd import vis
    pipe = build(args, _local_device())
    prompts = args.prompts * len(args.videos) if len(args.prompts) == 1 else args.prompts
    size = tuple(args.size) if args.size else ((576, 1024) if args.horizontal else (1024, 1024))
    written = []
    for video_path, prompt in zip(args.videos, prompts):
        video_path = video_path.replace(".gif", ".joblib")
        print(f"Video path: {video_path}, prompt: {prompt}")
        save_path = video_path.replace(".joblib", "_zsxl" if args.strength == 0.35 else f"_zsxl_s{args.strength}")
        if os.path.exists(save_path + ".joblib"):
            print(f"{save_path + '.joblib'} exists, skipping")
            continue
        frames = pipe(prompt, video=joblib.load(video_path), strength=args.strength, negative_prompt=args.negative_prompt,
                      generator=torch.manual_seed(args.seed), size=size, num_inference_steps=args.num_inference_steps)
        frames = (frames.cpu().numpy() * 255.0).astype(np.uint8)
        vis.save_frames(save_path, frames, ["gif", "joblib"], fps=8)
        print(f"Zeroscope XL upsampled image saved at: {save_path + '.gif'}")
        written.append(save_path)
    return written


if __name__ == "__main__":
    main()
