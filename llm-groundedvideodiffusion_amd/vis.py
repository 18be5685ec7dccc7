"""Output writers with the reference's file contract (utils/vis.py:142-161): `video_{k}.gif` and a bz2-compressed
joblib of the uint8 (F,H,W,3) frames that scripts/eval_owl_vit.py reads back."""
import joblib
import numpy as np


def save_frames(path, frames, formats="gif", fps=8):
    if isinstance(formats, (list, tuple)):
        for fmt in formats:
            save_frames(path, frames, fmt, fps)
        return
    frames = np.asarray(frames)
    if formats == "gif":
        from PIL import Image
        imgs = [Image.fromarray(f) for f in frames]
        imgs[0].save(f"{path}.gif", save_all=True, append_images=imgs[1:], loop=0, duration=int(1000 / fps))
    elif formats == "npz":
        np.savez_compressed(f"{path}.npz", frames)
    elif formats == "joblib":
        joblib.dump(frames, f"{path}.joblib", compress=("bz2", 3))
    else:
        raise ValueError(f"Unknown format: {formats}")
