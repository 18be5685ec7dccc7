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


def draw_boxes(frames, boxes, phrases, ignore_all_zeros=True):
    """Frames uint8 (F,H,W,3) with every object's box of that frame outlined in red and labelled (utils/utils.py:14-31
    `draw_box`, used by generation/lvd.py:179-192 for `save_annotated_videos`).  boxes[o][f] = (x0, y0, x1, y1) fractions;
    an all-zero box means the object is absent in that frame."""
    from PIL import Image, ImageDraw
    out = []
    for f, frame in enumerate(np.asarray(frames)):
        img = Image.fromarray(frame)
        draw = ImageDraw.Draw(img)
        W, H = img.size
        for track, phrase in zip(boxes, phrases):
            x0, y0, x1, y1 = track[f][:4]
            if ignore_all_zeros and x0 == 0 and y0 == 0 and x1 == 0 and y1 == 0:
                continue
            draw.rectangle([int(x0 * W), int(y0 * H), int(x1 * W), int(y1 * H)], outline="red", width=5)
            draw.text((int(x0 * W) + 5, int(y0 * H) + 5), phrase, fill=(255, 0, 0))
        out.append(np.asarray(img))
    return np.stack(out)
