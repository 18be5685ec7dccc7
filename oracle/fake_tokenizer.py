"""TEST INFRASTRUCTURE: a deterministic word-level stand-in for the CLIP tokenizer (its vocab files are not on disk).
Same call surface the DSL front-end uses: tokenizer([text], padding=..., max_length=77, return_tensors="np")["input_ids"],
_convert_id_to_token, eos_token.  Words and punctuation marks are separate tokens, as in CLIP's BPE for common words."""
import re
import zlib

import numpy as np


class FakeTokenizer:
    bos_token, eos_token = "<|startoftext|>", "<|endoftext|>"
    model_max_length = 77

    def __init__(self):
        self.vocab = {self.bos_token: 0, self.eos_token: 1}
        self.inv = {0: self.bos_token, 1: self.eos_token}

    def _id(self, tok):
        # a function of the token alone (not of the order in which texts were seen): two processes that tokenize different subsets of a
        # prompt list agree on every id, as a real vocabulary does
        if tok not in self.vocab:
            i = 2 + zlib.crc32(tok.encode()) % 49000
            while i in self.inv and self.inv[i] != tok:
                i += 1
            self.vocab[tok] = i
            self.inv[i] = tok
        return self.vocab[tok]

    def __call__(self, texts, padding="do_not_pad", max_length=77, return_tensors="np", **kw):
        out = []
        for t in texts:
            words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", t.lower())
            ids = [0] + [self._id(w + "</w>") for w in words][: max_length - 2] + [1]
            out.append(ids)
        return {"input_ids": np.array(out, dtype=np.int64) if len({len(o) for o in out}) == 1 else out}

    def _convert_id_to_token(self, i):
        return self.inv[int(i)]


class _Batch(dict):
    def __getattr__(self, k):
        return self[k]

    def to(self, device):
        return _Batch({k: v.to(device) for k, v in self.items()})


class FakeClipTokenizer(FakeTokenizer):
    """Adds the padding / torch-tensor call styles the pipeline uses (max_length padding, `padding=True`, `.to(device)`)."""

    def __call__(self, texts, padding="do_not_pad", max_length=77, return_tensors="np", truncation=False, **kw):
        import torch
        if isinstance(texts, str):
            texts = [texts]
        if return_tensors == "np":
            return super().__call__(texts, padding=padding, max_length=max_length, return_tensors="np")
        rows = [list(super(FakeClipTokenizer, self).__call__([t], max_length=max_length)["input_ids"][0]) for t in texts]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        rows = [r + [1] * (width - len(r)) for r in rows]
        return _Batch(input_ids=torch.tensor(rows, dtype=torch.long))


class FakeTextEncoder:
    """Deterministic stand-in for CLIPTextModel: hashed-id embeddings, `[0]` = hidden states, `.pooler_output` = mean."""
    dtype = None

    def __init__(self, dim):
        import torch
        self.dim = dim
        self.dtype = torch.float32

    def __call__(self, input_ids=None, **kw):
        import torch
        ids = input_ids
        g = torch.Generator().manual_seed(1234)
        table = torch.randn(4096, self.dim, generator=g)
        h = table[(ids.cpu() * 2654435761 % 4096).long()] + 0.1 * torch.arange(ids.shape[1])[None, :, None] / ids.shape[1]
        h = h.to(ids.device)

        class Out(tuple):
            pass
        out = Out((h,))
        out.pooler_output = h.mean(1)
        return out


def fake_vae_decode(latents):
    """Test stand-in for AutoencoderKL.decode + tensor2vid: (1,4,F,h,w) latents -> (1,F,8h,8w,3) floats in [0,1]."""
    import torch
    x = latents[:, :3].permute(0, 2, 3, 4, 1)  # (1,F,h,w,3)
    x = torch.sigmoid(x).repeat_interleave(8, 2).repeat_interleave(8, 3)
    return x.float().cpu().numpy()
