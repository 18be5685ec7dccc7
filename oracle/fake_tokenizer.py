"""TEST INFRASTRUCTURE: a deterministic word-level stand-in for the CLIP tokenizer (its vocab files are not on disk).
Same call surface the DSL front-end uses: tokenizer([text], padding=..., max_length=77, return_tensors="np")["input_ids"],
_convert_id_to_token, eos_token.  Words and punctuation marks are separate tokens, as in CLIP's BPE for common words."""
import re

import numpy as np


class FakeTokenizer:
    bos_token, eos_token = "<|startoftext|>", "<|endoftext|>"
    model_max_length = 77

    def __init__(self):
        self.vocab = {self.bos_token: 0, self.eos_token: 1}
        self.inv = {0: self.bos_token, 1: self.eos_token}

    def _id(self, tok):
        if tok not in self.vocab:
            self.vocab[tok] = len(self.vocab)
            self.inv[self.vocab[tok]] = tok
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
