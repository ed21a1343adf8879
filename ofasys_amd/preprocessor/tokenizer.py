"""Text -> dictionary-relative token ids for plain-text instruction spans and string columns.

The reference tokenises with GPT-2 byte-level BPE whose vocabulary / merge table it DOWNLOADS at run time
(preprocessor/tokenizer/gpt2_bpe.py:10-19, preprocessor/default/text.py:57-72) -- those assets cannot exist offline, so:

* `ByteLevelBPE(vocab_json, merges_txt)`: the real thing over user-supplied GPT-2 files (the `tokenizers` package does the
  merges); ids are GPT-2 ids, i.e. exactly the '<text>_<id>' symbols of the reference's dict.txt.
* `HashWordTokenizer(n)`: a deterministic stand-in (word / punctuation pieces -> crc32 % n) used when no BPE files are
  configured, so that string-valued instructions run end to end on synthetic data.  Not a linguistic tokenizer and not
  id-compatible with released checkpoints; token-level parity tests never go through it (fixtures start at token ids).
"""
import os
import re
import zlib
from typing import List

import numpy as np

N_GPT2 = 50260          # entries of the reference's dict.txt seen through '<text>_i' 


class HashWordTokenizer:
    _PIECES = re.compile(r"\w+|[^\w\s]")

    def __init__(self, n_ids: int = N_GPT2):
        self.n_ids = n_ids

    def encode(self, text: str) -> np.ndarray:
        return np.array([zlib.crc32(w.encode("utf-8")) % self.n_ids for w in self._PIECES.findall(text)], dtype=np.int64)

    def decode(self, ids) -> str:
        return " ".join(f"<{int(i)}>" for i in ids)


class ByteLevelBPE:
    def __init__(self, vocab_json: str, merges_txt: str):
        from tokenizers import ByteLevelBPETokenizer
        self._tok = ByteLevelBPETokenizer(vocab_json, merges_txt)
        self.n_ids = self._tok.get_vocab_size()

    def encode(self, text: str) -> np.ndarray:
        return np.array(self._tok.encode(text).ids, dtype=np.int64)

    def decode(self, ids) -> str:
        return self._tok.decode([int(i) for i in ids])


def build_tokenizer(bpe_dir: str = None):
    """ByteLevelBPE over `bpe_dir`/{encoder.json|vocab.json, vocab.bpe|merges.txt} (or $OFASYS_AMD_BPE_DIR), else the stand-in."""
    bpe_dir = bpe_dir or os.environ.get("OFASYS_AMD_BPE_DIR")
    if bpe_dir:
        for v, m in (("encoder.json", "vocab.bpe"), ("vocab.json", "merges.txt")):
            pv, pm = os.path.join(bpe_dir, v), os.path.join(bpe_dir, m)
            if os.path.exists(pv) and os.path.exists(pm):
                return ByteLevelBPE(pv, pm)
        raise FileNotFoundError(f"no GPT-2 BPE files (encoder.json + vocab.bpe) under {bpe_dir}")
    return HashWordTokenizer()


def text_infilling_noise(tokens: np.ndarray, ratio: float, mask_id: int, rng: np.random.Generator, poisson_lambda: float = 3.0,
                         random_ratio: float = 0.0, vocab_range=(4, 4 + N_GPT2)) -> np.ndarray:
    """BART text infilling as the reference configures it for `mask_ratio` slots (preprocessor/mask_utils.py:10-220 with
    span-poisson lengths, lambda 3, replace_length -1: every token of a chosen span becomes <mask>, nothing is deleted; a
    zero-length span INSERTS one <mask>).  About ceil(ratio * len) tokens are covered.  Train-time randomisation: the stream of
    random numbers is this function's own (numpy Generator), not torch's -- statistical, not bitwise, correspondence."""
    n = len(tokens)
    budget = int(np.ceil(n * ratio))
    if budget == 0 or n == 0:
        return tokens
    out = tokens.copy()
    spans, covered = [], 0
    while covered < budget:
        ln = int(rng.poisson(poisson_lambda))
        ln = min(ln, budget - covered)
        spans.append(ln)
        covered += max(ln, 0)
        if ln == 0 and len(spans) > 4 * budget:           # only zero-length draws left: stop
            break
    inserts = sum(1 for s in spans if s == 0)
    starts = rng.permutation(n)
    taken = np.zeros(n, dtype=bool)
    k = 0
    for ln in (s for s in spans if s > 0):
        while k < n and taken[starts[k]]:
            k += 1
        if k >= n:
            break
        a = int(starts[k])
        b = min(n, a + ln)
        taken[a:b] = True
        k += 1
    out[taken] = mask_id
    if random_ratio > 0:
        rnd = taken & (rng.random(n) < random_ratio)
        out[rnd] = rng.integers(vocab_range[0], vocab_range[1], size=int(rnd.sum()))
    if inserts:
        pos = np.sort(rng.integers(0, n + 1, size=inserts))
        out = np.insert(out, pos, mask_id)
    return out
