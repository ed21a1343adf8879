"""Deterministic weight / input recipe shared by the golden generator and the parity tests.

TEST INFRASTRUCTURE.  Golden fixtures do not store model weights (base = 145 M parameters); both the
reference (in `oracle/gen_golden.py`) and the implementation under test fill every state-dict entry
from this recipe, keyed by the state-dict key, so "identical inputs" holds by construction.

Values are deliberately NON-trivial (the reference's own init zeroes the rel-pos tables and the cls
token and sets c_attn / LayerNorm weights to one, which would hide bias / scaling bugs).
numpy's Philox generator is used because its stream is stable across numpy versions and hosts.
"""
import zlib

import numpy as np
import torch

_SKIP = ("version", "token_rp_bucket", "image_rp_bucket", "video_rp_bucket", "audio_rp_bucket", "num_batches_tracked")


def _gen(key, shape, scale, shift=0.0):
    g = np.random.Generator(np.random.Philox(key=zlib.crc32(key.encode())))
    return torch.from_numpy((g.standard_normal(tuple(shape), dtype=np.float32) * scale + shift).astype(np.float32))


def value_for(key, shape, pad_idx=1):
    """Recipe value for one state-dict entry (fp32)."""
    last = key.split(".")[-1]
    if key.endswith("c_attn"):
        return _gen(key, shape, 0.1, 1.0)
    if last == "running_mean":                        # BatchNorm buffers (image_resnet backbone)
        return _gen(key, shape, 0.1)
    if last == "running_var":
        return _gen(key, shape, 0.1).abs() + 0.8
    if "rel_pos_table" in key:
        return _gen(key, shape, 0.1)
    if key.endswith("cls_token"):
        return _gen(key, shape, 0.5)
    if len(shape) == 1 and last == "weight":          # LayerNorm gains
        return _gen(key, shape, 0.1, 1.0)
    if last == "bias":
        return _gen(key, shape, 0.02)
    if "embed_tokens" in key:
        w = _gen("shared.embed_tokens.weight", shape, 0.1)   # encoder/decoder/output projection share one matrix
        w[pad_idx].zero_()
        return w
    if "embed_positions" in key or "embed_image_positions" in key or "embed_frame_positions" in key or "embed_audio_positions" in key or "type_embedding" in key:
        return _gen(key, shape, 0.5)
    if "proj.weight" in key and len(shape) == 4:      # patch conv
        return _gen(key, shape, 0.05)
    return _gen(key, shape, 0.05)


def fill_state(state_dict):
    """In-place fill of a state dict (tensors keep their dtype/device); returns the same dict."""
    with torch.no_grad():
        for k, v in state_dict.items():
            if k.split(".")[-1] in _SKIP:
                continue
            v.copy_(value_for(k, v.shape).to(dtype=v.dtype, device=v.device))
    return state_dict


def tokens(key, shape, vocab, lengths=None, pad_idx=1, bos=None):
    """Token ids U[4, vocab) with optional right padding per row (and optional leading bos)."""
    g = np.random.Generator(np.random.Philox(key=zlib.crc32(key.encode())))
    t = torch.from_numpy(g.integers(4, vocab, size=shape, dtype=np.int64))
    if bos is not None:
        t[:, 0] = bos
    if lengths is not None:
        for r, n in enumerate(lengths):
            t[r, n:] = pad_idx
    return t


def floats(key, shape, scale=1.0):
    return _gen(key, shape, scale)
