"""Helpers shared by the oracle tests (CPU) and the HIP parity tests (GPU)."""
import os

import numpy as np
import torch

from oracle import recipe
from oracle.cases import CASES, VOCAB_EXTRA, make_target, make_value
from oracle.restate import OConfig, OSlot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCH = {  # model/ofa.py:557-610
    "tiny": dict(embed_dim=256, ffn_dim=1024, heads=4, enc_layers=4, dec_layers=4),
    "base": dict(embed_dim=768, ffn_dim=3072, heads=12, enc_layers=6, dec_layers=6),
    "large": dict(embed_dim=1024, ffn_dim=4096, heads=16, enc_layers=12, dec_layers=12),
}


def load_golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False))


def state_from_golden(g, dtype=torch.float32):
    """Rebuild the full state dict (reference key schema) from the recipe; integer buffers are recomputed."""
    from oracle.restate import make_token_bucket_position
    state = {}
    bucket = audio_bucket = None
    for item in g["state_keys"]:
        key, shape, _ = str(item).split("|")
        shape = tuple(int(x) for x in shape.strip("()").split(",") if x.strip())
        if key.endswith("version"):
            state[key] = torch.tensor([3.0])
        elif key.endswith("audio_rp_bucket"):
            if audio_bucket is None:
                audio_bucket = make_token_bucket_position(1024, 4096)      # adaptor/audio.py:50-60, bucket = max_position
            state[key] = audio_bucket
        elif key.endswith("video_rp_bucket"):
            state[key] = make_token_bucket_position(256, 1024)
        elif key.endswith("token_rp_bucket"):
            if bucket is None:
                bucket = make_token_bucket_position(256, 1024)
            state[key] = bucket
        elif key.endswith("image_rp_bucket"):
            from oracle.restate import make_image_bucket_position
            state[key] = make_image_bucket_position(42, (2 * 42 - 1) ** 2 + 3)
        elif key.endswith("num_batches_tracked"):
            state[key] = torch.zeros((), dtype=torch.long)
        else:
            state[key] = recipe.value_for(key, shape).to(dtype)
    return state


def oracle_cfg(case):
    ov = case["overrides"]
    ent = {k: v.get("entangle_position_embedding", False) for k, v in case["adaptor_overrides"].items()}
    D = ARCH[case["arch"]]["embed_dim"]
    esc = {k: float(D) ** 0.5 for k, v in case["adaptor_overrides"].items() if v.get("no_scale_embedding", True) is False}
    gsc = {k: float(v["scale_embedding_gradient"]) for k, v in case["adaptor_overrides"].items() if "scale_embedding_gradient" in v}
    layers = {"resnet50": (3, 4, 6), "resnet101": (3, 4, 23), "resnet152": (3, 8, 36)}[
        case["adaptor_overrides"].get("image_resnet", {}).get("resnet_type", "resnet152")]
    return OConfig(**ARCH[case["arch"]], use_self_attn_bias=ov.get("use_self_attn_bias", True),
                   entangle_position_embedding=ov.get("entangle_position_embedding", False), adaptor_entangle=ent,
                   adaptor_embed_scale=esc, adaptor_grad_scale=gsc,
                   resnet_layers=layers, training=bool(case.get("train", False)), modal_ffn=bool(ov.get("modal_ffn", False)),
                   activation_fn=ov.get("activation_fn", "gelu"),
                   share_attn_bias=ov.get("share_attn_bias", False), attn_scale_factor=float(ov.get("attn_scale_factor", 2.0)),
                   enc_normalize_before=ov.get("encoder_normalize_before", True),
                   dec_normalize_before=ov.get("decoder_normalize_before", True),
                   resnet_drop_path_rate=float(case["adaptor_overrides"].get("image_resnet", {}).get("resnet_drop_path_rate", 0.0)),
                   audio_mask_channel=_audio_mask_channel(case["adaptor_overrides"].get("audio_fbank", {})))


def _audio_mask_channel(ov):
    if ov.get("mask_channel_prob", 0.0) <= 0:
        return ""
    return "before" if ov.get("mask_channel_before", False) else "after"


def drop_keep_rows(g):
    """The recorded keep draws of a drop-path case, one [B] float tensor per DropPath call in call order."""
    return [torch.from_numpy(r.copy()) for r in g["droppath_keep"]] if "droppath_keep" in g else None


class replay_drop_path:
    """Context: ofasys_amd's per-sample drop-path draw (ops._drop_path_uniform) answers with the recorded keep decisions instead
    of fresh random numbers (u = 0.999 keeps a sample for any keep probability used here, u = 0 drops it)."""

    def __init__(self, keep_rows):
        self.rows = list(keep_rows or [])

    def __enter__(self):
        from ofasys_amd import ops
        self.ops, self.orig = ops, ops._drop_path_uniform
        if self.rows:
            rows = self.rows

            def replay(B, device):
                r = rows.pop(0)
                assert r.numel() == B
                return (r.float() * 0.999).to(device)
            ops._drop_path_uniform = replay
        return self

    def __exit__(self, *exc):
        self.ops._drop_path_uniform = self.orig
        if exc[0] is None:
            assert not self.rows, f"{len(self.rows)} recorded drop-path draws were not consumed"
        return False


def case_inputs(case):
    V = 4 + VOCAB_EXTRA
    vals, prev = [], None
    for mod, is_src, spec, attrs in case["slots"]:
        v = make_value(spec, V)
        vals.append((mod, is_src, v, attrs))
        if not is_src:
            prev = v
    return vals, make_target(prev)


def oracle_slots(vals):
    return [OSlot(m, s, v, a) for m, s, v, a in vals]


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def oracle_state_for(model):
    """Oracle state (reference key schema, fp32, CPU) for a built ofasys_amd model: float entries from the shared recipe (the
    model itself was filled from the same recipe by tests/model_util.build_model), integer buffers copied from the model (their
    bit-exactness is tested separately), the tied decoder embedding aliased to the encoder's as in the reference."""
    state = {}
    for k, v in model.state_dict().items():
        if k.endswith("version"):
            state[k] = torch.tensor([3.0])
        elif k.endswith("num_batches_tracked"):
            state[k] = torch.zeros((), dtype=torch.long)
        elif not v.is_floating_point():
            state[k] = v.detach().cpu().clone()
        else:
            state[k] = recipe.value_for(k, tuple(v.shape))
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    return state


def oracle_params(state):
    """Leaf tensors of an oracle state that take gradients (the shared embedding once)."""
    out = {}
    for k, v in state.items():
        if v.is_floating_point() and not k.endswith(("version", "running_mean", "running_var")) \
                and not k.startswith("decoder.adaptor.embed_tokens"):
            out[k] = v.requires_grad_(True)
    return out
