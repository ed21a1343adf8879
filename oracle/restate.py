"""CPU oracle: a plain-torch fp32 restatement of the OFASys encoder-decoder hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this file, and only as the *checker* / the timed CPU baseline.  The product
(`ofasys_amd`) never imports it and has no CPU fallback.

What it is: functional code over a flat `state` dict whose keys are the reference's state-dict
keys.  No nn.Module, no autograd tricks -- gradients come from
torch.autograd over these plain ops, which is what the reference itself does.  Every function
cites the reference file:line (relative to /root/reference/ofasys) it restates.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks this file against the golden vectors
in `tests/golden/*.npz`, which `oracle/gen_golden.py` produced by importing and running the
reference itself in the build container (the reference ships no tests of its own, SURVEY.md
section 4).

Scope: a1 Slot, a2 general adaptor dispatch, a3 adaptor post-hook,
a4 text adaptor (+ a5 box-as-tokens), a6 image_patch_embed adaptor, a10 abs/rel position bias
assembly, a11 encoder stack, a12 encoder layer, a13 attention (slow + fast path), a14 decoder,
a15 executor / normalized probs, and the CE criterion that closes the training step.
"""
import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# carrier types (a1) -- preprocessor/instruction.py:29-107, ofasys/__init__.py:29-38
# --------------------------------------------------------------------------------------------
MODALITY_ORDER = ["TEXT", "IMAGE", "BOX", "AUDIO", "MOTION", "PHONE", "VIDEO", "STRUCT", "CATEGORY"]

# adaptor/general.py:36-46
DEFAULT_ADAPTOR = {
    "TEXT": "text", "IMAGE": "image_resnet", "BOX": "text", "AUDIO": "audio_fbank", "PHONE": "text",
    "VIDEO": "video_image_sequence", "MOTION": "text", "STRUCT": "text", "CATEGORY": "text",
}


@dataclass
class OSlot:
    modality: str            # one of MODALITY_ORDER
    is_src: bool
    value: Any
    attributes: Optional[List[str]] = None

    def get_attr(self, key):  # preprocessor/instruction.py:73-83
        for a in self.attributes or []:
            if a.startswith(key + "="):
                return a[len(key) + 1:]
            if a == key:
                return ""
        return None


@dataclass
class OConfig:
    """The subset of GeneralistModelConfig (model/ofa.py:41-122) that shapes the math."""
    embed_dim: int
    ffn_dim: int
    heads: int
    enc_layers: int
    dec_layers: int
    attn_scale_factor: float = 2.0          # ofa.py:56-59
    use_self_attn_bias: bool = True         # ofa.py:110-113
    entangle_position_embedding: bool = False  # ofa.py:76-79 (model level)
    share_attn_bias: bool = False           # ofa.py:115-118
    pad_idx: int = 1                        # dictionary.py:21-53
    eps: float = 1e-5                       # module/layer_norm.py:27
    # per-adaptor flags (adaptor/base.py:56-81); NOT inherited from the model cfg (base.py:97-101)
    adaptor_entangle: Dict[str, bool] = field(default_factory=dict)
    adaptor_embed_scale: Dict[str, float] = field(default_factory=dict)       # sqrt(D) where no_scale_embedding=False (base.py:144)
    adaptor_grad_scale: Dict[str, float] = field(default_factory=dict)        # scale_embedding_gradient (base.py:174-176)
    patch: int = 14                         # adaptor/image_patch_embed.py:24-29
    resnet_layers: tuple = (3, 8, 36)       # adaptor/image_resnet.py:44-47 (default resnet152), module/resnet.py:249-261
    image_bucket_size: int = 42             # adaptor/image_resnet.py:62-65
    training: bool = False                  # BatchNorm batch statistics (dropout must be 0 for a deterministic oracle)
    modal_ffn: bool = False                 # ofa.py:119-121: one FFN expert per ModalityType
    activation_fn: str = "gelu"             # module/utils.py get_activation_fn (ofa.py: activation_fn)
    resnet_drop_path_rate: float = 0.0      # adaptor/image_resnet.py:49-52, module/resnet.py:114, 219-230
    drop_keep: Optional[list] = None        # the per-sample keep draws (0/1, [B] each) of the DropPath calls, in call order: random
                                            # INPUTS of a training-mode run, replayed (module/droppath.py:52-53)
    audio_mask_channel: str = ""            # "" (mask_channel_prob == 0) | "after" | "before" (adaptor/audio.py:452-466)
    enc_normalize_before: bool = True       # cfg.encoder.normalize_before (config/default_model.yaml; transformer_layer.py:47)
    dec_normalize_before: bool = True       # cfg.decoder.normalize_before (transformer_layer.py:265)


# --------------------------------------------------------------------------------------------
# small ops
# --------------------------------------------------------------------------------------------
def layer_norm(state, prefix, x, eps=1e-5):
    """torch.nn.LayerNorm(eps=1e-5, affine) -- module/layer_norm.py:27-32."""
    w, b = state[prefix + ".weight"], state[prefix + ".bias"]
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def linear(state, prefix, x):
    return F.linear(x, state[prefix + ".weight"], state.get(prefix + ".bias"))


def gelu(x):
    """module/gelu.py:18-19: exact-erf GELU computed in fp32 then cast back."""
    return F.gelu(x.float()).type_as(x)


def make_token_bucket_position(bucket_size, max_position):
    """adaptor/text.py:20-30 -- log-bucketed signed relative distance, integers (bit-exact)."""
    ctx = torch.arange(max_position, dtype=torch.long)[:, None]
    mem = torch.arange(max_position, dtype=torch.long)[None, :]
    rel = ctx - mem
    sign = torch.sign(rel)
    mid = bucket_size // 2
    abs_pos = torch.where((rel < mid) & (rel > -mid), torch.full_like(rel, mid - 1), torch.abs(rel))
    log_pos = torch.ceil(torch.log(abs_pos / mid) / math.log((max_position - 1) / mid) * (mid - 1)) + mid
    log_pos = log_pos.int()
    bucket = torch.where(abs_pos.le(mid), rel, log_pos * sign).long()
    return bucket + bucket_size - 1


def make_image_bucket_position(bucket_size, num_relative_distance):
    """2-D relative-position bucket table, integer, bit-exact (adaptor/image_resnet.py:25-40)."""
    coords = torch.stack(torch.meshgrid([torch.arange(bucket_size), torch.arange(bucket_size)], indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += bucket_size - 1
    rel[:, :, 1] += bucket_size - 1
    rel[:, :, 0] *= 2 * bucket_size - 1
    idx = torch.zeros(size=(bucket_size * bucket_size + 1,) * 2, dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = num_relative_distance - 3
    idx[0:, 0] = num_relative_distance - 2
    idx[0, 0] = num_relative_distance - 1
    return idx


def box_to_bins(coords, max_image_size, num_bins):
    """preprocessor/default/box.py:101-110 -- int((coord / max_image_size * (num_bins-1)).round()) evaluated on
    float32 tensors (round-half-to-even), restated with numpy float32 scalars."""
    import numpy as np
    return [int(np.round(np.float32(c) / np.float32(max_image_size) * np.float32(num_bins - 1))) for c in coords]


# --------------------------------------------------------------------------------------------
# adaptors (a3, a4, a6)
# --------------------------------------------------------------------------------------------
def _post_hook(state, cfg, side, name, slot, embed, pos_embed):
    """adaptor/base.py:152-181 (embed_scale == 1 by default: no_scale_embedding, base.py:69,144)."""
    p = f"{side}.adaptor.{name}"
    embed = cfg.adaptor_embed_scale.get(name, 1.0) * embed                # base.py:168
    if cfg.adaptor_entangle.get(name, False) and pos_embed is not None:   # base.py:170-171
        embed = embed + pos_embed
    if slot.is_src and (p + ".type_embedding.weight") in state:          # base.py:172-173
        embed = embed + state[p + ".type_embedding.weight"].squeeze()
    alpha = cfg.adaptor_grad_scale.get(name, 1.0)
    if alpha != 1.0:                                                      # base.py:174-176
        embed = embed * alpha + embed.detach() * (1 - alpha)
    if (p + ".layernorm_embedding.weight") in state:                      # base.py:177-178 (cfg.layernorm_embedding)
        embed = layer_norm(state, p + ".layernorm_embedding", embed, cfg.eps)
    if pos_embed is not None and (p + ".layernorm_position.weight") in state:          # base.py:179-180 (cfg.layernorm_position)
        pos_embed = layer_norm(state, p + ".layernorm_position", pos_embed, cfg.eps)
    return embed, pos_embed   # dropout (base.py:181) is identity in eval


def text_adaptor(state, cfg, side, slot):
    """adaptor/text.py:106-127 + get_rel_pos_bias :101-104 + base hook."""
    p = f"{side}.adaptor.text"
    tok = slot.value
    masks = tok.eq(cfg.pad_idx)                                           # text.py:119-122
    T = tok.shape[1]
    pos = torch.arange(T).unsqueeze(0).expand_as(tok)                     # module/utils.py:623-630
    pos_embed = F.embedding(pos, state[p + ".embed_positions.weight"])    # text.py:124
    embed = F.embedding(tok, state[f"{side}.adaptor.embed_tokens.weight"], padding_idx=cfg.pad_idx)
    embed, pos_embed = _post_hook(state, cfg, side, "text", slot, embed, pos_embed)
    rel = None
    if cfg.use_self_attn_bias:                                            # base.py:183-189
        L = cfg.enc_layers if side == "encoder" else cfg.dec_layers
        n_tab = 1 if cfg.share_attn_bias else L
        bucket = state[p + ".token_rp_bucket"][:T, :T]
        rel = []
        for l in range(n_tab):
            v = F.embedding(bucket, state[p + f".token_rel_pos_table_list.{l}.weight"])  # [T,T,A]
            rel.append(v.unsqueeze(0).expand(tok.shape[0], -1, -1, -1).permute(0, 3, 1, 2))  # base.py:242-256
    return embed, masks, pos_embed, rel


def image_patch_embed_adaptor(state, cfg, side, slot):
    """adaptor/image_patch_embed.py:62-80: Conv2d(3,D,k=p,s=p) -> flatten -> [B,N,D], cls token,
    learned positions, all-False mask, bias None (only valid with use_self_attn_bias=False)."""
    p = f"{side}.adaptor.image_patch_embed"
    img = slot.value
    B = img.shape[0]
    x = F.conv2d(img, state[p + ".proj.weight"], state[p + ".proj.bias"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)
    if (p + ".cls_token") in state:
        x = torch.cat((state[p + ".cls_token"].expand(B, -1, -1), x), dim=1)
    N = x.shape[1]
    masks = torch.zeros(B, N, dtype=torch.bool)
    pos = torch.arange(N).unsqueeze(0).expand(B, -1)
    pos_embed = F.embedding(pos, state[p + ".embed_image_positions.weight"])
    if cfg.use_self_attn_bias:
        raise NotImplementedError("image_patch_embed defines no get_rel_pos_bias (adaptor/base.py:183-189, 240)")
    embed, pos_embed = _post_hook(state, cfg, side, "image_patch_embed", slot, x, pos_embed)
    return embed, masks, pos_embed, None


def _bn(state, p, x, cfg):
    """nn.BatchNorm2d (module/resnet.py:105-128): batch statistics + running update when training, else running."""
    if cfg.training and (p + ".num_batches_tracked") in state:             # torch/nn/modules/batchnorm.py: _BatchNorm.forward
        state[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, state[p + ".running_mean"], state[p + ".running_var"], state[p + ".weight"], state[p + ".bias"],
                        training=cfg.training, momentum=0.1, eps=1e-5)


def _bottleneck(state, p, x, cfg, stride, has_down, drop=0.0):
    """module/resnet.py:112-137; drop > 0 in training: DropPath(drop, 0) on the residual branch (module/droppath.py:40-60)."""
    out = F.relu(_bn(state, p + ".bn1", F.conv2d(x, state[p + ".conv1.weight"]), cfg))
    out = F.relu(_bn(state, p + ".bn2", F.conv2d(out, state[p + ".conv2.weight"], stride=stride, padding=1), cfg))
    out = _bn(state, p + ".bn3", F.conv2d(out, state[p + ".conv3.weight"]), cfg)
    identity = x
    if has_down:
        identity = _bn(state, p + ".downsample.1", F.conv2d(x, state[p + ".downsample.0.weight"], stride=stride), cfg)
    if drop > 0.0 and cfg.training:                                       # x.div_(keep_prob); x * floor(keep_prob + U)
        out = out / (1.0 - drop) * cfg.drop_keep.pop(0).to(out.dtype).view(-1, 1, 1, 1)
    return F.relu(identity + out)


def resnet_backbone(state, p, x, cfg):
    """module/resnet.py:232-246: conv1-bn-relu-maxpool-layer1..3 (no layer4)."""
    x = F.relu(_bn(state, p + ".bn1", F.conv2d(x, state[p + ".conv1.weight"], stride=2, padding=3), cfg))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (blocks, stride) in enumerate(zip(cfg.resnet_layers, (1, 2, 2)), start=1):
        dpr = [v.item() for v in torch.linspace(0, cfg.resnet_drop_path_rate, blocks)]    # :219; block 0 is built without it (:207-217)
        for b in range(blocks):
            x = _bottleneck(state, f"{p}.layer{li}.{b}", x, cfg, stride if b == 0 else 1, b == 0, dpr[b] if b else 0.0)
    return x


def image_resnet_adaptor(state, cfg, side, slot):
    """adaptor/image_resnet.py:130-202: backbone -> flatten -> Linear(1024, D); position ids w + h*bucket + 1; rel-pos bias
    by the double gather into image_rp_bucket (:116-128)."""
    p = f"{side}.adaptor.image_resnet"
    img = slot.value
    B = img.shape[0]
    feat = resnet_backbone(state, p + ".embed_images", img, cfg)
    h, w = feat.shape[-2:]
    n = h * w
    masks = torch.zeros(B, n, dtype=torch.bool)
    idx = (torch.arange(w).unsqueeze(0).expand(h, w) + torch.arange(h).unsqueeze(1) * cfg.image_bucket_size + 1).view(-1)
    ids = idx[None, :].expand(B, n)
    embed = feat.flatten(2).transpose(1, 2)
    pos_embed = F.embedding(ids, state[p + ".embed_image_positions.weight"])
    embed = linear(state, p + ".image_proj", embed)
    rel = None
    if cfg.use_self_attn_bias:
        L = cfg.enc_layers if side == "encoder" else cfg.dec_layers
        bucket = state[p + ".image_rp_bucket"]
        S = bucket.size(1)
        rp = (bucket.unsqueeze(0).expand(B, S, S).gather(1, ids[:, :, None].expand(B, n, S))
              .gather(2, ids[:, None, :].expand(B, n, n)))
        rel = [F.embedding(rp, state[p + f".image_rel_pos_table_list.{l}.weight"]).permute(0, 3, 1, 2)
               for l in range(1 if cfg.share_attn_bias else L)]
    embed, pos_embed = _post_hook(state, cfg, side, "image_resnet", slot, embed, pos_embed)
    return embed, masks, pos_embed, rel


def video_image_sequence_adaptor(state, cfg, side, slot):
    """adaptor/video_image_sequence.py:111-208: frames through the image_resnet backbone + projection, image + frame
    positions, zero-frame padding rule, bias = frame rel-pos (+) image rel-pos broadcast to [B,A,F*P,F*P]."""
    p = f"{side}.adaptor.video_image_sequence"
    pi = f"{side}.adaptor.image_resnet"
    clips = slot.value.transpose(1, 2)                                                   # [B,F,3,H,W]
    B, Fr = clips.shape[:2]
    feat = resnet_backbone(state, pi + ".embed_images", clips.reshape(-1, *clips.shape[2:]), cfg)
    h, w = feat.shape[-2:]
    P = h * w
    T = P * Fr
    embed = feat.view(feat.size(0), feat.size(1), -1).transpose(1, 2).reshape(B, T, feat.size(1))
    masks = (clips.reshape(B, Fr, -1).abs().mean(dim=-1) == 0.0).unsqueeze(-1).expand(B, Fr, P).reshape(B, T)
    idx = (torch.arange(w).unsqueeze(0).expand(h, w) + torch.arange(h).unsqueeze(1) * cfg.image_bucket_size + 1).view(-1)
    ids = idx[None, :].expand(B, P)
    fids = (torch.arange(Fr) + 1)[None, :].expand(B, Fr)
    image_pos = F.embedding(ids, state[pi + ".embed_image_positions.weight"])
    frame_pos = F.embedding(fids, state[p + ".embed_frame_positions.weight"])
    pos_embed = (image_pos.unsqueeze(1) + frame_pos.unsqueeze(2)).reshape(B, T, -1)
    embed = linear(state, pi + ".image_proj", embed)
    rel = None
    if cfg.use_self_attn_bias:
        L = cfg.enc_layers if side == "encoder" else cfg.dec_layers
        bucket = state[pi + ".image_rp_bucket"]
        S = bucket.size(1)
        rp = (bucket.unsqueeze(0).expand(B, S, S).gather(1, ids[:, :, None].expand(B, P, S))
              .gather(2, ids[:, None, :].expand(B, P, P)))
        rel = []
        for l in range(L):
            vi = F.embedding(rp, state[pi + f".image_rel_pos_table_list.{l}.weight"]).permute(0, 3, 1, 2)   # [B,A,P,P]
            vf = F.embedding(state[p + ".video_rp_bucket"][:Fr, :Fr], state[p + f".video_rel_pos_table_list.{l}.weight"])
            vf = vf.transpose(1, 2).transpose(0, 1).contiguous()                                             # [A,F,F]
            vi = vi.view(vi.size(0), vi.size(1), 1, vi.size(2), 1, vi.size(3))
            vf = vf.view(vf.size(0), vf.size(1), 1, vf.size(1), 1)
            v = vf + vi
            rel.append(v.view(v.size(0), v.size(1), v.size(2) * v.size(3), v.size(4) * v.size(5)))
    embed, pos_embed = _post_hook(state, cfg, side, "video_image_sequence", slot, embed, pos_embed)
    return embed, masks, pos_embed, rel


def audio_fbank_adaptor(state, cfg, side, slot):
    """adaptor/audio.py:295-325 (source branch) + module/subsample.py:44-63: two Conv2d(3, stride 2)+ReLU, channel-major
    flatten, Linear; lengths ((l-1)/2+1).floor() twice; padding = positions past the subsampled length."""
    p = f"{side}.adaptor.audio_fbank"
    fbank, lengths = slot.value["fbank"], slot.value["fbank_lengths"]
    x = fbank.unsqueeze(1)
    x = F.relu(F.conv2d(x, state[p + ".subsample.conv.0.weight"], state[p + ".subsample.conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, state[p + ".subsample.conv.2.weight"], state[p + ".subsample.conv.2.bias"], stride=2))
    b, c, t, f = x.size()
    x = linear(state, p + ".subsample.out.0", x.transpose(1, 2).contiguous().view(b, t, c * f))
    out_len = lengths.clone()
    for _ in range(2):
        out_len = ((out_len.float() - 1) / 2 + 1).floor().long()
    masks = torch.zeros(b, t, dtype=torch.bool)
    for i, l in enumerate(out_len):                                                       # audio.py:307-310
        diff = int(l) - t
        if diff < 0:
            masks[i, diff:] = True
    pos_embed = F.embedding(torch.arange(t).unsqueeze(0).expand(b, t), state[p + ".embed_audio_positions.weight"])
    mi = slot.value.get("mask_indices")
    if mi is not None and slot.get_attr("use_mask") is not None:                          # apply_mask (:452-466), mask_prob > 0
        mch = slot.value["mask_channel_indices"].unsqueeze(1) if cfg.audio_mask_channel else None     # [B,1,C]: whole channels
        if cfg.audio_mask_channel == "before":
            x = x.masked_fill(mch, 0.0)
        x = torch.where(mi.unsqueeze(-1), state[p + ".mask_emb"].view(1, 1, -1), x)
        if cfg.audio_mask_channel == "after":
            x = x.masked_fill(mch, 0.0)
    embed, pos_embed = _post_hook(state, cfg, side, "audio_fbank", slot, x, pos_embed)
    rel = None
    if cfg.use_self_attn_bias:
        L = cfg.enc_layers if side == "encoder" else cfg.dec_layers
        bucket = state[p + ".audio_rp_bucket"][:t, :t]
        rel = [F.embedding(bucket, state[p + f".audio_rel_pos_table_list.{l}.weight"]).unsqueeze(0).expand(b, -1, -1, -1)
               .permute(0, 3, 1, 2) for l in range(1 if cfg.share_attn_bias else L)]
    return embed, masks, pos_embed, rel


_ADAPTORS = {"audio_fbank": audio_fbank_adaptor, "video_image_sequence": video_image_sequence_adaptor, "text": text_adaptor, "image_patch_embed": image_patch_embed_adaptor, "image_resnet": image_resnet_adaptor}


def general_adaptor(state, cfg, side, slots):
    """adaptor/general.py:120-158 (dispatch in ModalityType order, outputs kept in slot order) and
    concat :245-282 (abs-pos bias :223-243, per-layer clone + block-diagonal rel-pos add :270-280)."""
    outs = [None] * len(slots)
    for mod in MODALITY_ORDER:
        for i, s in enumerate(slots):
            if s.modality == mod:
                name = s.get_attr("adaptor") or DEFAULT_ADAPTOR[mod]       # general.py:103-118
                outs[i] = _ADAPTORS[name](state, cfg, side, s)
    embed = torch.cat([o[0] for o in outs], dim=1)
    masks = torch.cat([o[1] for o in outs], dim=1)
    pos_embed = torch.cat([o[2] for o in outs], dim=1)
    bias = None
    if cfg.use_self_attn_bias:
        B, T, D = pos_embed.shape
        A = cfg.heads
        if not cfg.entangle_position_embedding:
            pos_scaling = float(D / cfg.heads * cfg.attn_scale_factor) ** -0.5      # general.py:98
            pq = linear(state, f"{side}.adaptor.pos_q_linear", pos_embed).view(B, T, A, -1).transpose(1, 2) * pos_scaling
            pk = linear(state, f"{side}.adaptor.pos_k_linear", pos_embed).view(B, T, A, -1).transpose(1, 2)
            abs_bias = torch.matmul(pq, pk.transpose(2, 3))
        else:
            abs_bias = torch.zeros(B, A, T, T, dtype=pos_embed.dtype)               # general.py:234-242
        L = cfg.enc_layers if side == "encoder" else cfg.dec_layers
        bias = []
        for l in range(1 if cfg.share_attn_bias else L):
            b = abs_bias.clone()
            s0 = 0
            for o in outs:
                n = o[0].shape[1]
                if o[3] is not None and o[3][l] is not None:
                    b[:, :, s0:s0 + n, s0:s0 + n] = b[:, :, s0:s0 + n, s0:s0 + n] + o[3][l]
                s0 += n
            bias.append(b)
    if cfg.modal_ffn:
        general_adaptor.modal_mask = modal_mask_of([(s.modality, o[1].shape[0], o[1].shape[1]) for s, o in zip(slots, outs)])
    else:
        general_adaptor.modal_mask = None
    return embed, masks, pos_embed, bias


# --------------------------------------------------------------------------------------------
# attention (a13) -- module/multihead_attention.py
# --------------------------------------------------------------------------------------------
def mha_slow(state, prefix, cfg, query, key, key_padding_mask, attn_mask, attn_bias, need_head_weights=False):
    """multihead_attention.py:188-353.  query [T,B,D], key [S,B,D] (value == key for both uses).
    attn_bias [B*A,T,S] or None/False; attn_mask [T,S] additive (-inf upper triangle) or None."""
    T, B, D = query.shape
    A = cfg.heads
    hd = D // A
    scaling = float(hd * cfg.attn_scale_factor) ** -0.5                   # :54
    q = linear(state, prefix + ".q_proj", query) * scaling               # :199-218
    k = linear(state, prefix + ".k_proj", key)
    v = linear(state, prefix + ".v_proj", key)
    S = key.shape[0]
    q = q.contiguous().view(T, B * A, hd).transpose(0, 1)                 # :235-239
    k = k.contiguous().view(S, B * A, hd).transpose(0, 1)
    v = v.contiguous().view(S, B * A, hd).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))                                   # :308
    if attn_bias is not None and attn_bias is not False:
        w = w + attn_bias                                                 # :311-312
    if attn_mask is not None:
        w = torch.nan_to_num(w) + attn_mask.unsqueeze(0)                  # :314-317
    if key_padding_mask is not None:                                      # :319-326
        w = w.view(B, A, T, S).masked_fill(key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf"))
        w = w.view(B * A, T, S)
    p = F.softmax(w, dim=-1, dtype=torch.float32).type_as(w)              # module/utils.py:451-455
    o = torch.bmm(p, v)                                                   # :338 (attention dropout 0)
    o = o.transpose(0, 1).contiguous().view(T, B, D)
    c = state.get(prefix + ".c_attn")
    if c is not None:                                                     # :342-345
        o = (o.view(T, B, A, hd) * c.view(1, 1, A, 1)).reshape(T, B, D)
    o = linear(state, prefix + ".out_proj", o)                            # :346
    weights = p.view(B, A, T, S).transpose(1, 0) if need_head_weights else None   # :347-351
    return o, weights


def mha_fast(state, prefix, cfg, x, key_padding_mask):
    """multihead_attention.py:155-186: F.multi_head_attention_forward.  Uses head_dim**-0.5 scaling and
    ignores scale_factor and c_attn.  x [T,B,D], self-attention only."""
    T, B, D = x.shape
    A = cfg.heads
    hd = D // A
    q = linear(state, prefix + ".q_proj", x)
    k = linear(state, prefix + ".k_proj", x)
    v = linear(state, prefix + ".v_proj", x)
    q = q.view(T, B * A, hd).transpose(0, 1)
    k = k.view(T, B * A, hd).transpose(0, 1)
    v = v.view(T, B * A, hd).transpose(0, 1)
    w = torch.bmm(q * (1.0 / math.sqrt(hd)), k.transpose(1, 2))
    if key_padding_mask is not None:
        w = w.view(B, A, T, T).masked_fill(key_padding_mask.view(B, 1, 1, T), float("-inf")).view(B * A, T, T)
    p = F.softmax(w, dim=-1)
    o = torch.bmm(p, v).transpose(0, 1).contiguous().view(T, B, D)
    return linear(state, prefix + ".out_proj", o)


# --------------------------------------------------------------------------------------------
# layers (a12, a14) -- module/transformer_layer.py
# --------------------------------------------------------------------------------------------
def modal_mask_of(slots):
    """adaptor/general.py:143-146, 156-157: [B, T] int64, modality value - 1 over the columns of each slot (slot order).
    A token slot contributes its own width; callers pass the per-slot widths."""
    return torch.cat([torch.full((b, t), MODALITY_ORDER_VALUE[mod] - 1, dtype=torch.int64) for mod, b, t in slots], dim=1)


# ModalityType values (preprocessor/instruction.py:21-31)
MODALITY_ORDER_VALUE = {"TEXT": 1, "IMAGE": 2, "BOX": 3, "AUDIO": 4, "MOTION": 5, "PHONE": 6, "VIDEO": 7, "STRUCT": 8, "CATEGORY": 9}


def modal_for_ffn(state, p, modal_mask, x):
    """transformer_layer.py:116-130 + sparse_dispatcher.py:44-110 with one-hot gates: every row goes through the expert its gate
    selects.  The gates are indexed by the mask flattened BATCH-major (`modal_mask.view(bs * seq_len)`, :118-121) while the rows
    of x are flattened TIME-major (`x.view(seq_len * bs, dim)`, :124-125): row r takes the modality of mask.view(-1)[r].  (The
    reference's trailing x.half() is what restricts it to fp16 models; the oracle computes in fp32.)"""
    T, B, D = x.shape
    flat = modal_mask.reshape(-1)
    x2 = x.reshape(T * B, D)
    out = None
    for e in sorted(set(flat.tolist())):
        idx = (flat == e).nonzero().squeeze(1)
        y = linear(state, f"{p}.{e}", x2[idx])
        if out is None:
            out = torch.zeros(T * B, y.shape[1], dtype=y.dtype)
        out = out.index_add(0, idx, y)                                    # dispatcher.combine: zeros.index_add (:102-104)
    return out.view(T, B, -1)


def _ffn(state, p, cfg, x, modal_mask=None, pre=True):
    """transformer_layer.py:186-208 / :471-494 (scale_fc LayerNorm over F inside the FFN; scale_resids: the residual is scaled
    per channel by w_resid; post-LN layers normalise AFTER the residual add)."""
    r = x
    if pre:
        x = layer_norm(state, p + ".final_layer_norm", x, cfg.eps)
    act = {"gelu": gelu, "relu": F.relu, "linear": lambda t: t}[cfg.activation_fn]
    x = act(modal_for_ffn(state, p + ".experts_fc1", modal_mask, x) if cfg.modal_ffn else linear(state, p + ".fc1", x))
    if (p + ".ffn_layernorm.weight") in state:
        x = layer_norm(state, p + ".ffn_layernorm", x, cfg.eps)
    x = modal_for_ffn(state, p + ".experts_fc2", modal_mask, x) if cfg.modal_ffn else linear(state, p + ".fc2", x)
    if (p + ".w_resid") in state:                                         # :204-205 / :490-491
        r = state[p + ".w_resid"] * r
    x = r + x
    return x if pre else layer_norm(state, p + ".final_layer_norm", x, cfg.eps)       # :207-208 / :493-494


def encoder_layer(state, p, cfg, x, padding_mask, self_attn_bias, modal_mask=None):
    """transformer_layer.py:132-209 (dropout/droppath identity in eval)."""
    pre = cfg.enc_normalize_before
    r = x
    h = layer_norm(state, p + ".self_attn_layer_norm", x, cfg.eps) if pre else x
    if self_attn_bias is None:
        h = mha_fast(state, p + ".self_attn", cfg, h, padding_mask)
    else:
        h, _ = mha_slow(state, p + ".self_attn", cfg, h, h, padding_mask, None, self_attn_bias)
    if (p + ".attn_ln.weight") in state:
        h = layer_norm(state, p + ".attn_ln", h, cfg.eps)                 # :179-180
    x = r + h
    if not pre:
        x = layer_norm(state, p + ".self_attn_layer_norm", x, cfg.eps)    # :183-184
    return _ffn(state, p, cfg, x, modal_mask, pre)


def decoder_layer(state, p, cfg, x, enc, enc_padding_mask, self_attn_mask, self_attn_padding_mask,
                  self_attn_bias, cross_attn_bias, need_head_weights, modal_mask=None):
    """transformer_layer.py:351-495."""
    pre = cfg.dec_normalize_before
    r = x
    h = layer_norm(state, p + ".self_attn_layer_norm", x, cfg.eps) if pre else x
    h, _ = mha_slow(state, p + ".self_attn", cfg, h, h, self_attn_padding_mask, self_attn_mask, self_attn_bias)
    if (p + ".self_attn_ln.weight") in state:
        h = layer_norm(state, p + ".self_attn_ln", h, cfg.eps)            # :431-432
    x = r + h
    if not pre:
        x = layer_norm(state, p + ".self_attn_layer_norm", x, cfg.eps)    # :435-436
    r = x
    h = layer_norm(state, p + ".encoder_attn_layer_norm", x, cfg.eps) if pre else x   # :440-441
    h, cross_w = mha_slow(state, p + ".encoder_attn", cfg, h, enc, enc_padding_mask, None, cross_attn_bias,
                          need_head_weights=need_head_weights)            # :453-463 (static_kv -> slow path)
    if (p + ".cross_attn_ln.weight") in state:
        h = layer_norm(state, p + ".cross_attn_ln", h, cfg.eps)           # :464-465
    x = r + h
    if not pre:
        x = layer_norm(state, p + ".encoder_attn_layer_norm", x, cfg.eps)  # :468-469
    if modal_mask is not None:
        modal_mask = modal_mask[:x.shape[1], :x.shape[0]]                 # :477, 486
    return _ffn(state, p, cfg, x, modal_mask, pre), cross_w


# --------------------------------------------------------------------------------------------
# stacks (a11, a14, a15) -- model/transformer.py, model/ofa.py
# --------------------------------------------------------------------------------------------
def encoder_forward(state, cfg, slots, record=None):
    """model/transformer.py:78-156."""
    embed, masks, pos_embed, bias = general_adaptor(state, cfg, "encoder", slots)
    modal_mask = general_adaptor.modal_mask
    if record is not None:
        record["enc_embed"] = embed                                       # adaptor output, before pad zeroing
    has_pad = bool(masks.any())                                           # :110
    if has_pad:
        embed = embed * (1 - masks.unsqueeze(-1).type_as(embed))          # :111-112 (in place in the reference)
    x = embed.transpose(0, 1)
    T = x.shape[0]
    if record is not None:
        record["enc_pos_embed"] = pos_embed
        if bias is not None:
            record["enc_bias0"] = bias[0]
            record["enc_bias_last"] = bias[-1]
    for l in range(cfg.enc_layers):
        b = None
        if cfg.use_self_attn_bias:                                        # :121-126
            b = bias[0 if cfg.share_attn_bias else l].view(-1, T, T)
        x = encoder_layer(state, f"encoder.layers.{l}", cfg, x, masks if has_pad else None, b, modal_mask)
        if record is not None:
            record[f"enc_layer{l}"] = x
    if "encoder.layer_norm.weight" in state:                              # only a pre-LN stack has it (transformer.py:61-64)
        x = layer_norm(state, "encoder.layer_norm", x, cfg.eps)           # :142-143
    return {"encoder_out": x, "encoder_padding_mask": masks, "position_embeddings": pos_embed}


def decoder_forward(state, cfg, slots, enc_out, record=None):
    """model/transformer.py:365-522 + forward :349-363 (tied output projection adaptor/text.py:94-96,142)."""
    embed, masks, pos_embed, bias = general_adaptor(state, cfg, "decoder", slots)
    modal_mask = general_adaptor.modal_mask
    B, Tt, D = embed.shape
    A = cfg.heads
    enc = enc_out["encoder_out"]
    cross_bias = None
    if not cfg.entangle_position_embedding:                               # :441-445 -> :280-299
        src_pos = enc_out["position_embeddings"]
        Ts = src_pos.shape[1]
        pos_scaling = float(D / cfg.heads * cfg.attn_scale_factor) ** -0.5   # decoder adaptor's, :291
        pq = linear(state, "decoder.cross_pos_q_linear", pos_embed).view(B, Tt, A, -1).transpose(1, 2) * pos_scaling
        pk = linear(state, "decoder.cross_pos_k_linear", src_pos).view(B, Ts, A, -1).transpose(1, 2)
        cross_bias = torch.matmul(pq, pk.transpose(2, 3)).reshape(-1, Tt, Ts)
    x = embed.transpose(0, 1)
    future = torch.triu(torch.full((Tt, Tt), float("-inf")), 1)           # :528-539
    attn = None
    for l in range(cfg.dec_layers):
        if cfg.use_self_attn_bias:                                        # :468-475
            sb = bias[0 if cfg.share_attn_bias else l].view(-1, Tt, Tt)
        else:
            sb = False                                                    # :477 (forces the slow path)
        last = l == cfg.dec_layers - 1                                    # alignment_layer default, :421-422
        x, cw = decoder_layer(state, f"decoder.layers.{l}", cfg, x, enc, enc_out["encoder_padding_mask"],
                              future, masks, sb, cross_bias, need_head_weights=last, modal_mask=modal_mask)
        if record is not None:
            record[f"dec_layer{l}"] = x
        if last:
            attn = cw.float().mean(dim=0)                                 # :498-506 (names swapped upstream: this IS cross-attn)
    if "decoder.layer_norm.weight" in state:                              # (transformer.py:255-258)
        x = layer_norm(state, "decoder.layer_norm", x, cfg.eps)           # :508-509
    x = x.transpose(0, 1)
    logits = F.linear(x, state["decoder.adaptor.embed_tokens.weight"])    # adaptor/base.py:131
    return logits, {"attn": attn, "last_hidden_state": x}


def model_forward(state, cfg, slots, record=None):
    """model/ofa.py:165-285: split by is_src, encoder then decoder."""
    enc = encoder_forward(state, cfg, [s for s in slots if s.is_src], record)
    if record is not None:
        record["encoder_out"] = enc["encoder_out"]
    logits, extra = decoder_forward(state, cfg, [s for s in slots if not s.is_src], enc, record)
    return logits, extra


# --------------------------------------------------------------------------------------------
# incremental decoding: multihead_attention.py:188-279 with incremental_state,
# transformer_layer.py:386-470, model/transformer.py:447-490, beam reorder multihead_attention.py:393-409
# --------------------------------------------------------------------------------------------
def mha_step(state, prefix, cfg, query, key, key_padding_mask, attn_bias, cache, static_kv, need_head_weights=False):
    """One decoding step.  query [1,B,D]; cache: dict with prev_key / prev_value [B,A,n,hd] and prev_key_padding_mask,
    updated in place (saved_state, :254-279)."""
    T, B, D = query.shape
    A = cfg.heads
    hd = D // A
    scaling = float(hd * cfg.attn_scale_factor) ** -0.5
    q = linear(state, prefix + ".q_proj", query) * scaling
    q = q.contiguous().view(T, B * A, hd).transpose(0, 1)
    if static_kv and "prev_key" in cache:                                 # :188-195: keys are static, nothing to compute
        k = cache["prev_key"].view(B * A, -1, hd)
        v = cache["prev_value"].view(B * A, -1, hd)
        kpm = cache.get("prev_key_padding_mask")
    else:
        src = key if static_kv else query
        k = linear(state, prefix + ".k_proj", src)
        v = linear(state, prefix + ".v_proj", src)
        k = k.contiguous().view(-1, B * A, hd).transpose(0, 1)
        v = v.contiguous().view(-1, B * A, hd).transpose(0, 1)
        kpm = key_padding_mask
        if "prev_key" in cache:                                           # :244-262 concatenate with the cached steps
            k = torch.cat([cache["prev_key"].view(B * A, -1, hd), k], dim=1)
            v = torch.cat([cache["prev_value"].view(B * A, -1, hd), v], dim=1)
            prev_m = cache.get("prev_key_padding_mask")
            S_ = k.shape[1]
            if prev_m is not None and kpm is not None:                    # :356-391
                kpm = torch.cat([prev_m.float(), kpm.float()], dim=1)
            elif prev_m is not None:
                kpm = torch.cat([prev_m.float(), torch.zeros(B, S_ - prev_m.shape[1])], dim=1)
            elif kpm is not None:
                kpm = torch.cat([torch.zeros(B, S_ - kpm.shape[1]), kpm.float()], dim=1)
    S = k.shape[1]
    cache["prev_key"] = k.view(B, A, S, hd)
    cache["prev_value"] = v.view(B, A, S, hd)
    cache["prev_key_padding_mask"] = kpm
    w = torch.bmm(q, k.transpose(1, 2))
    if attn_bias is not None and attn_bias is not False:
        w = w + attn_bias
    if kpm is not None:
        w = w.view(B, A, T, S).masked_fill(kpm.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf")).view(B * A, T, S)
    p = F.softmax(w, dim=-1, dtype=torch.float32).type_as(w)
    o = torch.bmm(p, v).transpose(0, 1).contiguous().view(T, B, D)
    c = state.get(prefix + ".c_attn")
    if c is not None:
        o = (o.view(T, B, A, hd) * c.view(1, 1, A, 1)).reshape(T, B, D)
    o = linear(state, prefix + ".out_proj", o)
    return o, (p.view(B, A, T, S).transpose(1, 0) if need_head_weights else None)


def decoder_step(state, cfg, slots, enc_out, inc):
    """One incremental decoder call: `slots` hold the WHOLE target prefix (the adaptor embeds it all, the layers only see
    its last position, model/transformer.py:447-450); `inc` = {(layer, "self"|"cross"): cache}.  Returns logits [B,1,V]."""
    embed, masks, pos_embed, bias = general_adaptor(state, cfg, "decoder", slots)
    B, Tt, D = embed.shape
    A = cfg.heads
    enc = enc_out["encoder_out"]
    cross_bias = None
    if not cfg.entangle_position_embedding:
        src_pos = enc_out["position_embeddings"]
        Ts = src_pos.shape[1]
        pos_scaling = float(D / cfg.heads * cfg.attn_scale_factor) ** -0.5
        pq = linear(state, "decoder.cross_pos_q_linear", pos_embed).view(B, Tt, A, -1).transpose(1, 2) * pos_scaling
        pk = linear(state, "decoder.cross_pos_k_linear", src_pos).view(B, Ts, A, -1).transpose(1, 2)
        cross_bias = torch.matmul(pq, pk.transpose(2, 3)).reshape(-1, Tt, Ts)[:, -1:, :]
    x = embed[:, -1:].transpose(0, 1)
    m = masks[:, -1:]
    attn = None
    for l in range(cfg.dec_layers):
        p = f"decoder.layers.{l}"
        sb = bias[0 if cfg.share_attn_bias else l].view(-1, Tt, Tt)[:, -1:, :] if cfg.use_self_attn_bias else False
        r = x
        h = layer_norm(state, p + ".self_attn_layer_norm", x, cfg.eps)
        h, _ = mha_step(state, p + ".self_attn", cfg, h, h, m, sb, inc.setdefault((l, "self"), {}), False)
        if (p + ".self_attn_ln.weight") in state:
            h = layer_norm(state, p + ".self_attn_ln", h, cfg.eps)
        x = r + h
        r = x
        h = layer_norm(state, p + ".encoder_attn_layer_norm", x, cfg.eps)
        last = l == cfg.dec_layers - 1
        h, cw = mha_step(state, p + ".encoder_attn", cfg, h, enc, enc_out["encoder_padding_mask"], cross_bias,
                         inc.setdefault((l, "cross"), {}), True, need_head_weights=last)
        if (p + ".cross_attn_ln.weight") in state:
            h = layer_norm(state, p + ".cross_attn_ln", h, cfg.eps)
        x = _ffn(state, p, cfg, r + h)
        if last:
            attn = cw.float().mean(dim=0)
    x = layer_norm(state, "decoder.layer_norm", x, cfg.eps).transpose(0, 1)
    return F.linear(x, state["decoder.adaptor.embed_tokens.weight"]), {"attn": attn}


def reorder_incremental_state(inc, new_order):
    """multihead_attention.py:393-409: every cached tensor follows new_order along the batch; a static (cross) cache whose
    batch size already equals len(new_order) is left untouched."""
    for (l, kind), cache in inc.items():
        for k in list(cache.keys()):
            t = cache[k]
            if t is None:
                continue
            if kind == "cross" and t.size(0) == new_order.size(0):
                break
            cache[k] = t.index_select(0, new_order)
    return inc


def reorder_encoder_out(enc_out, new_order):
    """model/transformer.py:158-196 on the fields this restatement carries."""
    return {"encoder_out": enc_out["encoder_out"].index_select(1, new_order),
            "encoder_padding_mask": enc_out["encoder_padding_mask"].index_select(0, new_order),
            "position_embeddings": enc_out["position_embeddings"].index_select(0, new_order)}


def cross_entropy(logits, target, pad_idx=1):
    """engine/criterion/cross_entropy.py:50-67, 27-41: fp32 log-softmax, NLL sum, ignore pad;
    sample_size = number of non-pad targets."""
    lprobs = F.log_softmax(logits.float(), dim=-1)
    loss = F.nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=pad_idx, reduction="sum")
    return loss, int(target.ne(pad_idx).sum())


# --------------------------------------------------------------------------------------------
# train-step update arithmetic -- pinned by tests/golden/trainstep.npz
# --------------------------------------------------------------------------------------------
def multiply_clip(grads, sample_size, clip_norm, world=1):
    """engine/trainer.py:849-860 `multiply_grads(world / sum(task_sample_size))` (after DDP's division by world the net factor on
    the SUM of gradients is 1/sum sample_size) + module/utils.py:342-384 `clip_grad_norm_`: total_norm = || [ ||g||_2 per tensor ] ||_2
    in fp32, clip_coef = clamp(max_norm / (total_norm + 1e-6), max=1).  grads: {key: tensor or None}, modified copies returned.
    -> (grads, total_norm, clip_coef)"""
    c = world / (sample_size or 1.0) / world
    out = {k: (None if g is None else g.detach().float() * c) for k, g in grads.items()}
    norms = [torch.norm(g, p=2, dtype=torch.float32) for g in out.values() if g is not None]
    total = torch.norm(torch.stack(norms)) if norms else torch.tensor(0.0)
    coef = 1.0
    if clip_norm > 0:
        coef = float((clip_norm / (total + 1e-6)).clamp(max=1))
        out = {k: (None if g is None else g * coef) for k, g in out.items()}
    return out, float(total), coef


def adam_update(param, grad, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay):
    """engine/optim/adam.py:192-212 (class `Adam`, the AdamW-style decoupled decay), one tensor, in place; `step` is the
    1-based update count.  fp32 throughout (the reference keeps fp32 copies of low-precision params, :163-165, :214-215)."""
    b1, b2 = betas
    exp_avg.mul_(b1).add_(grad, alpha=1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    denom = exp_avg_sq.sqrt().add_(eps)
    step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    if weight_decay != 0:
        param.add_(param, alpha=-weight_decay * lr)
    param.addcdiv_(exp_avg, denom, value=-step_size)


def loss_scaler_step(state, raw_norm, sample_size, clip_norm, scale_factor=2.0, scale_window=2000, tolerance=0.0, threshold=None,
                     min_loss_scale=1e-4):
    """One update of the fp16 optimizer's scalar arithmetic around the dynamic loss scaler: engine/optim/fp16_optimizer.py:170-204,
    228-229 + dynamic_loss_scaler.py:9-70.  state: dict(loss_scale, iter, last_overflow_iter, last_rescale_iter,
    overflows_since_rescale), modified in place.  raw_norm: ||g|| of the loss-scaled, summed gradients.
    -> (status "ok" | "overflow" | "fatal", multiply_factor applied to the gradients (0 when the update is skipped))"""
    mf = 1.0 / state["loss_scale"] / sample_size
    grad_norm = mf * raw_norm
    if grad_norm > clip_norm > 0.0:
        mf *= clip_norm / grad_norm
    if grad_norm == float("inf") or grad_norm != grad_norm:              # check_overflow :44-70
        prev = state["loss_scale"]
        since = state["iter"] - state["last_rescale_iter"]
        state["last_overflow_iter"] = state["iter"]
        state["overflows_since_rescale"] += 1
        if state["overflows_since_rescale"] / float(since) >= tolerance:
            state["loss_scale"] /= scale_factor
            if threshold is not None:
                state["loss_scale"] = max(state["loss_scale"], threshold)
            state["last_rescale_iter"] = state["iter"]
            state["overflows_since_rescale"] = 0
        if state["loss_scale"] <= min_loss_scale:
            state["loss_scale"] = prev
            return "fatal", 0.0
        state["iter"] += 1
        return "overflow", 0.0
    if (state["iter"] - state["last_overflow_iter"]) % scale_window == 0:   # update :33-37
        state["loss_scale"] *= scale_factor
        state["last_rescale_iter"] = state["iter"]
    state["iter"] += 1
    return "ok", mf


def new_loss_scaler(init_scale=2.0 ** 15):
    """dynamic_loss_scaler.py:10-28"""
    return {"loss_scale": float(init_scale), "iter": 0, "last_overflow_iter": -1, "last_rescale_iter": -1, "overflows_since_rescale": 0}


# --------------------------------------------------------------------------------------------
# the reference's fused-softmax extensions restated
# --------------------------------------------------------------------------------------------
def label_smoothed_cross_entropy(logits, target, eps, pad_idx=1, constraint_range=None, constraint_masks=None,
                                 drop_worst_ratio=0.0):
    """engine/criterion/label_smoothed_cross_entropy.py: get_constraint_masks :140-151, get_lprobs_and_target :153-169
    (masked logits -> -inf, fp32 log-softmax), compute_loss :171-189 (pad rows removed), label_smoothed_nll_loss :62-94.
    Returns (loss_sum, nll_sum, ntokens)."""
    V = logits.shape[-1]
    cm = None
    if constraint_range is not None:
        cm = torch.ones(logits.shape, dtype=torch.bool)
        cm[..., 4:constraint_range[0]] = 0
        cm[..., constraint_range[1]:] = 0
        if constraint_masks is not None:
            cm = torch.logical_and(constraint_masks, cm)
    elif constraint_masks is not None:
        cm = constraint_masks
    x = logits if cm is None else logits.masked_fill(~cm, -math.inf)
    lprobs = F.log_softmax(x.float(), dim=-1).view(-1, V)
    target = target.reshape(-1)
    keep = target != pad_idx
    if cm is not None:
        cm = cm.reshape(-1, V)[keep]
    lprobs, target = lprobs[keep], target[keep]
    nll = -lprobs.gather(dim=-1, index=target.unsqueeze(-1)).squeeze(-1)
    if cm is not None:
        smooth = -lprobs.masked_fill(~cm, 0).sum(dim=-1)
        eps_i = eps / (cm.sum(1) - 1 + 1e-6)
    else:
        smooth = -lprobs.sum(dim=-1)
        eps_i = eps / (V - 1)
    loss = (1.0 - eps - eps_i) * nll + eps_i * smooth
    if drop_worst_ratio > 0:
        loss, idx = torch.topk(loss, k=int(loss.shape[0] * (1 - drop_worst_ratio)), largest=False)
        nll = nll[idx]
    return loss.sum(), nll.sum(), loss.numel()


def scaled_softmax(x, scale):
    """fused_kernels/scaled_masked_softmax.h:98-201: y = softmax(scale*x) over the last dim, fp32 accumulate."""
    return F.softmax(x.float() * scale, dim=-1).type_as(x)


def scaled_softmax_bwd(dy, y, scale):
    """scaled_masked_softmax.h:329-423: dx = scale * (dy*y - y*sum(dy*y))."""
    dy, y = dy.float(), y.float()
    return (scale * (dy * y - y * (dy * y).sum(-1, keepdim=True)))


def scaled_masked_softmax(x, mask, scale):
    """scaled_masked_softmax.h:209-327: masked (mask==1) positions are REPLACED by -10000.0 (:269-273)."""
    z = x.float() * scale
    z = torch.where(mask.bool(), torch.full_like(z, -10000.0), z)
    return F.softmax(z, dim=-1).type_as(x)


def scaled_upper_triang_masked_softmax(x, scale):
    """scaled_upper_triang_masked_softmax.h:113-230: causal softmax on [attn_batches, sq, sq]; masked outputs are 0."""
    sq = x.shape[-1]
    z = x.float() * scale
    m = torch.triu(torch.ones(sq, sq, dtype=torch.bool), 1)
    z = z.masked_fill(m, float("-inf"))
    return F.softmax(z, dim=-1).type_as(x)


def get_batch_per_block(sq, sk, b, np_):
    """scaled_masked_softmax.h:426-438 (32-lane-warp arithmetic of the reference, kept for API parity)."""
    log2 = 0
    while (1 << log2) < sk:
        log2 += 1
    pow2 = 1 << log2
    warp_size = pow2 if pow2 < 32 else 32
    batches_per_warp = 2 if pow2 <= 128 else 1
    warps_per_block = 128 // warp_size
    return warps_per_block * batches_per_warp
