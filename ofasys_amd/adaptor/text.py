"""TextAdaptor (reference: adaptor/text.py:20-142): token + learned-position embedding, 1-D log-bucketed relative
position bias tables (one per layer), tied output projection.  BOX / STRUCT / MOTION / PHONE / CATEGORY slots route
here too (adaptor/general.py:36-46): boxes are <bin>_k vocabulary tokens."""
import math
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..configure import register_config
from ..module import Embedding, OfaLinear
from ..preprocessor import Dictionary, Slot
from .base import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig


def make_token_bucket_position(bucket_size, max_position):
    """The reference's [max_position, max_position] int64 table of 1-D relative-position buckets (adaptor/text.py:20-30; the table is a
    buffer of the state dict, so it has to come out bit for bit -- pinned by the CRC in tests/golden/tiny_text.npz and, at the audio
    adaptor's sizes, against the recorded state dict).  Built from what the table IS rather than as the reference's 2-D tensor
    expression: the bucket of (i, j) depends on the distance d = i - j alone; distances up to mid = bucket_size / 2 keep their own
    bucket, beyond that bucket mid + n covers the geometric band
        mid * r^((n-1)/(mid-1)) < |d| <= mid * r^(n/(mid-1)),   r = (max_position - 1) / mid,   n = ceil((mid-1) * ln(|d|/mid) / ln r),
    mirrored for negative d and shifted by bucket_size - 1 (indices 0 .. 2 * bucket_size - 2): ONE 1-D lookup over |d|, gathered by
    |i - j|.  The band index is evaluated in float32 in the reference's operation order, because the table is DEFINED by that rounding:
    at (1024, 4096), the audio adaptor's sizes, d = 3396 has (mid-1) * ln(d/mid) / ln r = 465.0000008 -- float32 rounds it to 465.0 and
    the reference's table says band 465, exact arithmetic would say 466."""
    mid = bucket_size // 2
    dist = torch.arange(max_position, dtype=torch.long)
    band = torch.ceil(torch.log(dist.clamp_min(1) / mid) / math.log((max_position - 1) / mid) * (mid - 1)).long()
    magnitude = torch.where(dist <= mid, dist, mid + band)                       # |bucket| by distance
    rel = dist[:, None] - dist[None, :]
    return torch.sign(rel) * magnitude[rel.abs()] + (bucket_size - 1)


@dataclass
class TextAdaptorConfig(BaseAdaptorConfig):
    token_bucket_size: int = field(default=256, metadata={"help": "token bucket size"})
    share_input_output_embed: bool = True
    output_embed_dim: Optional[int] = 512
    output_dim: Optional[int] = None
    output_bias: bool = False


@register_config("ofasys.adaptor", "text", TextAdaptorConfig)
class TextAdaptor(BaseAdaptor):
    pos_batch_invariant = True          # positions are arange- / grid-derived: identical for every batch row

    def __init__(self, embed_tokens: Embedding, dictionary: Dictionary, is_src: bool, general_adaptor,
                 cfg: TextAdaptorConfig):
        super().__init__(embed_tokens, dictionary, is_src, general_adaptor, cfg)
        self.embed_positions = Embedding(cfg.max_position + 2, cfg.embed_dim)
        token_num_rel_dis = 2 * cfg.token_bucket_size - 1
        token_rp_bucket = make_token_bucket_position(cfg.token_bucket_size, cfg.max_position)
        num_rel_pos_tables = 1 if self.cfg.share_attn_bias else self.num_layers
        self.token_rel_pos_table_list = nn.ModuleList(
            [Embedding(token_num_rel_dis, cfg.num_attention_heads, zero_init=True) for _ in range(num_rel_pos_tables)])
        self.register_buffer("token_rp_bucket", token_rp_bucket)
        self.share_input_output_embed = bool(cfg.share_input_output_embed)
        self.output_dim = cfg.output_dim if cfg.output_dim is not None else len(dictionary)
        self.output_embed_dim = cfg.output_embed_dim
        self.output_embed_bias = cfg.output_bias
        self.output_projection = None
        self.build_output_projection(dictionary)

    def build_output_projection(self, dictionary):
        if self.share_input_output_embed:
            self.output_projection = self.embed_tokens_T
        else:
            self.output_projection = OfaLinear(self.output_embed_dim, self.output_dim, bias=self.output_embed_bias)
            nn.init.normal_(self.output_projection.weight, mean=0, std=self.output_embed_dim ** -0.5)

    def get_rel_pos_bias(self, batch_size, seq_length, idx, **kwargs):
        """table_l[bucket[:T,:T]] -> [T,T,A]  (adaptor/text.py:101-104)."""
        if seq_length > self.token_rp_bucket.size(0):                  # the reference fails on the size mismatch (slicing clamps)
            raise ValueError(f"sequence length {seq_length} exceeds the {self.token_rp_bucket.size(0)} positions of token_rp_bucket")
        rp_bucket = ops.cached_index(self, ("text", seq_length), lambda: self.token_rp_bucket[:seq_length, :seq_length].contiguous())
        return ops.embedding(rp_bucket, self.token_rel_pos_table_list[idx].weight, plan_key=("text", ops.owner_token(self)))

    def forward(self, slot: Slot, **kwargs) -> AdaptorOutput:
        src_tokens = slot.value
        bsz, seq_len = src_tokens.shape
        pad = self.dictionary.pad()
        if pad is not None and src_tokens.is_cuda:
            token_embedding, padding_masks = self.embed_tokens_and_pad_mask(src_tokens, pad)      # one launch for both
        else:
            token_embedding = self.embed_tokens(src_tokens)
            padding_masks = src_tokens.eq(pad) if pad is not None else torch.zeros_like(src_tokens, dtype=torch.bool)
        # positions are arange(T) for every row regardless of padding (utils.py:623-630): looked up ONCE, [1, T, D], and handed on as
        # the [B, T, D] the contract names through a stride-0 expand (ops.shared_rows: the post-hook, the entangled add and the
        # position bias read the one copy)
        positions = ops.cached_index(self, ("arange", seq_len), lambda: torch.arange(seq_len, device=src_tokens.device).unsqueeze(0))
        pos_embed = self.embed_positions(positions).expand(bsz, -1, -1)
        return AdaptorOutput(token_embedding, padding_masks, pos_embed, [])

    def forward_output(self, x: Tensor, extra: Dict[str, Any], slot: Slot, **kwargs):
        """[B,T,D] -> logits [B,T,V] through the tied embedding (adaptor/text.py:129-142).  The logits storage has its
        row stride padded to a multiple of 8 elements so the criterion and the dgrad/wgrad GEMMs stay vectorised."""
        return self.output_projection(x), extra
