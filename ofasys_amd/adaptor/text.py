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
    """Integer bucket table, bit-exact with adaptor/text.py:20-30 (built once on the host at construction)."""
    context_pos = torch.arange(max_position, dtype=torch.long)[:, None]
    memory_pos = torch.arange(max_position, dtype=torch.long)[None, :]
    relative_pos = context_pos - memory_pos
    sign = torch.sign(relative_pos)
    mid = bucket_size // 2
    abs_pos = torch.where((relative_pos < mid) & (relative_pos > -mid), mid - 1, torch.abs(relative_pos))
    log_pos = torch.ceil(torch.log(abs_pos / mid) / math.log((max_position - 1) / mid) * (mid - 1)) + mid
    log_pos = log_pos.int()
    bucket_pos = torch.where(abs_pos.le(mid), relative_pos, log_pos * sign).long()
    return bucket_pos + bucket_size - 1


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
        if self.dictionary.pad() is not None:
            padding_masks = src_tokens.eq(self.dictionary.pad())
        else:
            padding_masks = torch.zeros_like(src_tokens, dtype=torch.bool)
        bsz, seq_len = src_tokens.shape
        positions = torch.arange(seq_len, device=src_tokens.device).unsqueeze(0).expand(bsz, seq_len)   # utils.py:623-630
        pos_embed = self.embed_positions(positions)
        token_embedding = self.embed_tokens(src_tokens)
        return AdaptorOutput(token_embedding, padding_masks, pos_embed, [])

    def forward_output(self, x: Tensor, extra: Dict[str, Any], slot: Slot, **kwargs):
        """[B,T,D] -> logits [B,T,V] through the tied embedding (adaptor/text.py:129-142).  The logits storage has its
        row stride padded to a multiple of 8 elements so the criterion and the dgrad/wgrad GEMMs stay vectorised."""
        return self.output_projection(x), extra
