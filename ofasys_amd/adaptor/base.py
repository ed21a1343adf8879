"""AdaptorOutput / BaseAdaptorConfig / BaseAdaptor -- the plugin base the `@register_config("ofasys.adaptor", ...)`
adaptors derive from (reference: adaptor/base.py:19-266).  Same constructor signature, hook semantics and state-dict
keys; the post-forward hook (scale, optional position entangling, type embedding, LayerNorms, dropout, rel-pos bias
list) runs on the HIP kernels."""
import math
from abc import abstractmethod
from dataclasses import dataclass, field
from typing import Any, Dict, List, Union

import torch
from torch import Tensor

from .. import ops
from ..configure import BaseDataclass
from ..module import Dropout, Embedding, LayerNorm
from ..preprocessor import Dictionary, Slot


@dataclass
class AdaptorOutput:
    """embed [B,T,H], masks bool [B,T], pos_embed [B,T,H], self_attn_bias List[[B,A,T,T]]  (adaptor/base.py:19-53)."""
    embed: torch.Tensor
    masks: torch.Tensor
    pos_embed: torch.Tensor
    self_attn_bias: List[torch.Tensor]
    modal_mask: torch.Tensor = None
    pos_shared: bool = False      # pos_embed (and the rel-pos bias) is the same for every batch row: see BaseAdaptor.pos_batch_invariant

    def __post_init__(self):
        assert self.embed is not None
        batch_size, seq_length, hidden_size = self.embed.shape
        if self.masks is not None:
            assert self.masks.shape == (batch_size, seq_length)
        if self.pos_embed is not None:
            assert self.pos_embed.shape == (batch_size, seq_length, hidden_size)

    @property
    def seq_length(self):
        return self.embed.shape[1]


@dataclass
class BaseAdaptorConfig(BaseDataclass):
    is_active: bool = field(default=False, metadata={"help": "is active for config_store"})
    layernorm_embedding: bool = True
    layernorm_position: bool = True
    add_type_embedding: bool = True
    entangle_position_embedding: bool = False
    no_scale_embedding: bool = True
    scale_embedding_gradient: float = 1.0
    dropout: float = None
    embed_dim: int = None
    num_attention_heads: int = None
    encoder_layers: int = None
    decoder_layers: int = None
    max_position: int = None
    use_self_attn_bias: bool = None
    share_attn_bias: bool = None

    # adaptor field -> where an unset (None) value is inherited from in the model config (adaptor/base.py:83-101).
    # entangle_position_embedding is in the table for parity with the reference, but its dataclass default is False, never None:
    # it is NOT inherited from the model -- an adaptor entangles positions only when its own config says so.
    _INHERITED = {
        "dropout": "dropout",
        "embed_dim": "encoder.embed_dim",
        "num_attention_heads": "encoder.attention_heads",
        "encoder_layers": "encoder.layers",
        "decoder_layers": "decoder.layers",
        "max_position": "max_source_positions",
        "use_self_attn_bias": "use_self_attn_bias",
        "share_attn_bias": "share_attn_bias",
        "entangle_position_embedding": "entangle_position_embedding",
    }

    def parse_from_model_cfg(self, model_cfg):
        """Fill every field the adaptor's own config left unset from the model config (table above)."""
        for name, path in self._INHERITED.items():
            if getattr(self, name) is None:
                node = model_cfg
                for part in path.split("."):
                    node = getattr(node, part)
                setattr(self, name, node)


class BaseAdaptor(torch.nn.Module):
    # True when the adaptor's pos_embed rows do not depend on the batch row (text: embed_positions(arange), adaptor/text.py:124;
    # image / video: the patch grid, image_resnet.py:153-157; audio: arange).  The position bias of such slots is ONE [A, T, T]
    # matrix per layer and the general adaptor builds it once instead of B times (ops.SharedBias).  A custom adaptor whose
    # positions vary per sample keeps the default False and gets the reference's dense [B, A, T, T] assembly.
    pos_batch_invariant = False

    def __init__(self, embed_tokens: Embedding, dictionary: Dictionary, is_src: bool, general_adaptor,
                 cfg: BaseAdaptorConfig):
        super().__init__()
        D = cfg.embed_dim
        # the shared token embedding is reached through closures, NOT registered as a child module: the state dict must list it once,
        # under the general adaptor (adaptor/base.py:128-136)
        self.embed_tokens = lambda ids: embed_tokens(ids)
        self.embed_tokens_T = lambda rows: ops.linear(rows, embed_tokens.weight)      # tied output projection
        # (embedding rows, ids == pad) from one launch (adaptor/text.py:108-125 computes the two separately)
        self.embed_tokens_and_pad_mask = lambda ids, pad: ops.embedding_with_pad_mask(ids, embed_tokens.weight, embed_tokens.padding_idx, pad)
        self.cfg, self.dictionary, self.is_src = cfg, dictionary, is_src
        self._general_adaptor = [general_adaptor]                                      # (a list: not a child module either)
        self.num_layers = cfg.encoder_layers if is_src else cfg.decoder_layers
        self.embed_scale = 1.0 if cfg.no_scale_embedding else math.sqrt(D)
        # optional post-hook pieces.  Attribute names are state-dict keys and the REGISTRATION ORDER is part of the contract too: it is
        # the key order of the state dict and the order the initialiser draws its random numbers in (tests/test_init_cpu.py compares both
        # with the reference build) -- dropout, the two LayerNorms, then the type embedding.
        self.dropout_module = Dropout(cfg.dropout, module_name=type(self).__name__)
        self.layernorm_embedding = LayerNorm(D) if cfg.layernorm_embedding else None
        self.layernorm_position = LayerNorm(D) if cfg.layernorm_position else None
        self.type_embedding = Embedding(1, D) if cfg.add_type_embedding else None
        self.register_forward_hook(BaseAdaptor.forward_hook_fn)

    @property
    def general_adaptor(self):
        return self._general_adaptor[0]

    def forward_hook_fn(self, inputs, output: AdaptorOutput):
        """adaptor/base.py:152-191."""
        slot: Slot = inputs[0]
        embed = ops.scale(output.embed, self.embed_scale)                    # :168 (sqrt(D) unless no_scale_embedding)
        pos = output.pos_embed if (self.cfg.entangle_position_embedding and output.pos_embed is not None) else None
        typ = self.type_embedding.weight.view(-1) if (slot.is_src and self.type_embedding is not None) else None
        if pos is not None or typ is not None:
            embed = ops.add_rowvec_mask(embed, pos, typ,                     # :170-173 in one pass
                                        vec_param=self.type_embedding.weight if typ is not None else None)
        if self.cfg.scale_embedding_gradient != 1.0:
            # :174-176 `embed * a + embed.detach() * (1 - a)`: the value is unchanged, the gradient flowing back into the
            # embedding / position / type tables is multiplied by a
            embed = ops.scale(embed, 1.0, float(self.cfg.scale_embedding_gradient))
        fused_dropout = False
        if self.layernorm_embedding is not None:
            p_drop = float(self.dropout_module.p)
            if embed.is_cuda and p_drop > 0 and (self.training or self.dropout_module.apply_during_inference):
                # dropout(layernorm_embedding(embed)) (:178, :182) in ONE pass: the residual-join kernel without a residual
                embed, _ = ops.residual_join(embed, None, self.layernorm_embedding, p_drop, True, None, eps=self.layernorm_embedding.eps)
                fused_dropout = True
            else:
                embed = self.layernorm_embedding(embed)
        if self.layernorm_position is not None and output.pos_embed is not None:
            base = ops.shared_rows(output.pos_embed)
            if base is not None:
                # positions shared by the batch (every built-in adaptor: a stride-0 expand of [1, T, D]): normalise the T rows once; the
                # [B, T, D] the contract promises stays a view, and the gradient arrives summed over the batch through the expand
                output.pos_embed = self.layernorm_position(base).expand_as(output.pos_embed)
            else:
                output.pos_embed = self.layernorm_position(output.pos_embed)
        output.embed = embed if fused_dropout else self.dropout_module(embed)
        output.pos_shared = bool(self.pos_batch_invariant)
        if not output.self_attn_bias and self.cfg.use_self_attn_bias:
            output.self_attn_bias = []
            batch_size, seq_length = output.embed.size()[:2]
            num_rel_pos_tables = 1 if self.cfg.share_attn_bias else self.num_layers
            for idx in range(num_rel_pos_tables):
                values = self.get_rel_pos_bias(batch_size, seq_length, idx)
                output.self_attn_bias.append(self.expand_rel_pos_bias(values, batch_size))
        return output

    @abstractmethod
    def forward(self, inputs: Union[Slot, List[Slot]], **kwargs) -> AdaptorOutput:
        raise NotImplementedError

    def forward_output(self, x: Tensor, extra: Dict[str, Any], slot: Slot, **kwargs):
        return x, extra

    @abstractmethod
    def get_rel_pos_bias(self, batch_size, seq_length, idx, **kwargs):
        raise NotImplementedError

    def expand_rel_pos_bias(self, values: Tensor, batch_size: int):
        """[T,T,A] -> [B,A,T,T] expand view (adaptor/base.py:242-256).  The view remembers the values it expands (`_ofa_values`): the
        general adaptor's bias assembly takes them back directly -- through autograd the round trip expand -> [0] would cost a
        zero-filled [B,A,T,T] gradient and a reduction over the batch per slot and layer."""
        out = values.unsqueeze(0).expand(batch_size, -1, -1, -1).permute([0, 3, 1, 2])
        out._ofa_values = values
        return out

    def upgrade_state_dict_named(self, state_dict, name):
        pass

    def update_sample(self, sample):
        return sample

    def check_adaptor_slot(self, slot):
        return self.general_adaptor.get_adaptor(slot) is self
