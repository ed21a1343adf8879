"""OFAGeneralAdaptor (reference: adaptor/general.py:36-316): dispatches each Slot to its adaptor (in ModalityType
order, outputs kept in slot order), concatenates along T and assembles the per-layer attention bias
(absolute-position bias from pos_q/pos_k projections + each slot's relative-position bias on its diagonal block)."""
from dataclasses import fields
from typing import Any, Dict, List

import torch
from torch import Tensor

from .. import ops
from ..configure import ConfigStore
from ..module import Embedding, OfaLinear
from ..preprocessor import ModalityType, Slot
from .base import AdaptorOutput, BaseAdaptor

OFAAdaptorConfig = ConfigStore().make_dataclass("ofasys.adaptor", "OFAAdaptorConfig", __name__,
                                                ["text", "image_resnet", "image_patch_embed", "image_vqgan", "box"])   # general.py:30-35

default_adaptor = {   # adaptor/general.py:36-46
    ModalityType.TEXT: "text",
    ModalityType.IMAGE: "image_resnet",
    ModalityType.BOX: "text",
    ModalityType.AUDIO: "audio_fbank",
    ModalityType.PHONE: "text",
    ModalityType.VIDEO: "video_image_sequence",
    ModalityType.MOTION: "text",
    ModalityType.STRUCT: "text",
    ModalityType.CATEGORY: "text",
}


class OFAGeneralAdaptor(torch.nn.Module):
    _embed_tokens = None

    def __init__(self, cfg, dictionary, is_src):
        super().__init__()
        self.embed_tokens = self.build_embedding(cfg, dictionary)
        self.cfg = cfg
        self.is_src = is_src
        self.name2adaptor: Dict[str, BaseAdaptor] = {}
        # the dataclass' own fields (reference order: state-dict order and the initial RNG stream), then adaptors a user registered
        # after ofasys_amd was imported (their configs appear on first access: configure.ConfigStore.make_dataclass)
        for name in [f.name for f in fields(cfg.adaptor)] + ConfigStore().late_plugins(cfg.adaptor):
            if name.startswith("_"):
                continue
            if name == "image_vqgan" and is_src:                        # general.py:73-80
                continue
            if name in ("image_resnet", "video_image_sequence", "image_vit") and not is_src:
                continue
            config = getattr(cfg.adaptor, name)
            config.parse_from_model_cfg(cfg)
            if config.is_active is False:
                continue
            self.name2adaptor[name] = ConfigStore().get("ofasys.adaptor", name).target(
                self.embed_tokens, dictionary, is_src, self, config)
            setattr(self, name, self.name2adaptor[name])
        embed_dim = cfg.encoder_embed_dim if is_src else cfg.decoder_embed_dim
        self.num_attention_heads = cfg.encoder_attention_heads if is_src else cfg.decoder_attention_heads
        self.pos_scaling = float(embed_dim / cfg.encoder_attention_heads * cfg.attn_scale_factor) ** -0.5   # :98
        if not self.cfg.entangle_position_embedding:
            self.pos_q_linear = OfaLinear(embed_dim, embed_dim)
            self.pos_k_linear = OfaLinear(embed_dim, embed_dim)

    # `embed_tokens` is a shared module registered once per general adaptor (encoder and decoder hold the same object)
    def get_adaptor(self, slot: Slot) -> BaseAdaptor:
        if slot.get_attr("adaptor"):
            return self.name2adaptor[slot.get_attr("adaptor")]
        return self.name2adaptor[default_adaptor[slot.modality]]

    def forward(self, slots: List[Slot], **kwargs):
        modality_outputs = [None for _ in range(len(slots))]
        cnt = 0
        for mod in ModalityType:                                        # general.py:137-149 (RNG/dropout order)
            for i, slot in enumerate(slots):
                if slot.modality == mod:
                    modality_outputs[i] = self.get_adaptor(slot)(slot, **kwargs)
                    cnt += 1
            if cnt == len(slots):
                break
        assert cnt == len(slots), cnt
        output = self.concat(modality_outputs)
        modal_mask = None
        if getattr(self.cfg, "modal_ffn", False):                       # general.py:143-146, 156-157: [B, T] int64, value = modality - 1
            parts = [torch.full_like(o.masks, int(slot.modality.value) - 1, dtype=torch.int64)
                     for o, slot in zip(modality_outputs, slots)]
            modal_mask = torch.cat(parts, dim=-1)
            # every column belongs to one slot: the layers route on this host-side layout (module/transformer_layer.py, _modal_plan)
            modal_mask._ofa_cols = tuple((int(slot.modality.value) - 1, int(o.masks.shape[-1])) for o, slot in zip(modality_outputs, slots))
        return output.embed, output.masks, output.pos_embed, output.self_attn_bias, modal_mask

    def forward_output(self, x: Tensor, extra: Dict[str, Any], slots: List[Slot], **kwargs):
        output_slot = None
        for slot in slots:
            if not slot.is_src:
                assert output_slot is None, "supports only one target slot"
                output_slot = slot
        assert output_slot
        return self.get_adaptor(output_slot).forward_output(x, extra, slot=output_slot)

    def build_embedding(self, cfg, dictionary):
        if OFAGeneralAdaptor._embed_tokens is not None:
            return OFAGeneralAdaptor._embed_tokens
        assert cfg.share_all_embeddings
        assert cfg.encoder_embed_dim == cfg.decoder_embed_dim
        embed_tokens = Embedding(num_embeddings=len(dictionary), embedding_dim=cfg.encoder_embed_dim,
                                 padding_idx=dictionary.pad())
        cfg.share_decoder_input_output_embed = True
        if cfg.freeze_encoder_embedding:
            embed_tokens.weight.requires_grad = False
        OFAGeneralAdaptor._embed_tokens = embed_tokens
        return embed_tokens

    def build_abs_pos_bias(self, pos_embed):
        """pos_q*pos_scaling @ pos_k^T per head -> [B,A,T,T] (general.py:223-243)."""
        batch_size, seq_length = pos_embed.size(0), pos_embed.size(1)
        if not self.cfg.entangle_position_embedding:
            pos_q = self.pos_q_linear(pos_embed, alpha=self.pos_scaling)
            pos_k = self.pos_k_linear(pos_embed)
            return ops.heads_matmul_nt(pos_q, pos_k, self.num_attention_heads)
        return torch.zeros(batch_size, self.num_attention_heads, seq_length, seq_length, dtype=pos_embed.dtype,
                           device=pos_embed.device)

    def concat(self, modality_outputs: List[AdaptorOutput]) -> AdaptorOutput:
        """general.py:245-282."""
        if len(modality_outputs) == 1:
            o = modality_outputs[0]
            output = AdaptorOutput(o.embed, o.masks, o.pos_embed, None)
        else:
            pos_parts = [x.pos_embed for x in modality_outputs]
            bases = [ops.shared_rows(t) for t in pos_parts]
            if all(b is not None for b in bases):
                # every slot's positions are shared by the batch: concatenate the [1, n, D] copies, expand the result (stride 0)
                pos_embed = torch.cat(bases, dim=1).expand(pos_parts[0].shape[0], -1, -1)
            else:
                pos_embed = torch.cat(tuple(pos_parts), dim=1)
            # `lazy_embed_concat` (set by the stacks around their call when they will pack the rows): the slots' embeddings are handed on
            # unconcatenated (ops.LazyCat) and the packed rows are gathered from them directly
            embeds = tuple(x.embed for x in modality_outputs)
            embed = ops.LazyCat(embeds) if getattr(self, "lazy_embed_concat", False) else torch.cat(embeds, dim=1)
            output = AdaptorOutput(embed, torch.cat(tuple(x.masks for x in modality_outputs), dim=1), pos_embed, None)
        self.last_pos_shared = all(getattr(mo, "pos_shared", False) for mo in modality_outputs)    # (read by the stacks)
        if not self.cfg.use_self_attn_bias:
            return output
        output.self_attn_bias = []
        # The position bias is the same for every sample when every slot's positions are (all built-in adaptors): it is then built
        # ONCE from row 0 -- [1, A, T, T] instead of general.py:223-282's [B, A, T, T], 154 MB per layer at cfg-2b -- and handed to
        # the attention kernels as an ops.SharedBias, which also sum its gradient over the batch in-kernel
        num_layers = self.cfg.encoder.layers if self.is_src else self.cfg.decoder.layers
        num_rel_pos_tables = 1 if self.cfg.share_attn_bias else num_layers
        # slot biases arrive as the reference's [B,A,T,T] expand view of [T,T,A] values (take the values back), or -- an adaptor whose
        # forward() fills `self_attn_bias` itself, which the post-hook leaves alone (adaptor/base.py:183-189) -- as a genuinely
        # per-sample [B,A,n,n] tensor ("batch")
        layer_values = [[_unexpand(mo.self_attn_bias[idx] if mo.self_attn_bias else None) for mo in modality_outputs]
                        for idx in range(num_rel_pos_tables)]
        per_sample = any(isinstance(v, _PerSample) for values in layer_values for v in values)
        shared = self.last_pos_shared and not per_sample
        output.pos_shared = shared
        abs_pos_bias = self.build_abs_pos_bias(ops.first_sample(output.pos_embed) if shared else output.pos_embed)
        starts, s = [], 0
        for mo in modality_outputs:
            starts.append(s)
            s += mo.seq_length
        assert s == output.seq_length
        abs_fan = ops.fan_out(abs_pos_bias, num_rel_pos_tables)      # one view per layer: their gradients are summed in one launch
        for idx, values in enumerate(layer_values):
            kinds, tensors = [], []
            for v in values:
                if v is None:
                    kinds.append(None)
                elif isinstance(v, ops.OuterRelPos):
                    kinds.append("outer")
                    tensors += [v.frames, v.patches]
                elif isinstance(v, _PerSample):
                    kinds.append("batch")
                    tensors.append(v.t)
                else:
                    kinds.append("dense")
                    tensors.append(v)
            b, swz_row, swz_col = ops.BiasAssembleFn.apply(abs_fan[idx], starts, kinds, *tensors)
            # (squeeze, not b[0]: a select's backward zero-fills a whole [1, A, T, T] tensor and copies the gradient into it, per layer)
            output.self_attn_bias.append(ops.SharedBias(b.squeeze(0), (swz_row, swz_col) if swz_row is not None else None) if shared else b)
        return output

    def upgrade_state_dict_named(self, state_dict, name):
        for adaptor_name in self.name2adaptor:
            self.name2adaptor[adaptor_name].upgrade_state_dict_named(state_dict, "{}.{}".format(name, adaptor_name))
        return state_dict

    def update_sample(self, sample):
        for adaptor_name in self.name2adaptor:
            self.name2adaptor[adaptor_name].update_sample(sample)
        return sample


class _PerSample:
    """A slot's own [B, A, n, n] attention bias (one matrix per sample): added block-wise to the dense [B, A, T, T] assembly."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def _unexpand(b):
    """[B,A,T,T] batch-expanded view (stride 0 on batch) of [T,T,A] values -> the [T,T,A] values; a genuinely per-sample bias
    (a custom adaptor's own `self_attn_bias`, adaptor/base.py:183-189) -> _PerSample: the whole layer bias is then assembled
    densely, as the reference does (adaptor/general.py:265-280), and runs on the dense-bias attention kernels."""
    if b is None:
        return None
    if isinstance(b, ops.LazyRelPosBias):                             # (video: frame-level + patch-level tables, never materialised)
        return b.values
    v = getattr(b, "_ofa_values", None)                              # BaseAdaptor.expand_rel_pos_bias: the values themselves
    if v is not None:
        return v
    if b.dim() == 4 and (b.stride(0) == 0 or b.size(0) == 1):      # (a batch of one: nothing to expand, any stride)
        return b[0].permute(1, 2, 0)
    if b.dim() == 3:
        return b
    if b.dim() == 4:
        return _PerSample(b)
    raise ValueError(f"self_attn_bias of an adaptor must be [B, A, n, n] (or [n, n, A] values), got shape {tuple(b.shape)}")
