"""GeneralistModel + config + executor indirection (reference: model/ofa.py:41-650), same constructor / initialize /
forward signatures, arch presets and state-dict keys; the encoder/decoder run on the gfx950 kernels."""
import logging
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import List, Optional

import torch
from torch import Tensor
from torch.nn import Module, ModuleDict

from .. import ops
from ..adaptor.general import OFAAdaptorConfig
from ..configure import BaseDataclass, register_config
from ..module import init_bert_params
from ..preprocessor import Dictionary, Slot
from .transformer import TransformerDecoder, TransformerEncoder

logger = logging.getLogger(__name__)


@dataclass
class EncDecBaseConfig(BaseDataclass):       # module/transformer_config.py:24-62
    embed_path: Optional[str] = None
    embed_dim: int = 512
    ffn_embed_dim: int = 2048
    layers: int = 6
    attention_heads: int = 8
    normalize_before: bool = False
    learned_pos: bool = False
    layerdrop: float = 0
    layers_to_keep: Optional[str] = None


@dataclass
class DecoderConfig(EncDecBaseConfig):
    input_dim: int = 512
    output_dim: int = 512


@dataclass
class GeneralistModelConfig(BaseDataclass):
    """TransformerConfig (module/transformer_config.py:65-177) + GeneralistModelConfig (model/ofa.py:41-122) with the
    values of config/default_model.yaml:1-23 as defaults.  Flat fairseq-style aliases (`encoder_embed_dim`, ...) resolve
    to the nested configs like module/transformer_config.py:179-192."""
    activation_fn: str = "gelu"
    dropout: float = 0.1
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    relu_dropout: float = 0.0
    encoder: EncDecBaseConfig = field(default_factory=lambda: EncDecBaseConfig(normalize_before=True, learned_pos=True))
    decoder: DecoderConfig = field(default_factory=lambda: DecoderConfig(normalize_before=True, learned_pos=True))
    max_source_positions: int = 1024
    max_target_positions: int = 1024
    share_decoder_input_output_embed: bool = True
    share_all_embeddings: bool = True
    layernorm_embedding: bool = True
    no_scale_embedding: bool = True
    tie_adaptive_weights: bool = False
    checkpoint_activations: bool = False
    offload_activations: bool = False
    no_cross_attention: bool = False
    cross_self_attention: bool = False
    min_params_to_wrap: int = 100000000
    arch: str = "tiny"
    encode_drop_path_rate: float = 0.0
    decode_drop_path_rate: float = 0.0
    attn_scale_factor: float = 2
    freeze_encoder: bool = False
    freeze_encoder_embedding: bool = False
    freeze_decoder_embedding: bool = False
    add_type_embedding: bool = True
    entangle_position_embedding: bool = False
    sync_bn: bool = False
    scale_attn: bool = True
    scale_fc: bool = True
    scale_heads: bool = True
    scale_resids: bool = False
    checkpoint_adaptor_activations: bool = False
    use_fused: bool = False
    use_self_attn_bias: bool = True
    adaptor: OFAAdaptorConfig = field(default_factory=OFAAdaptorConfig)
    share_attn_bias: bool = False
    modal_ffn: bool = False

    def __getattr__(self, name):
        if name.startswith("encoder_") and name != "encoder_":
            return getattr(self.__dict__["encoder"], name[len("encoder_"):])
        if name.startswith("decoder_") and name != "decoder_":
            return getattr(self.__dict__["decoder"], name[len("decoder_"):])
        raise AttributeError(name)

    def __setattr__(self, name, value):
        """module/transformer_config.py:186-192: a flat `encoder_x` / `decoder_x` write lands in the nested config (so that
        `cfg.encoder_normalize_before = False` reaches the layers, which read `cfg.encoder.normalize_before`)."""
        for side in ("encoder", "decoder"):
            if name.startswith(side + "_") and name != side + "_" and side in self.__dict__:
                setattr(self.__dict__[side], name[len(side) + 1:], value)
                return
        super().__setattr__(name, value)


class OFAExecutor(ABC):      # model/ofa.py:125-154
    @abstractmethod
    def forward(self, ofa_model, slots, **kwargs):
        raise NotImplementedError

    @abstractmethod
    def get_normalized_probs(self, ofa_model, net_output, log_probs, sample=None):
        raise NotImplementedError

    @abstractmethod
    def forward_decoder(self, ofa_model, prev_output_tokens, **kwargs):
        raise NotImplementedError


class OFAEncoderDecoderExecutor(OFAExecutor):
    def __init__(self, encoder_name: str = "transformer_encoder", decoder_name: str = "transformer_decoder") -> None:
        super().__init__()
        self.encoder_name, self.decoder_name = encoder_name, decoder_name

    def forward(self, ofa_model, slots: List[Slot], features_only: bool = False, full_context_alignment: bool = False,
                alignment_layer: Optional[int] = None, alignment_heads: Optional[int] = None,
                return_all_hiddens: bool = False, return_encoder_out: bool = False, return_hf_dict: bool = False,
                return_all_attention_weights: bool = False, pack=None):
        """model/ofa.py:165-285.  pack (not in the reference): a packing.PackPlan -- the stack runs on the non-pad positions only
        and the logits come back packed, [1, plan.dec rows, V] (row r = padded position plan.dec_index[r])."""
        encoder = ofa_model.get_model_by_name(self.encoder_name)
        decoder = ofa_model.get_model_by_name(self.decoder_name)
        pk = {} if pack is None else {"pack": pack}
        encoder_out = encoder([s for s in slots if s.is_src], return_all_hiddens=return_all_hiddens,
                              return_all_attention_weights=return_all_attention_weights, **pk)
        decoder_out, decoder_extra_out = decoder(
            [s for s in slots if not s.is_src], encoder_out=encoder_out, features_only=features_only,
            full_context_alignment=full_context_alignment, alignment_layer=alignment_layer,
            alignment_heads=alignment_heads, return_all_hiddens=return_all_hiddens,
            return_all_attention_weights=return_all_attention_weights, **pk)
        if return_hf_dict:
            ret = {"last_hidden_state": decoder_extra_out["last_hidden_state"]}
            if return_all_attention_weights:
                ret["decoder_attentions"] = decoder_extra_out["decoder_attentions"]
                ret["cross_attentions"] = decoder_extra_out["cross_attentions"]
            if return_all_hiddens:
                ret["decoder_hidden_states"] = decoder_extra_out["inner_states"]
            if not features_only:
                ret["decoder_adaptor_out"] = decoder_out
            if return_encoder_out:
                ret["encoder_last_hidden_state"] = encoder_out["encoder_out"]
                if return_all_attention_weights:
                    ret["encoder_attentions"] = encoder_out["encoder_attention_weights"]
                if return_all_hiddens:
                    ret["encoder_hidden_states"] = encoder_out["encoder_states"]
            return ret
        if return_encoder_out:
            return decoder_out, decoder_extra_out, encoder_out
        return decoder_out, decoder_extra_out

    def get_logits_from_net_output(self, net_output):
        return net_output["decoder_adaptor_out"] if isinstance(net_output, dict) else net_output[0]

    def get_normalized_probs(self, ofa_model, net_output, log_probs: bool, sample=None):
        """fp32 (log-)softmax of the logits (model/ofa.py:287-299, module/utils.py:451-462) via the row-softmax kernels."""
        logits = self.get_logits_from_net_output(net_output)
        return ops.log_softmax_fp32(logits) if log_probs else ops.softmax_fp32(logits)

    def forward_decoder(self, ofa_model, prev_output_tokens, **kwargs):
        return ofa_model.get_model_by_name(self.decoder_name)(prev_output_tokens, **kwargs)


class OFAExecutorContext(object):
    def __init__(self, ofa_model, ofa_executor) -> None:
        self.ofa_model, self.ofa_executor = ofa_model, ofa_executor
        self.previous_ofa_executor = ofa_model.get_active_executor()

    def __enter__(self):
        self.ofa_model.set_active_executor(self.ofa_executor)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        self.ofa_model.set_active_executor(self.previous_ofa_executor)


@register_config("ofasys.model", "unify", dataclass=GeneralistModelConfig)
class GeneralistModel(Module):
    def __init__(self, cfg: GeneralistModelConfig = None):
        super().__init__()
        if cfg is None:
            cfg = GeneralistModelConfig()        # == config/default_model.yaml
        self.cfg = cfg
        if cfg.offload_activations:
            cfg.checkpoint_activations = True
        if cfg.arch:
            globals()["ofa_arch_" + cfg.arch](cfg)

    def initialize(self, global_dict: Dictionary):
        """model/ofa.py:360-385."""
        self.encoder = TransformerEncoder(self.cfg, global_dict)
        self.decoder = TransformerDecoder(self.cfg, global_dict, self.cfg.no_cross_attention)
        self.extra_models = ModuleDict()
        self.active_executor: OFAExecutor = OFAEncoderDecoderExecutor()
        self.apply(init_bert_params)
        if self.cfg.freeze_encoder:
            self.encoder.requires_grad_(False)
        self.global_dict = global_dict

    @property
    def supported_targets(self):
        return {"self"}

    def executor_context(self, executor) -> OFAExecutorContext:
        return OFAExecutorContext(self, ofa_executor=executor)

    def get_active_executor(self):
        return self.active_executor

    def set_active_executor(self, executor: OFAExecutor) -> None:
        assert isinstance(executor, OFAExecutor)
        self.active_executor = executor

    def get_model_by_name(self, model_name: str) -> Module:
        if model_name == "transformer_encoder":
            return self.encoder
        if model_name == "transformer_decoder":
            return self.decoder
        assert model_name in self.extra_models, "Warning!! " + model_name
        return self.extra_models[model_name]

    def forward(self, slots: List[Slot], features_only: bool = False, full_context_alignment: bool = False,
                alignment_layer: Optional[int] = None, alignment_heads: Optional[int] = None,
                return_all_hiddens: bool = False, return_encoder_out: bool = False, return_hf_dict: bool = False,
                return_all_attention_weights: bool = False, pack=None):
        """The reference's keyword list (model/ofa.py:410-421) + `pack` (ofasys_amd/packing.py: ragged row packing)."""
        kw = {} if pack is None else {"pack": pack}
        return self.active_executor.forward(
            self, slots=slots, features_only=features_only, full_context_alignment=full_context_alignment,
            alignment_layer=alignment_layer, alignment_heads=alignment_heads, return_all_hiddens=return_all_hiddens,
            return_encoder_out=return_encoder_out, return_hf_dict=return_hf_dict,
            return_all_attention_weights=return_all_attention_weights, **kw)

    def get_normalized_probs(self, net_output, log_probs: bool, sample=None):
        return self.active_executor.get_normalized_probs(self, net_output=net_output, log_probs=log_probs, sample=sample)

    def get_targets(self, sample, net_output):
        return sample["target"]                    # model/fairseq_model.py:61-63

    def update_sample(self, sample):
        sample = self.encoder.adaptor.update_sample(sample)
        sample = self.decoder.adaptor.update_sample(sample)
        return sample

    def forward_decoder(self, prev_output_tokens, **kwargs):
        return self.active_executor.forward_decoder(self, prev_output_tokens, **kwargs)

    def max_positions(self):
        return (self.encoder.max_positions(), self.decoder.max_positions())

    def max_decoder_positions(self):
        return self.decoder.max_positions()

    def upgrade_state_dict_named(self, state_dict, name):
        """Drop outdated keys and complete missing ones from the model (model/ofa.py:443-475)."""
        del_keys = ["decoder.output_projection.weight"]
        if not self.cfg.use_self_attn_bias:
            del_keys += [f"{p}.{q}" for p in ("decoder.cross_pos_q_linear", "decoder.cross_pos_k_linear",
                                             "encoder.adaptor.pos_q_linear", "encoder.adaptor.pos_k_linear",
                                             "decoder.adaptor.pos_q_linear", "decoder.adaptor.pos_k_linear")
                         for q in ("weight", "bias")]
        for k in del_keys:
            state_dict.pop(k, None)
        prefix = name + "." if name != "" else ""
        own = self.state_dict()
        for param_name in own:
            if (prefix + param_name) not in state_dict:
                state_dict[prefix + param_name] = own[param_name]


def _arch(cfg, dim, layers_e, layers_d, heads, ffn=None, resnet="resnet101"):
    cfg.encoder.embed_dim = cfg.decoder.embed_dim = dim
    cfg.encoder.ffn_embed_dim = cfg.decoder.ffn_embed_dim = ffn if ffn is not None else 4 * dim
    cfg.decoder.input_dim = cfg.decoder.output_dim = dim
    cfg.encoder.layers, cfg.decoder.layers = layers_e, layers_d
    cfg.encoder.attention_heads = cfg.decoder.attention_heads = heads
    if hasattr(cfg.adaptor, "image_resnet"):
        cfg.adaptor.image_resnet.resnet_type = resnet


# model/ofa.py:557-650
def ofa_arch_base(cfg): _arch(cfg, 768, 6, 6, 12)
def ofa_arch_asr_small(cfg): _arch(cfg, 256, 12, 6, 4, ffn=2048)
def ofa_arch_asr_base(cfg): _arch(cfg, 768, 12, 6, 12)
def ofa_arch_tiny(cfg): _arch(cfg, 256, 4, 4, 4, resnet="resnet50")
def ofa_arch_medium(cfg): _arch(cfg, 512, 4, 4, 8)
def ofa_arch_large(cfg): _arch(cfg, 1024, 12, 12, 16, resnet="resnet152")
def ofa_arch_huge(cfg): _arch(cfg, 1280, 24, 12, 16, resnet="resnet152")
def ofa_arch_6b(cfg): _arch(cfg, 2560, 36, 24, 32, resnet=None)
def ofa_arch_8b(cfg): _arch(cfg, 2560, 48, 36, 32, resnet=None)
def ofa_arch_10b(cfg): _arch(cfg, 2816, 48, 36, 32, resnet=None)
