"""TransformerEncoder / TransformerDecoder stacks (reference: model/transformer.py:33-156, 206-539) over the gfx950
layers.  Activations live batch-major ([B,T,C] storage); the [T,B,C] tensors handed to the layers and returned in
`encoder_out` are transposed views, so the reference's tensor contract holds without layout copies.

Deliberate differences (same results): no `masks.any()` host sync (transformer.py:110) -- padded rows are zeroed and
the padding mask is passed unconditionally, which is arithmetically the identity when nothing is padded."""
import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import kernels as K
from .. import ops
from ..adaptor import AdaptorOutput, OFAGeneralAdaptor
from ..module import LayerNorm, Linear, OfaLinear, TransformerDecoderLayer, TransformerEncoderLayer
from ..module.transformer_layer import LayerChain
from ..preprocessor import Dictionary, Slot


def _check_packable(cfg, wants_unsupported):
    """Packed rows produce no per-layer extras (no [B,A,Tt,Ts] attention map exists in this mode)."""
    if wants_unsupported:
        raise NotImplementedError("row packing: hidden-state / attention-weight outputs, incremental decoding and full-context "
                                  "alignment are only available on padded batches")


def _packed_bias_ok(bias_list, prefix_ok, what):
    """A position bias over packed rows: it has to be the batch-shared form (indexed by the position inside the sample), and packed
    row r of a sample has to BE padded position r -- the sample's valid positions a prefix of its padded row (packing.PackPlan)."""
    if not bias_list:
        return
    if not isinstance(bias_list[0], ops.SharedBias):
        raise NotImplementedError(f"row packing: the {what} position bias is a per-sample [B,A,T,T] tensor (an adaptor whose positions "
                                  "depend on the batch row produced it); run such batches padded")
    if not prefix_ok:
        raise NotImplementedError(f"row packing with a position bias needs every sample's valid {what} positions to be a prefix of its "
                                  "padded row (one ragged slot, at the end); run this batch padded")


_CKPT_NOTED = False


def _note_activation_checkpointing(cfg):
    """checkpoint_activations / offload_activations / checkpoint_adaptor_activations (model/transformer.py:50-51, 68-72, 230-231,
    270-274; model/ofa.py:349-350): the reference wraps the adaptor / every layer so that its forward is re-run during backward
    with the RNG state restored -- the SAME numbers for fewer live activations.  The flags are accepted and the activations are
    simply kept: a cfg-2 step holds a few GB of them out of 288 GB of HBM per device, and re-running the forward would cost a
    third more MFMA time per step for memory this part does not lack.  Results are those of the reference either way."""
    global _CKPT_NOTED
    if (cfg.checkpoint_adaptor_activations or cfg.checkpoint_activations or cfg.offload_activations) and not _CKPT_NOTED:
        _CKPT_NOTED = True
        warnings.warn("ofasys_amd: checkpoint_activations / offload_activations are accepted but activations are kept resident "
                      "(same results; 288 GB of HBM per device make the recompute a pure loss)")


def kept_layers(layers, p, training):
    """LayerDropModuleList.__iter__ (module/layer_drop.py:37-41): in training every layer survives an iteration with probability
    1 - p -- one uniform draw per layer from torch's default CPU generator, as the reference does; evaluation keeps them all.  The
    callers enumerate the SURVIVORS (so a survivor's `idx` -- which selects its per-layer position bias -- counts survivors, as in
    the reference's `for idx, layer in enumerate(self.layers)`).  The kept set is a host decision per step: it cannot be part of a
    captured step graph."""
    if not training or p <= 0.0:
        return list(layers)
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise NotImplementedError("LayerDrop draws the kept layers on the host every step: run the step eagerly (TrainStep(use_graph=False))")
    probs = torch.empty(len(layers)).uniform_()
    return [m for m, u in zip(layers, probs.tolist()) if u > p]


class TransformerEncoder(nn.Module):
    def __init__(self, cfg, dictionary: Dictionary):
        super().__init__()
        self.cfg = cfg
        self.dictionary = dictionary
        self.register_buffer("version", torch.Tensor([3]))
        OFAGeneralAdaptor._embed_tokens = None          # a fresh shared embedding per model (transformer.py:48)
        self.adaptor = OFAGeneralAdaptor(cfg, dictionary, True)
        _note_activation_checkpointing(cfg)
        self.layerdrop = float(cfg.encoder_layerdrop)                 # (model/transformer.py:53-54)
        self.layers = nn.ModuleList([])
        dpr = torch.linspace(0, cfg.encode_drop_path_rate, cfg.encoder_layers)
        self.layers.extend([self.build_encoder_layer(cfg, drop_path_rate=float(dpr[i])) for i in range(cfg.encoder_layers)])
        self.layer_norm = LayerNorm(cfg.encoder_embed_dim) if cfg.encoder_normalize_before else None

    def build_encoder_layer(self, cfg, drop_path_rate=0.0):
        return TransformerEncoderLayer(cfg, drop_path_rate=drop_path_rate)

    def forward(self, slots: List[Slot], return_all_hiddens: bool = False, return_all_attention_weights: bool = False, pack=None):
        """See model/transformer.py:78-102 for the returned dict.  pack: packing.PackPlan -- only the non-pad positions go through
        the layers ("encoder_out" is then [rows, 1, C] packed rows and "encoder_padding_mask" the plan's segment tables)."""
        if len(slots) == 0:
            return None
        self.adaptor.lazy_embed_concat = pack is not None       # packed rows are gathered from the slots' outputs: no concatenated [B,T,C]
        try:
            adaptor_output = AdaptorOutput(*self.adaptor(slots))
        finally:
            self.adaptor.lazy_embed_concat = False
        if pack is not None:
            _check_packable(self.cfg, return_all_hiddens or return_all_attention_weights)
            if self.cfg.use_self_attn_bias:
                _packed_bias_ok(adaptor_output.self_attn_bias, pack.enc_prefix, "encoder")
            x = ops.pack_rows(adaptor_output.embed, pack.enc_index, pack.enc_inverse).transpose(0, 1)   # [rows, 1, C] view
            layer_mask = pack.enc_self                               # padded rows are simply absent: nothing to zero or mask
        else:
            # zero the padded positions (transformer.py:110-112); unconditional, no host sync
            adaptor_output.embed = ops.add_rowvec_mask(adaptor_output.embed, None, None, adaptor_output.masks)
            x = adaptor_output.embed.transpose(0, 1)                 # B x T x C -> T x B x C (view)
            layer_mask = adaptor_output.masks
        T = x.size(0)
        encoder_states = [x] if return_all_hiddens else []
        encoder_attention_states = []
        chain = LayerChain()
        layers = kept_layers(self.layers, self.layerdrop, self.training)
        for idx, layer in enumerate(layers):
            if self.cfg.use_self_attn_bias:
                b = adaptor_output.self_attn_bias[0 if self.cfg.share_attn_bias else idx]
                self_attn_bias = b if isinstance(b, ops.SharedBias) else b.view(-1, T, T)   # (SharedBias: [A,T,T], one for the batch)
            else:
                self_attn_bias = None
            chain.next_ln = layers[idx + 1].self_attn_layer_norm if idx + 1 < len(layers) else self.layer_norm
            x, self_attn_weights = layer(x, encoder_padding_mask=layer_mask, self_attn_bias=self_attn_bias,
                                         need_attn=return_all_attention_weights, modal_mask=adaptor_output.modal_mask,
                                         chain=chain)
            if return_all_hiddens:
                encoder_states.append(x)
            if return_all_attention_weights:
                encoder_attention_states.append(self_attn_weights)
        normed = chain.take()
        if normed is not None:
            x = normed                                              # the last layer's join already applied layer_norm
        elif self.layer_norm is not None:
            x = self.layer_norm(x)
        return {
            "encoder_out": [x],                                     # T x B x C
            "encoder_padding_mask": [adaptor_output.masks if pack is None else pack],   # B x T (packed: the plan)
            "encoder_embedding": [adaptor_output.embed],            # B x T x C (packed mode, several slots: an ops.LazyCat of the slots' outputs)
            "encoder_states": encoder_states,
            "position_embeddings": [adaptor_output.pos_embed],      # B x T x C
            "position_embeddings_shared": [bool(getattr(self.adaptor, "last_pos_shared", False))],   # (every row identical: see SharedBias)
            "encoder_attention_weights": encoder_attention_states,
        }

    def reorder_encoder_out(self, encoder_out: Dict[str, List[Tensor]], new_order):
        def sel(key, dim):
            return [] if len(encoder_out[key]) == 0 else [ops.materialize(encoder_out[key][0]).index_select(dim, new_order)]
        return {
            "encoder_out": sel("encoder_out", 1), "encoder_padding_mask": sel("encoder_padding_mask", 0),
            "encoder_embedding": sel("encoder_embedding", 0),
            "encoder_states": [s.index_select(1, new_order) for s in encoder_out["encoder_states"]],
            "position_embeddings": sel("position_embeddings", 0),
            "position_embeddings_shared": list(encoder_out.get("position_embeddings_shared", [])),
        }

    def max_positions(self):
        return self.cfg.max_source_positions


class TransformerDecoder(nn.Module):
    def __init__(self, cfg, dictionary, no_encoder_attn=False):
        super().__init__()
        self.cfg = cfg
        self.dictionary = dictionary
        self.register_buffer("version", torch.Tensor([3]))
        self._future_mask = torch.empty(0)
        self.adaptor = OFAGeneralAdaptor(cfg, dictionary, False)
        _note_activation_checkpointing(cfg)
        self.layerdrop = float(cfg.decoder_layerdrop)                 # (model/transformer.py:244-245)
        self.share_input_output_embed = cfg.share_decoder_input_output_embed
        self.num_attention_heads = cfg.decoder_attention_heads
        embed_dim = cfg.decoder_embed_dim
        self.embed_dim = embed_dim
        self.output_embed_dim = int(cfg.decoder_output_dim)
        if self.cfg.use_self_attn_bias:
            self.cross_pos_q_linear = OfaLinear(embed_dim, embed_dim)
            self.cross_pos_k_linear = OfaLinear(embed_dim, embed_dim)
        self.cross_self_attention = cfg.cross_self_attention
        self.layers = nn.ModuleList([])
        # (sic) the reference builds the decoder's drop-path rates from the ENCODER's rate and layer count, :249-252
        dpr = torch.linspace(0, cfg.encode_drop_path_rate, cfg.encoder_layers)
        self.layers.extend([self.build_decoder_layer(cfg, no_encoder_attn, drop_path_rate=float(dpr[i]))
                            for i in range(cfg.decoder_layers)])
        self.num_layers = len(self.layers)
        self.layer_norm = LayerNorm(embed_dim) if cfg.decoder_normalize_before else None
        self.project_out_dim = (Linear(embed_dim, self.output_embed_dim, bias=False)
                                if embed_dim != self.output_embed_dim and not cfg.tie_adaptive_weights else None)
        self.adaptive_softmax = None

    def build_decoder_layer(self, cfg, no_encoder_attn=False, drop_path_rate=0.0):
        return TransformerDecoderLayer(cfg, no_encoder_attn, drop_path_rate=drop_path_rate)

    def get_cross_pos_info(self, embed, tgt_pos_embed, src_pos_embed, shared=False):
        """abs position bias for cross attention -> [B,A,Tt,Ts] (model/transformer.py:280-299); shared (both position embeddings are
        the same for every batch row): built once from row 0 -> ops.SharedBias [A,Tt,Ts]."""
        if shared:
            tgt_pos_embed, src_pos_embed = ops.first_sample(tgt_pos_embed), ops.first_sample(src_pos_embed)
        pos_q = self.cross_pos_q_linear(tgt_pos_embed, alpha=self.adaptor.pos_scaling)
        pos_k = self.cross_pos_k_linear(src_pos_embed)
        b = ops.heads_matmul_nt(pos_q, pos_k, self.num_attention_heads)
        return ops.SharedBias.of(b.squeeze(0)) if shared else b           # (squeeze: a view both ways; b[0]'s backward zero-fills + copies)

    def _cross_kv(self, enc, incremental_state):
        """ops.CrossKVShared for this forward -- the k|v projections of the encoder output for ALL layers as one GEMM -- when the
        layers' projection weights sit adjacent in a trainer's arena (trainer.FlatParams) and the fused 16-bit path runs; else None
        (every layer projects for itself)."""
        if enc is None or incremental_state is not None or enc.dtype not in (torch.bfloat16, torch.float16):
            return None
        packs = [getattr(getattr(layer, "encoder_attn", None), "_cross_all", None) for layer in self.layers]
        if any(p is None for p in packs) or any(p[0] is not packs[0][0] for p in packs) or packs[0][0]["layers"] != len(self.layers):
            return None
        rows = ops.batch_major(enc)                                   # [B, S, D] storage (a view: the stacks keep batch-major rows)
        return ops.CrossKVShared(rows.reshape(-1, rows.shape[-1]), packs[0][0])

    def _pos_shared(self, adaptor_output, encoder_out):
        """Are the target AND the source position embeddings identical for every batch row (all built-in adaptors)?"""
        tgt = bool(getattr(self.adaptor, "last_pos_shared", False))
        src = bool(encoder_out is not None and encoder_out.get("position_embeddings_shared") and encoder_out["position_embeddings_shared"][0])
        return tgt and src

    def forward(self, slots: List[Slot], encoder_out: Optional[Dict[str, List[Tensor]]] = None,
                incremental_state=None, features_only: bool = False, full_context_alignment: bool = False,
                alignment_layer: Optional[int] = None, alignment_heads: Optional[int] = None,
                return_all_hiddens: bool = False, return_all_attention_weights: bool = False, pack=None):
        x, extra = self.extract_features(slots, encoder_out=encoder_out, incremental_state=incremental_state,
                                         full_context_alignment=full_context_alignment, alignment_layer=alignment_layer,
                                         alignment_heads=alignment_heads, return_all_hiddens=return_all_hiddens,
                                         return_all_attention_weights=return_all_attention_weights, pack=pack)
        extra["last_hidden_state"] = x
        if not features_only:
            return self.adaptor.forward_output(x, extra, slots)
        return x, extra

    def extract_features(self, slots: List[Slot], encoder_out, incremental_state=None, full_context_alignment: bool = False,
                         alignment_layer: Optional[int] = None, alignment_heads: Optional[int] = None,
                         return_all_hiddens: bool = False, return_all_attention_weights: bool = False, pack=None):
        adaptor_output = AdaptorOutput(*self.adaptor(slots))
        if pack is not None:
            return self._extract_features_packed(adaptor_output, encoder_out, pack, incremental_state, full_context_alignment,
                                                 return_all_hiddens or return_all_attention_weights)
        bsz, slen = adaptor_output.embed.size()[:2]
        if alignment_layer is None:
            alignment_layer = self.num_layers - 1
        enc = padding_mask = src_pos_embed = None
        if encoder_out is not None and len(encoder_out["encoder_out"]) > 0:
            enc = encoder_out["encoder_out"][0]
            assert enc.size()[1] == bsz, f"Expected enc.shape == (t, {bsz}, c) got {enc.shape}"
        if encoder_out is not None and len(encoder_out["encoder_padding_mask"]) > 0:
            padding_mask = encoder_out["encoder_padding_mask"][0]
        if encoder_out is not None and len(encoder_out["position_embeddings"]) > 0:
            src_pos_embed = encoder_out["position_embeddings"][0]
        tgt_embed, tgt_pos_embed = adaptor_output.embed, adaptor_output.pos_embed
        self_attn_padding_mask = adaptor_output.masks
        all_self_attn_bias = adaptor_output.self_attn_bias
        pos_shared = self._pos_shared(adaptor_output, encoder_out)
        if not self.cfg.entangle_position_embedding:
            cross_abs_pos_bias = self.get_cross_pos_info(tgt_embed, tgt_pos_embed, src_pos_embed=src_pos_embed, shared=pos_shared)
            if isinstance(cross_abs_pos_bias, ops.SharedBias):
                if incremental_state is not None:                    # (a decoding step slices the bias: the tensor form)
                    cross_abs_pos_bias = ops.expand_shared_bias(cross_abs_pos_bias.t, bsz, tgt_pos_embed.shape[1], src_pos_embed.shape[1])
            else:
                cross_abs_pos_bias = cross_abs_pos_bias.reshape(-1, *cross_abs_pos_bias.size()[-2:])
        else:
            cross_abs_pos_bias = None
        if incremental_state is not None:                            # one step: the last target position only (:447-450)
            tgt_embed = tgt_embed[:, -1:]
            cross_abs_pos_bias = cross_abs_pos_bias[:, -1:, :] if cross_abs_pos_bias is not None else None
            self_attn_padding_mask = self_attn_padding_mask[:, -1:] if self_attn_padding_mask is not None else None
        x = tgt_embed.transpose(0, 1)                                 # T x B x C (view)
        attn = None
        inner_states: List[Optional[Tensor]] = [x] if return_all_hiddens else []
        decoder_attentions, cross_attentions = [], []
        chain = LayerChain()
        layers = kept_layers(self.layers, self.layerdrop, self.training)
        cross_kv = self._cross_kv(enc, incremental_state) if len(layers) == len(self.layers) else None   # (dropped layers: own projections)
        cross_biases = ops.fan_out_bias(cross_abs_pos_bias, len(layers))    # one view per layer: their gradients are summed in one launch
        for idx, layer in enumerate(layers):
            chain.next_ln = layers[idx + 1].self_attn_layer_norm if idx + 1 < len(layers) else self.layer_norm
            self_attn_mask = (self.buffered_future_mask(x) if incremental_state is None and not full_context_alignment
                              else None)
            if self.cfg.use_self_attn_bias:
                b = all_self_attn_bias[0 if self.cfg.share_attn_bias else idx]
                if isinstance(b, ops.SharedBias) and incremental_state is None:
                    self_attn_bias = b
                else:
                    if isinstance(b, ops.SharedBias):
                        b = ops.expand_shared_bias(b.t, bsz, b.t.shape[1], b.t.shape[2])
                    self_attn_bias = b.view(-1, *b.size()[-2:])
                if incremental_state is not None:
                    self_attn_bias = self_attn_bias[:, -1:, :]        # the new position's row against every cached key
            else:
                self_attn_bias = False                                # forces the slow attention path, :477
            x, layer_self_attn, layer_cross_attn = layer(
                x, enc, padding_mask, incremental_state, self_attn_mask=self_attn_mask,
                self_attn_padding_mask=self_attn_padding_mask,
                need_attn=bool((idx == alignment_layer) or return_all_attention_weights),
                need_head_weights=bool(idx == alignment_layer), self_attn_bias=self_attn_bias,
                cross_attn_bias=cross_biases[idx], modal_mask=adaptor_output.modal_mask, chain=chain, cross_kv=cross_kv)
            if return_all_attention_weights:
                decoder_attentions.append(layer_self_attn)
                cross_attentions.append(layer_cross_attn)
            inner_states.append(x)
            if layer_self_attn is not None and idx == alignment_layer:
                attn = layer_self_attn                                # [A,B,Tt,Ts]; (sic) this is the CROSS attention, :479
        if attn is not None:
            A, B_, Tt, Ts = attn.shape
            w = attn.detach()
            if alignment_heads is not None:
                w = w[:alignment_heads]
                A = w.shape[0]
            # mean over heads (:501-506) with the head-mean kernel: [A,B,T,S] view -> [B,A,T,S] storage
            attn = K.mean_heads(w.transpose(0, 1).reshape(B_ * A, Tt, Ts), B_, A)
        normed = chain.take()
        if normed is not None:
            x = normed                                                # the last layer's join already applied layer_norm
        elif self.layer_norm is not None:
            x = self.layer_norm(x)
        x = x.transpose(0, 1)                                         # T x B x C -> B x T x C
        if self.project_out_dim is not None:
            x = self.project_out_dim(x)
        return x, {"attn": [attn], "inner_states": inner_states, "decoder_attentions": decoder_attentions,
                   "cross_attentions": cross_attentions}

    def _extract_features_packed(self, adaptor_output, encoder_out, pack, incremental_state, full_context_alignment, wants_extras):
        """extract_features on packed rows (ofasys_amd/packing.py): same layers, same order; the self-attention is causal inside
        each sample's segment, the cross-attention reads the packed encoder rows of the same sample.  Attention maps are not
        produced (no [B,A,Tt,Ts] tensor exists in this mode): extra["attn"] is [None]."""
        _check_packable(self.cfg, wants_extras or incremental_state is not None or full_context_alignment)
        from ..packing import causal_tag
        enc = encoder_out["encoder_out"][0]                           # [enc rows, 1, C]
        x = ops.pack_rows(adaptor_output.embed, pack.dec_index, pack.dec_inverse).transpose(0, 1)       # [dec rows, 1, C]
        tag = causal_tag(x.device)
        self_bias = cross_bias = None
        if self.cfg.use_self_attn_bias:
            self_bias = adaptor_output.self_attn_bias
            _packed_bias_ok(self_bias, pack.dec_prefix, "decoder")
        if not self.cfg.entangle_position_embedding:
            if not (self._pos_shared(adaptor_output, encoder_out) and pack.dec_prefix and pack.enc_prefix):
                _packed_bias_ok([None], False, "cross-attention")
            cross_bias = self.get_cross_pos_info(None, adaptor_output.pos_embed, src_pos_embed=encoder_out["position_embeddings"][0],
                                                 shared=True)
        chain = LayerChain()
        layers = kept_layers(self.layers, self.layerdrop, self.training)
        cross_kv = self._cross_kv(enc, incremental_state) if len(layers) == len(self.layers) else None
        cross_biases = ops.fan_out_bias(cross_bias, len(layers))
        for idx, layer in enumerate(layers):
            chain.next_ln = layers[idx + 1].self_attn_layer_norm if idx + 1 < len(layers) else self.layer_norm
            sb = self_bias[0 if self.cfg.share_attn_bias else idx] if self_bias is not None else False
            x, _, _ = layer(x, enc, pack.cross, None, self_attn_mask=tag, self_attn_padding_mask=pack.dec_self, need_attn=False,
                            need_head_weights=False, self_attn_bias=sb, cross_attn_bias=cross_biases[idx],
                            modal_mask=adaptor_output.modal_mask, chain=chain, cross_kv=cross_kv)
        normed = chain.take()
        if normed is not None:
            x = normed
        elif self.layer_norm is not None:
            x = self.layer_norm(x)
        x = x.transpose(0, 1)                                         # [1, dec rows, C]
        if self.project_out_dim is not None:
            x = self.project_out_dim(x)
        return x, {"attn": [None], "inner_states": [], "decoder_attentions": [], "cross_attentions": []}

    def reorder_incremental_state_scripting(self, incremental_state, new_order):
        """Beam reorder of every attention cache (model/incremental_decoder.py:81-96)."""
        for module in self.modules():
            if module is not self and hasattr(module, "reorder_incremental_state"):
                result = module.reorder_incremental_state(incremental_state, new_order)
                if result is not None:
                    incremental_state = result
        return incremental_state

    def max_positions(self):
        return self.cfg.max_target_positions

    def buffered_future_mask(self, tensor):
        """triu(-inf, 1) [T,T] (model/transformer.py:528-539), tagged so MultiheadAttention applies causality inside the
        kernel instead of reading a T x T mask from HBM."""
        dim = tensor.size(0)
        if self._future_mask.size(0) < dim or self._future_mask.device != tensor.device or \
                self._future_mask.dtype != tensor.dtype:
            m = torch.triu(torch.full((dim, dim), float("-inf"), dtype=tensor.dtype, device=tensor.device), 1)
            self._future_mask = m
        out = self._future_mask[:dim, :dim]
        out._ofa_causal = True
        return out
