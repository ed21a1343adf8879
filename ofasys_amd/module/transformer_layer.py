"""TransformerEncoderLayer / TransformerDecoderLayer with the reference's names, signatures and state-dict keys
(module/transformer_layer.py:18-209, 212-495), on the gfx950 kernels.

Fusions relative to the reference's op-by-op graph (same arithmetic, fewer HBM round trips):
  * dropout + residual add                      -> one kernel   (transformer_layer.py:181-182, 203-206)
  * GELU + ffn_layernorm (LayerNorm over 4D)     -> one kernel   (:194-197); the pre-GELU fc1 output is what is saved
  * q/k/v/out projections, fc1, fc2              -> MFMA GEMM with bias in the epilogue
  * [attn_ln] + dropout + residual add + the NEXT LayerNorm (this layer's final_layer_norm, or the next layer's first
    pre-LN handed in through a LayerChain) -> one "residual join" kernel each way (csrc/join.hip); pre-LN stacks only
Only the configuration space of OFASys' GeneralistModel is implemented: pre- or post-LN, scale_attn / scale_fc /
scale_heads / scale_resids, modal_ffn (one FFN expert per modality, routed by the adaptor's modality mask);
cross_self_attention is refused loudly.
"""
import copy
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..preprocessor.instruction import ModalityType
from .layers import Dropout, DropPath, LayerNorm, OfaLinear
from .multihead_attention import MultiheadAttention


class LayerChain:
    """Threads `LN_first(x)` from one layer's closing residual join into the next layer of a pre-LN stack.

    The stack sets `next_ln` to the LayerNorm that will consume this layer's output first (the next layer's
    self_attn_layer_norm, or the stack's final layer_norm); the layer leaves that LayerNorm's output in `normed` and the
    consumer takes it instead of normalising again.  Same arithmetic as the reference's op-by-op order."""
    __slots__ = ("normed", "next_ln")

    def __init__(self):
        self.normed = None
        self.next_ln = None

    def take(self):
        z, self.normed = self.normed, None
        return z


def _act_name(cfg):
    return getattr(cfg, "activation_fn", "gelu")


def _activation(name, h):
    """The FFN activation by its fairseq name (module/utils.py get_activation_fn): gelu (OFASys' default: erf form, fused with the
    LayerNorm behind it where the layer allows), relu (the kernel of the ResNet stack), linear; the others are not implemented."""
    if name == "gelu":
        return ops.gelu(h)
    if name == "relu":
        return ops.relu(h)
    if name == "linear":
        return h
    raise NotImplementedError(f"activation_fn={name!r}: gelu, relu and linear are implemented")


class _FFNMixin:
    def _joinable(self):
        """The fused residual joins cover the pre-LN layer without scale_resids / DropPath (OFASys' defaults)."""
        return (self.normalize_before and self.w_resid is None and not self.modal_ffn
                and (self.drop_path.drop_prob == 0.0 or not self.training))

    def _join(self, x, residual, ln_a, ln_b, x_bias=None):
        """(residual + dropout(ln_a(x)), ln_b(that)) -- :181-186 / :203-208 fused; ln_a / ln_b may be None.
        x_bias: see _join_owns_bias."""
        return ops.residual_join(x, residual, ln_a, self.dropout_module.p, self.training, ln_b,
                                 eps=(ln_b or ln_a or self.final_layer_norm).eps, x_bias=x_bias)

    def _join_owns_bias(self, bias, ln_a, ln_b):
        """May the join that consumes a Linear's output also produce that Linear's bias gradient (the column sums of dx
        are free in the join's backward kernel; a separate pass re-reads the whole gradient)?  Training with gradient
        sinks on every parameter involved only."""
        ps = [bias] + [q for ln in (ln_a, ln_b) if ln is not None for q in (ln.weight, ln.bias)]
        return self.training and ops.join_takes_bias_grad(*ps)

    # ---- modal_ffn (transformer_layer.py:50-54, 116-130 / 300-304, 335-349; sparse_dispatcher.py:44-110): one copy of fc1 / fc2 per
    # ModalityType, every row multiplied by the expert of its modality.  The reference builds one-hot gates from the adaptor's
    # [B, T] modality mask FLATTENED BATCH-MAJOR and applies them to the rows of x flattened TIME-MAJOR ([T, B, D]): row
    # r = t*B + b takes the modality of mask.view(-1)[r] = mask[r // T, r % T].  That pairing is reproduced as it is (it is what
    # a reference checkpoint was trained with; tests/golden/tiny_modal_ffn.npz is a reference run).  Every column of the mask
    # belongs to one slot, so the routing is a function of (B, T, slot modalities and widths) alone: the row permutation that
    # groups the rows by expert is built on the host once per batch structure -- no nonzero() / tolist() sync as in the
    # dispatcher, hipGraph-safe -- and each expert is one GEMM over a contiguous row range.  The reference's trailing x.half()
    # (:129) is a no-op in an fp16 model, the only one it runs in; other dtypes are kept here.
    def _init_experts(self):
        self.modal_ffn = bool(getattr(self.cfg, "modal_ffn", False))
        if self.modal_ffn:
            self.experts_num = len(ModalityType)
            self.experts_fc1 = nn.ModuleList([copy.deepcopy(self.fc1) for _ in range(self.experts_num)])
            self.experts_fc2 = nn.ModuleList([copy.deepcopy(self.fc2) for _ in range(self.experts_num)])

    _MODAL_PLANS = {}

    @classmethod
    def _modal_plan(cls, cols, B, T, device):
        """cols: ((expert, width), ...) of the mask's columns -> (perm, inverse, [(expert, start, end)]) for [T*B] rows."""
        key = (cols, B, T, str(device))
        plan = cls._MODAL_PLANS.get(key)
        if plan is None:
            col_expert = [e for e, w in cols for _ in range(w)][:T]
            assert len(col_expert) == T, (cols, T)
            row_expert = [col_expert[r % T] for r in range(T * B)]          # (sic) see above
            order = sorted(range(T * B), key=lambda r: (row_expert[r], r))
            inverse = [0] * (T * B)
            for pos, r in enumerate(order):
                inverse[r] = pos
            ranges, start = [], 0
            for e in sorted(set(row_expert)):
                n = row_expert.count(e)
                ranges.append((e, start, start + n))
                start += n
            # the same permutation for rows stored batch-major (the [T, B, D] VIEW of [B, T, D] memory the stacks pass around):
            # logical row r = t*B + b lives at b*T + t
            bm = lambda r: (r % B) * T + r // B
            order_b = [bm(r) for r in order]
            inverse_b = [0] * (T * B)
            for r in range(T * B):
                inverse_b[bm(r)] = inverse[r]
            dev = lambda v: torch.tensor(v, dtype=torch.long, device=device)
            plan = (dev(order), dev(inverse), ranges, dev(order_b), dev(inverse_b))
            cls._MODAL_PLANS[key] = plan
        return plan

    def _modal_linear(self, modal_mask, x, experts):
        T, B, D = x.shape
        cols = getattr(modal_mask, "_ofa_cols", None)
        if cols is None:
            raise ValueError("modal_ffn: the modality mask must come from OFAGeneralAdaptor (it carries the slot layout)")
        perm, inverse, ranges, perm_b, inverse_b = self._modal_plan(cols, B, T, x.device)
        batch_major = x.transpose(0, 1).is_contiguous()                     # keep the memory layout of the input: the row kernels
        if batch_major:                                                     # downstream add tensors element by element in memory order
            xs = ops.PackRowsFn.apply(x.transpose(0, 1).reshape(B * T, D), perm_b, inverse_b)
        else:
            xs = ops.PackRowsFn.apply(x.contiguous().view(T * B, D), perm, inverse)   # rows grouped by expert
        ys = [experts[e](xs[a:b]) for e, a, b in ranges]
        y = ys[0] if len(ys) == 1 else torch.cat(ys, 0)
        if batch_major:
            return ops.PackRowsFn.apply(y, inverse_b, perm_b).view(B, T, -1).transpose(0, 1)
        return ops.PackRowsFn.apply(y, inverse, perm).view(T, B, -1)

    def _decoder_modal_mask(self, modal_mask, x):
        """The decoder hands `modal_mask[:x.shape[1], :x.shape[0]]` to the experts (transformer_layer.py:477, 486)."""
        if not self.modal_ffn or modal_mask is None:
            return modal_mask
        m = modal_mask[:x.shape[1], :x.shape[0]]
        m._ofa_cols = modal_mask._ofa_cols                                  # (_modal_plan keeps the first T columns)
        return m

    def _upgrade_experts(self, state_dict, name):
        """Keys of this layer that a checkpoint lacks are filled from the module: an expert from the module's current fc1 / fc2
        (transformer_layer.py:105-114)."""
        if not self.modal_ffn:
            return
        prefix = name + "." if name != "" else ""
        own = self.state_dict()
        for k in own:
            if prefix + k in state_dict:
                continue
            src = k
            for i in range(self.experts_num):
                src = src.replace(f"experts_fc1.{i}.", "fc1.").replace(f"experts_fc2.{i}.", "fc2.")
            state_dict[prefix + k] = own[src]

    def _ffn(self, x, normed=None, chain=None, modal_mask=None):
        """residual + dropout(fc2(ffn_ln(act_dropout(act(fc1(LN(x)))))))   -- :186-208 / :471-494.
        normed: final_layer_norm(x) when the preceding join already produced it; chain: see LayerChain."""
        if normed is not None:
            residual, x = x, normed
        elif self.normalize_before:
            residual, x = self.final_layer_norm.fork(x)
        else:
            residual = x
        act_p = self.activation_dropout_module.p if self.training else 0.0
        if self.modal_ffn:
            x = _activation(self._act, self._modal_linear(modal_mask, x, self.experts_fc1))
            x = self.activation_dropout_module(x)
            if self.ffn_layernorm is not None:
                x = self.ffn_layernorm(x)
        elif self.ffn_layernorm is not None and act_p == 0.0 and self._gelu and self.fc1.bias is not None:
            x = ops.linear_gelu_layer_norm(x, self.fc1.weight, self.fc1.bias, self.ffn_layernorm.weight,
                                           self.ffn_layernorm.bias, self.ffn_layernorm.eps)
        else:
            h = self.fc1(x)
            x = _activation(self._act, h)
            x = self.activation_dropout_module(x)
            if self.ffn_layernorm is not None:
                x = self.ffn_layernorm(x)
        if normed is not None:
            next_ln = chain.next_ln if chain is not None else None
            own = self._join_owns_bias(self.fc2.bias, None, next_ln)
            x = self.fc2(x, skip_bias_grad=own)
            x, z = self._join(x, residual, None, next_ln, x_bias=self.fc2.bias if own else None)
            if chain is not None:
                chain.normed = z
            return x
        x = self._modal_linear(modal_mask, x, self.experts_fc2) if self.modal_ffn else self.fc2(x)
        if self.w_resid is not None:
            residual = ops.mul_rowvec(residual, self.w_resid)                   # :204-205
        x = ops.dropout_add(self.drop_path(x), residual, self.dropout_module.p, self.training)
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x


class TransformerEncoderLayer(nn.Module, _FFNMixin):
    def __init__(self, args, drop_path_rate=0.0):
        super().__init__()
        cfg = args
        self.cfg = cfg
        self.embed_dim = cfg.encoder.embed_dim
        self.self_attn = self.build_self_attention(self.embed_dim, cfg)
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        self.dropout_module = Dropout(cfg.dropout, module_name=self.__class__.__name__)
        self._act = _act_name(cfg)
        self._gelu = self._act == "gelu"
        activation_dropout_p = cfg.activation_dropout
        if activation_dropout_p == 0:
            activation_dropout_p = cfg.relu_dropout or 0
        self.activation_dropout_module = Dropout(float(activation_dropout_p), module_name=self.__class__.__name__)
        self.normalize_before = cfg.encoder.normalize_before
        self.fc1 = OfaLinear(self.embed_dim, cfg.encoder.ffn_embed_dim)
        self.fc2 = OfaLinear(cfg.encoder.ffn_embed_dim, self.embed_dim)
        self._init_experts()
        self.attn_ln = LayerNorm(self.embed_dim) if cfg.scale_attn else None
        self.nh = self.self_attn.num_heads
        self.head_dim = self.self_attn.head_dim
        self.ffn_layernorm = LayerNorm(cfg.encoder.ffn_embed_dim) if cfg.scale_fc else None
        self.w_resid = nn.Parameter(torch.ones(self.embed_dim), requires_grad=True) if cfg.scale_resids else None
        self.final_layer_norm = LayerNorm(self.embed_dim)
        self.drop_path = DropPath(float(drop_path_rate), batch_axis=1)

    def build_self_attention(self, embed_dim, cfg):
        return MultiheadAttention(embed_dim, cfg.encoder.attention_heads, dropout=cfg.attention_dropout,
                                  self_attention=True, scale_factor=cfg.attn_scale_factor, scale_heads=cfg.scale_heads,
                                  use_fused=cfg.use_fused)

    def residual_connection(self, x, residual):
        return ops.dropout_add(self.drop_path(x), residual, 0.0, False)

    def forward(self, x, encoder_padding_mask: Optional[Tensor], attn_mask: Optional[Tensor] = None,
                self_attn_bias: Optional[Tensor] = None, need_attn: bool = False, modal_mask=None,
                chain: Optional[LayerChain] = None):
        """x: (seq_len, batch, embed_dim); see transformer_layer.py:141-158.  chain: LayerChain of the enclosing stack."""
        if attn_mask is not None:
            attn_mask = attn_mask.masked_fill(attn_mask.to(torch.bool), -1e8 if x.dtype == torch.float32 else -1e4)
        join = self._joinable()
        # (the earliest node of the layer: autograd runs it last, and it launches the layer's weight gradients as one group)
        x = ops.wgrad_boundary(x)
        normed = chain.take() if chain is not None else None
        if normed is not None and self.normalize_before:
            residual, x = x, normed
        elif self.normalize_before:
            residual, x = self.self_attn_layer_norm.fork(x)
        else:
            residual = x
        own = join and self._join_owns_bias(self.self_attn.out_proj.bias, self.attn_ln, self.final_layer_norm)
        x, self_attn_weights = self.self_attn(query=x, key=x, value=x, key_padding_mask=encoder_padding_mask,
                                              need_weights=need_attn, attn_mask=attn_mask, attn_bias=self_attn_bias,
                                              out_proj_skip_bias_grad=own)
        if join:
            x, h = self._join(x, residual, self.attn_ln, self.final_layer_norm,
                              x_bias=self.self_attn.out_proj.bias if own else None)
            return self._ffn(x, normed=h, chain=chain, modal_mask=modal_mask), self_attn_weights
        if self.attn_ln is not None:
            x = self.attn_ln(x)
        x = ops.dropout_add(self.drop_path(x), residual, self.dropout_module.p, self.training)   # :181-182
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        x = self._ffn(x, modal_mask=modal_mask)
        return x, self_attn_weights

    def upgrade_state_dict_named(self, state_dict, name):
        layer_norm_map = {"0": "self_attn_layer_norm", "1": "final_layer_norm"}
        for old, new in layer_norm_map.items():
            for m in ("weight", "bias"):
                k = "{}.layer_norms.{}.{}".format(name, old, m)
                if k in state_dict:
                    state_dict["{}.{}.{}".format(name, new, m)] = state_dict.pop(k)
        self._upgrade_experts(state_dict, name)


class TransformerDecoderLayer(nn.Module, _FFNMixin):
    def __init__(self, args, no_encoder_attn=False, add_bias_kv=False, add_zero_attn=False, drop_path_rate=0.0):
        super().__init__()
        cfg = args
        self.cfg = cfg
        if cfg.cross_self_attention:
            raise NotImplementedError("cross_self_attention is not used by OFASys' GeneralistModel")
        self.embed_dim = cfg.decoder.embed_dim
        self.dropout_module = Dropout(cfg.dropout, module_name=self.__class__.__name__)
        self.cross_self_attention = cfg.cross_self_attention
        self.self_attn = self.build_self_attention(self.embed_dim, cfg, add_bias_kv=add_bias_kv, add_zero_attn=add_zero_attn)
        self.self_attn_ln = LayerNorm(self.embed_dim) if cfg.scale_attn else None
        self.cross_attn_ln = LayerNorm(self.embed_dim) if cfg.scale_attn else None
        self.nh = self.self_attn.num_heads
        self.head_dim = self.self_attn.head_dim
        self._act = _act_name(cfg)
        self._gelu = self._act == "gelu"
        activation_dropout_p = cfg.activation_dropout
        if activation_dropout_p == 0:
            activation_dropout_p = cfg.relu_dropout or 0
        self.activation_dropout_module = Dropout(float(activation_dropout_p), module_name=self.__class__.__name__)
        self.normalize_before = cfg.decoder.normalize_before
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        if no_encoder_attn:
            self.encoder_attn = None
            self.encoder_attn_layer_norm = None
        else:
            self.encoder_attn = self.build_encoder_attention(self.embed_dim, cfg)
            self.encoder_attn_layer_norm = LayerNorm(self.embed_dim)
        self.ffn_layernorm = LayerNorm(cfg.decoder.ffn_embed_dim) if cfg.scale_fc else None
        self.w_resid = nn.Parameter(torch.ones(self.embed_dim), requires_grad=True) if cfg.scale_resids else None
        self.fc1 = OfaLinear(self.embed_dim, cfg.decoder.ffn_embed_dim)
        self.fc2 = OfaLinear(cfg.decoder.ffn_embed_dim, self.embed_dim)
        self._init_experts()
        self.final_layer_norm = LayerNorm(self.embed_dim)
        self.need_attn = True
        self.drop_path = DropPath(float(drop_path_rate), batch_axis=1)

    def build_self_attention(self, embed_dim, cfg, add_bias_kv=False, add_zero_attn=False):
        return MultiheadAttention(embed_dim, cfg.decoder.attention_heads, dropout=cfg.attention_dropout,
                                  add_bias_kv=add_bias_kv, add_zero_attn=add_zero_attn,
                                  self_attention=not cfg.cross_self_attention, scale_factor=cfg.attn_scale_factor,
                                  scale_heads=cfg.scale_heads, use_fused=cfg.use_fused)

    def build_encoder_attention(self, embed_dim, cfg):
        return MultiheadAttention(embed_dim, cfg.decoder.attention_heads, kdim=cfg.encoder.embed_dim,
                                  vdim=cfg.encoder.embed_dim, dropout=cfg.attention_dropout,
                                  encoder_decoder_attention=True, scale_factor=cfg.attn_scale_factor,
                                  scale_heads=cfg.scale_heads, use_fused=cfg.use_fused)

    def forward(self, x, encoder_out: Optional[torch.Tensor] = None, encoder_padding_mask: Optional[torch.Tensor] = None,
                incremental_state: Optional[Dict[str, Dict[str, Optional[Tensor]]]] = None,
                prev_self_attn_state: Optional[List[torch.Tensor]] = None,
                prev_attn_state: Optional[List[torch.Tensor]] = None, self_attn_mask: Optional[torch.Tensor] = None,
                self_attn_padding_mask: Optional[torch.Tensor] = None, need_attn: bool = False,
                need_head_weights: bool = False, self_attn_bias: Optional[Tensor] = None,
                cross_attn_bias: Optional[Tensor] = None, modal_mask=None, chain: Optional[LayerChain] = None, cross_kv=None):
        """x: (seq_len, batch, embed_dim); see transformer_layer.py:367-385.  chain: LayerChain of the enclosing stack."""
        if need_head_weights:
            need_attn = True
        for attn_mod, prev in ((self.self_attn, prev_self_attn_state), (self.encoder_attn, prev_attn_state)):
            if prev is not None:                                          # externally supplied cache (:392-402, :446-456)
                assert incremental_state is not None
                saved = {"prev_key": prev[0], "prev_value": prev[1]}
                if len(prev) >= 3:
                    saved["prev_key_padding_mask"] = prev[2]
                attn_mod._set_input_buffer(incremental_state, saved)
        join = self._joinable()
        cross = self.encoder_attn is not None and encoder_out is not None
        x = ops.wgrad_boundary(x)                     # (see the encoder layer)
        normed = chain.take() if chain is not None else None
        if normed is not None and self.normalize_before:
            residual, x = x, normed
        elif self.normalize_before:
            residual, x = self.self_attn_layer_norm.fork(x)
        else:
            residual = x
        ln_next = self.encoder_attn_layer_norm if cross else self.final_layer_norm
        own = join and self._join_owns_bias(self.self_attn.out_proj.bias, self.self_attn_ln, ln_next)
        x, self_attn_weights = self.self_attn(query=x, key=x, value=x, key_padding_mask=self_attn_padding_mask,
                                              incremental_state=incremental_state, need_weights=need_attn,
                                              attn_mask=self_attn_mask, attn_bias=self_attn_bias, out_proj_skip_bias_grad=own)
        h = None
        if join:
            x, h = self._join(x, residual, self.self_attn_ln, ln_next, x_bias=self.self_attn.out_proj.bias if own else None)
        else:
            if self.self_attn_ln is not None:
                x = self.self_attn_ln(x)
            x = ops.dropout_add(self.drop_path(x), residual, self.dropout_module.p, self.training)
            if not self.normalize_before:
                x = self.self_attn_layer_norm(x)
        cross_attn_weights = None
        if cross:
            if h is not None:
                residual, x = x, h
            elif self.normalize_before:
                residual, x = self.encoder_attn_layer_norm.fork(x)
            else:
                residual = x
            own = join and self._join_owns_bias(self.encoder_attn.out_proj.bias, self.cross_attn_ln, self.final_layer_norm)
            x, cross_attn_weights = self.encoder_attn(
                query=x, key=encoder_out, value=encoder_out, key_padding_mask=encoder_padding_mask,
                incremental_state=incremental_state, static_kv=True,
                need_weights=need_attn or (not self.training and self.need_attn), need_head_weights=need_head_weights,
                attn_bias=cross_attn_bias, out_proj_skip_bias_grad=own, kv_shared=cross_kv)
            if join:
                x, h = self._join(x, residual, self.cross_attn_ln, self.final_layer_norm,
                                  x_bias=self.encoder_attn.out_proj.bias if own else None)
            else:
                if self.cross_attn_ln is not None:
                    x = self.cross_attn_ln(x)
                x = ops.dropout_add(self.drop_path(x), residual, self.dropout_module.p, self.training)
                if not self.normalize_before:
                    x = self.encoder_attn_layer_norm(x)
        x = self._ffn(x, normed=h, chain=chain, modal_mask=self._decoder_modal_mask(modal_mask, x))
        return x, cross_attn_weights, self_attn_weights       # (sic) the reference returns them in this order, :495

    def make_generation_fast_(self, need_attn: bool = False, **kwargs):
        self.need_attn = need_attn

    def upgrade_state_dict_named(self, state_dict, name):
        layer_norm_map = {"0": "self_attn_layer_norm", "1": "encoder_attn_layer_norm", "2": "final_layer_norm"}
        for old, new in layer_norm_map.items():
            for m in ("weight", "bias"):
                k = "{}.layer_norms.{}.{}".format(name, old, m)
                if k in state_dict:
                    state_dict["{}.{}.{}".format(name, new, m)] = state_dict.pop(k)
        self._upgrade_experts(state_dict, name)
