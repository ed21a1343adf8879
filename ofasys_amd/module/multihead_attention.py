"""MultiheadAttention with the reference's constructor, forward signature and parameter names
(module/multihead_attention.py:21-353), computing through the gfx950 kernels.

Two regimes, as in the reference:
  * fast path  (:155-186)  attn_bias is None, no incremental state, not static_kv  -> F.multi_head_attention_forward
    semantics: scaling head_dim**-0.5, NO c_attn, softmax in the input dtype;
  * slow path  (:188-353)  everything else: scaling (head_dim*scale_factor)**-0.5, additive attn_bias [B*A,T,S],
    causal mask, key padding, fp32 softmax, per-head c_attn scale.
Both run the same fused kernel (bf16) or the exact unfused kernels (fp32 / need_weights / attention dropout);
only `scale`, `c_attn` and `bias` differ.  Tensors at the boundary are Time x Batch x Channel like the reference.
"""
import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import kernels as K
from .. import ops
from .layers import Dropout, OfaLinear


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True, add_bias_kv=False,
                 add_zero_attn=False, self_attention=False, encoder_decoder_attention=False, scale_factor=2,
                 scale_heads=False, use_fused=True):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.num_heads = num_heads
        self.dropout_module = Dropout(dropout, module_name=self.__class__.__name__)
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = float(self.head_dim * scale_factor) ** -0.5          # :54
        self.self_attention = self_attention
        self.encoder_decoder_attention = encoder_decoder_attention
        self.c_attn = nn.Parameter(torch.ones((self.num_heads,)), requires_grad=True) if scale_heads else None  # :58
        assert not self.self_attention or self.qkv_same_dim
        if add_bias_kv or add_zero_attn:
            raise NotImplementedError("add_bias_kv / add_zero_attn are not used by OFASys' GeneralistModel")
        self.k_proj = OfaLinear(self.kdim, embed_dim, bias=bias)
        self.v_proj = OfaLinear(self.vdim, embed_dim, bias=bias)
        self.q_proj = OfaLinear(embed_dim, embed_dim, bias=bias)
        self.out_proj = OfaLinear(embed_dim, embed_dim, bias=bias)
        self.bias_k = self.bias_v = None
        self.add_zero_attn = False
        self.use_fused = use_fused      # kept for config compatibility; the HIP path is always "fused"
        # filled by ofasys_amd.trainer.FlatParams when k|v|q (or k|v) weights sit next to each other in the flat arena:
        # zero-copy packed projection weights / gradient sinks for the single N = 3D (2D) GEMM
        self._pack = {}
        self.reset_parameters()

    def reset_parameters(self):                                             # :93-111
        gain = 1 / math.sqrt(2) if self.qkv_same_dim else 1.0
        nn.init.xavier_uniform_(self.k_proj.weight, gain=gain)
        nn.init.xavier_uniform_(self.v_proj.weight, gain=gain)
        nn.init.xavier_uniform_(self.q_proj.weight, gain=gain)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key: Optional[Tensor], value: Optional[Tensor], key_padding_mask: Optional[Tensor] = None,
                incremental_state: Optional[Dict[str, Dict[str, Optional[Tensor]]]] = None, need_weights: bool = True,
                static_kv: bool = False, attn_mask: Optional[Tensor] = None, need_head_weights: bool = False,
                attn_bias: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """Input shape: Time x Batch x Channel (see the reference docstring, :126-140)."""
        if need_head_weights:
            need_weights = True
        if incremental_state is not None:
            raise NotImplementedError("incremental decoding (KV cache) is outside the train-step hot path "
                                      "(SURVEY.md section 8f-4)")
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim, f"query dim {embed_dim} != {self.embed_dim}"
        fast = (not static_kv) and attn_bias is None                       # :155-162
        xq = ops.batch_major(query)                                         # [B,T,D]
        if self.self_attention or key is None or key is query:
            xk = xq
        else:
            xk = ops.batch_major(key)
        if not (self.self_attention or self.encoder_decoder_attention) and value is not key:
            xv = ops.batch_major(value)
        else:
            xv = xk
        src_len = xk.shape[1]
        scale = (1.0 / math.sqrt(self.head_dim)) if fast else self.scaling
        c_attn = None if fast else self.c_attn

        bias = attn_bias if torch.is_tensor(attn_bias) else None            # decoder passes False: slow path, no bias (:311)
        if bias is not None:
            bias = bias.reshape(bsz * self.num_heads, tgt_len, src_len)
        causal = False
        if attn_mask is not None:
            if getattr(attn_mask, "_ofa_causal", False):
                causal = True                                                # triu(-inf, 1) from buffered_future_mask
            else:                                                            # arbitrary additive mask: fold into the bias
                m = attn_mask.to(xq.dtype).unsqueeze(0).expand(bsz * self.num_heads, tgt_len, src_len).contiguous()
                bias = m if bias is None else ops.add_rowvec_mask(bias.contiguous(), m)
        if key_padding_mask is not None and key_padding_mask.dim() == 0:
            key_padding_mask = None
        p_drop = self.dropout_module.p if (self.training or self.dropout_module.apply_during_inference) else 0.0
        fused = (xq.dtype == torch.bfloat16 and self.head_dim == 64 and p_drop == 0.0 and not need_weights
                 and self.q_proj.bias is not None)
        if bias is not None and bias.dtype != xq.dtype:
            bias = bias.to(xq.dtype)
        probs = None
        if fused and xk is xq and xv is xq:
            # one packed k|v|q projection + fused attention (csrc/gemm_mfma.hip, csrc/attention.hip)
            out = ops.PackedSelfAttentionFn.apply(
                xq, self.k_proj.weight, self.v_proj.weight, self.q_proj.weight, self.k_proj.bias, self.v_proj.bias,
                self.q_proj.bias, bias, key_padding_mask, c_attn, self.num_heads, scale, causal, self._pack)
        elif fused and xv is xk:
            out = ops.PackedCrossAttentionFn.apply(
                xq, xk, self.k_proj.weight, self.v_proj.weight, self.q_proj.weight, self.k_proj.bias, self.v_proj.bias,
                self.q_proj.bias, bias, key_padding_mask, c_attn, self.num_heads, scale, self._pack)
        else:
            q = self.q_proj(xq)
            k = self.k_proj(xk)
            v = self.v_proj(xv)
            out, probs = ops.attention(q, k, v, self.num_heads, scale, bias=bias, key_padding_mask=key_padding_mask,
                                       c_attn=c_attn, causal=causal, dropout_p=p_drop, need_weights=need_weights)
        out = self.out_proj(out).transpose(0, 1)                            # back to T x B x C (a view)
        attn_weights = None
        if need_weights:
            if need_head_weights:
                attn_weights = probs.view(bsz, self.num_heads, tgt_len, src_len).transpose(1, 0)   # :348
            else:
                attn_weights = K.mean_heads(probs.detach(), bsz, self.num_heads)                   # :349-351
        return out, attn_weights

    def upgrade_state_dict_named(self, state_dict, name):
        prefix = name + "." if name != "" else ""
        items_to_add, keys_to_remove = {}, []
        for k in state_dict.keys():
            if k.endswith(prefix + "in_proj_weight"):                      # :411-438
                dim = int(state_dict[k].shape[0] / 3)
                items_to_add[prefix + "q_proj.weight"] = state_dict[k][:dim]
                items_to_add[prefix + "k_proj.weight"] = state_dict[k][dim:2 * dim]
                items_to_add[prefix + "v_proj.weight"] = state_dict[k][2 * dim:]
                keys_to_remove.append(k)
                k_bias = prefix + "in_proj_bias"
                if k_bias in state_dict.keys():
                    dim = int(state_dict[k].shape[0] / 3)
                    items_to_add[prefix + "q_proj.bias"] = state_dict[k_bias][:dim]
                    items_to_add[prefix + "k_proj.bias"] = state_dict[k_bias][dim:2 * dim]
                    items_to_add[prefix + "v_proj.bias"] = state_dict[k_bias][2 * dim:]
                    keys_to_remove.append(prefix + "in_proj_bias")
        for k in keys_to_remove:
            del state_dict[k]
        for key, value in items_to_add.items():
            state_dict[key] = value
