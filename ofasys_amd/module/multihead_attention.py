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
import uuid
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
        self._incremental_state_id = str(uuid.uuid4())     # module/incremental_decoding_utils.py:16-17
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
                attn_bias: Optional[Tensor] = None, out_proj_skip_bias_grad: bool = False,
                kv_shared=None) -> Tuple[Tensor, Optional[Tensor]]:
        """Input shape: Time x Batch x Channel (see the reference docstring, :126-140).  out_proj_skip_bias_grad (not in
        the reference): the caller's residual join produces out_proj's bias gradient (see OfaLinear.forward)."""
        if need_head_weights:
            need_weights = True
        if incremental_state is not None:
            return self._forward_incremental(query, key, key_padding_mask, incremental_state, need_weights, static_kv,
                                             attn_mask, need_head_weights, attn_bias)
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

        # attn_bias: the reference's dense [B*A, T, S] tensor, False (decoder: slow path, no bias, :311), or an ops.SharedBias: a
        # POSITION bias does not depend on the batch row, so the stacks hand it over once, as [A, Tb, Sb] (general.py's [B, A, T, T]
        # is B copies of it); the fused kernels index it by (head, position, position) for every sample -- also over packed rows --
        # and return the batch-summed gradient
        shared = attn_bias.shared_arg() if isinstance(attn_bias, ops.SharedBias) else False   # (True, or the swizzled images)
        bias = attn_bias.t if shared else (attn_bias if torch.is_tensor(attn_bias) else None)
        if bias is not None and not shared:
            bias = bias.reshape(bsz * self.num_heads, tgt_len, src_len)
        causal = False
        if attn_mask is not None:
            if getattr(attn_mask, "_ofa_causal", False):
                causal = True                                                # triu(-inf, 1) from buffered_future_mask
            else:                                                            # arbitrary additive mask: fold into the bias
                if shared:
                    bias, shared = ops.expand_shared_bias(bias, bsz, tgt_len, src_len), False
                m = attn_mask.to(xq.dtype).unsqueeze(0).expand(bsz * self.num_heads, tgt_len, src_len).contiguous()
                bias = m if bias is None else ops.add_rowvec_mask(bias.contiguous(), m)
        if torch.is_tensor(key_padding_mask) and key_padding_mask.dim() == 0:
            key_padding_mask = None
        p_drop = self.dropout_module.p if (self.training or self.dropout_module.apply_during_inference) else 0.0
        fused = (xq.dtype in (torch.bfloat16, torch.float16) and self.head_dim == 64 and p_drop == 0.0 and not need_weights
                 and self.q_proj.bias is not None)
        if bias is not None and bias.dtype != xq.dtype:
            bias = bias.to(xq.dtype)
        probs = None
        if fused and xk is xq and xv is xq:
            # one packed k|v|q projection + fused attention (csrc/gemm_mfma.hip, csrc/attention.hip)
            out = ops.PackedSelfAttentionFn.apply(
                xq, self.k_proj.weight, self.v_proj.weight, self.q_proj.weight, self.k_proj.bias, self.v_proj.bias,
                self.q_proj.bias, bias, key_padding_mask, c_attn, self.num_heads, scale, causal, self._pack, shared)
        elif fused and xv is xk:
            # kv_shared (ops.CrossKVShared, from the decoder stack): k | v of this layer are a column slice of the ONE projection of
            # the encoder output that serves all decoder layers
            layer = self._cross_all[1] if (kv_shared is not None and getattr(self, "_cross_all", None) is not None
                                           and self._cross_all[0] is kv_shared.pack) else None
            out = ops.PackedCrossAttentionFn.apply(
                xq, xk, self.k_proj.weight, self.v_proj.weight, self.q_proj.weight, self.k_proj.bias, self.v_proj.bias,
                self.q_proj.bias, bias, key_padding_mask, c_attn, self.num_heads, scale, self._pack, shared,
                kv_shared if layer is not None else None, layer or 0)
        else:
            if shared:                                                       # exact tier / weights output: the reference's tensor
                if key_padding_mask is not None and not torch.is_tensor(key_padding_mask):
                    raise NotImplementedError("packed (ragged) batches need the fused attention kernels (16-bit, head_dim 64, no "
                                              "attention dropout / weights output): a position bias over packed rows exists only there")
                bias = ops.expand_shared_bias(bias, bsz, tgt_len, src_len)
            q = self.q_proj(xq)
            k = self.k_proj(xk)
            v = self.v_proj(xv)
            out, probs = ops.attention(q, k, v, self.num_heads, scale, bias=bias, key_padding_mask=key_padding_mask,
                                       c_attn=c_attn, causal=causal, dropout_p=p_drop, need_weights=need_weights)
        out = self.out_proj(out, skip_bias_grad=out_proj_skip_bias_grad).transpose(0, 1)   # back to T x B x C (a view)
        attn_weights = None
        if need_weights:
            if need_head_weights:
                attn_weights = probs.view(bsz, self.num_heads, tgt_len, src_len).transpose(1, 0)   # :348
            else:
                attn_weights = K.mean_heads(probs.detach(), bsz, self.num_heads)                   # :349-351
        return out, attn_weights

    # ------------------------------------------------------------------ incremental decoding
    # The cache of one attention module lives in incremental_state[<module uuid>.attn_state] as
    #   {"k", "v": [B, capacity, D] row buffers (the layout the kernels read; appending a step never moves the cache),
    #    "len": valid rows, "kpm": bool [B, capacity] or None}
    # `_get_input_buffer` / `_set_input_buffer` present it in the reference's form (prev_key / prev_value of shape
    # (bsz, heads, len, head_dim), prev_key_padding_mask (bsz, len); multihead_attention.py:254-279, 393-409).
    def _state_key(self):
        return "{}.{}".format(self._incremental_state_id, "attn_state")

    def _cache(self, incremental_state):
        return incremental_state.get(self._state_key()) if incremental_state is not None else None

    def _get_input_buffer(self, incremental_state) -> Dict[str, Optional[Tensor]]:
        c = self._cache(incremental_state)
        if c is None:
            return {}
        B, n, H, hd = c["k"].shape[0], c["len"], self.num_heads, self.head_dim
        out = {"prev_key": c["k"][:, :n].view(B, n, H, hd).transpose(1, 2),
               "prev_value": c["v"][:, :n].view(B, n, H, hd).transpose(1, 2)}
        out["prev_key_padding_mask"] = c["kpm"][:, :n] if c["kpm"] is not None else None
        return out

    def _set_input_buffer(self, incremental_state, buffer: Dict[str, Optional[Tensor]]):
        if incremental_state is None:
            return incremental_state
        if not buffer or "prev_key" not in buffer:
            incremental_state.pop(self._state_key(), None)
            return incremental_state
        pk, pv = buffer["prev_key"], buffer["prev_value"]                    # (bsz, heads, len, head_dim)
        B, H, n, hd = pk.shape
        k = pk.transpose(1, 2).reshape(B, n, H * hd).contiguous()
        v = pv.transpose(1, 2).reshape(B, n, H * hd).contiguous()
        m = buffer.get("prev_key_padding_mask")
        incremental_state[self._state_key()] = {"k": k, "v": v, "len": n, "kpm": m.bool().contiguous() if m is not None else None}
        return incremental_state

    def reorder_incremental_state(self, incremental_state, new_order: Tensor):
        """Beam reorder (multihead_attention.py:393-409): rows follow `new_order`; the static encoder-decoder cache is
        left alone when its batch size already matches (every beam of a sentence holds the same keys)."""
        c = self._cache(incremental_state)
        if c is not None:
            if self.encoder_decoder_attention and c["k"].size(0) == new_order.size(0):
                return incremental_state
            # (whole buffers, spare capacity included: the gather is the one copy of this step, no regrow afterwards)
            inplace = bool(incremental_state.get("__static__")) and c["k"].size(0) == new_order.size(0)
            for name in ("k", "v", "kpm"):
                if c[name] is None:
                    continue
                g = c[name].index_select(0, new_order)
                if inplace:
                    c[name].copy_(g)                      # captured decode steps hold the buffer addresses
                else:
                    c[name] = g
        return incremental_state

    def reset_incremental_state(self, incremental_state):
        """Start a new sequence in the SAME buffers (ofasys_amd.generator.StepDecoder: hipGraphs of the decode steps keep
        the cache addresses): the self-attention cache is emptied, the static encoder-decoder cache is marked stale and
        recomputed in place by the next call."""
        c = self._cache(incremental_state)
        if c is not None:
            if self.encoder_decoder_attention:
                c["stale"] = True
            else:
                c["len"] = 0
                if c["kpm"] is not None:
                    c["kpm"].zero_()
        return incremental_state

    def _decode_pack(self):
        """k|v|q weights / biases as one [3D, D] / [3D] operand for the decode step: the trainer's arena view when the
        parameters are adjacent there, else a concatenated copy cached per parameter version (inference: weights are static)."""
        w, b = self._pack.get("w"), self._pack.get("b")
        if w is not None and b is not None:
            return w, b
        ps = (self.k_proj.weight, self.v_proj.weight, self.q_proj.weight, self.k_proj.bias, self.v_proj.bias, self.q_proj.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cached = getattr(self, "_decode_pack_cache", None)
        if cached is None:
            with torch.no_grad():
                cached = (key, torch.cat(ps[:3], 0).contiguous(), torch.cat(ps[3:], 0).contiguous())
            self._decode_pack_cache = cached
        elif cached[0] != key:                               # parameters changed: refresh IN PLACE (captured decode steps
            with torch.no_grad():                            # hold the addresses; StepDecoder.begin calls this eagerly)
                cached[1].copy_(torch.cat(ps[:3], 0))
                cached[2].copy_(torch.cat(ps[3:], 0))
            cached = (key, cached[1], cached[2])
            self._decode_pack_cache = cached
        return cached[1], cached[2]

    @staticmethod
    def _grow(buf, need, dim=1):
        """Capacity doubling along `dim` (rows are appended in place; a reallocation copies the valid prefix once)."""
        cap = buf.shape[dim]
        if need <= cap:
            return buf
        new_cap = max(need, 2 * cap, 64)
        shape = list(buf.shape)
        shape[dim] = new_cap
        nb = buf.new_zeros(shape)
        nb.narrow(dim, 0, cap).copy_(buf)
        return nb

    def _forward_incremental(self, query, key, key_padding_mask, incremental_state, need_weights, static_kv, attn_mask,
                             need_head_weights, attn_bias):
        """One decoding step (multihead_attention.py:188-353 with incremental_state): always the slow-path arithmetic
        (scaling (head_dim*scale_factor)^-0.5, additive bias row, fp32 softmax, c_attn).  Inference only."""
        tgt_len, bsz, embed_dim = query.size()
        if tgt_len != 1:
            raise NotImplementedError("incremental decoding feeds one target position per call (model/transformer.py:447-450)")
        if attn_mask is not None:
            raise NotImplementedError("attn_mask together with incremental_state (the reference passes None, :464-467)")
        H, D = self.num_heads, self.embed_dim
        xq = query.reshape(bsz, D)                                            # T == 1: [1,B,D] -> [B,D]
        c = self._cache(incremental_state)
        kvq = None
        if (not static_kv and self.self_attention and not torch.is_grad_enabled() and self.q_proj.bias is not None
                and xq.dtype in (torch.bfloat16, torch.float16)):
            # ONE packed k|v|q projection per step instead of three 10-us launches (the step is launch-bound)
            W, Bv = self._decode_pack()
            kvq = K.gemm(xq.contiguous(), W, False, True, bias=Bv)            # [B, 3D]
            q = kvq[:, 2 * D:]
        else:
            q = self.q_proj(xq)
        if static_kv:                                                         # encoder-decoder attention: keys computed once
            assert self.encoder_decoder_attention and not self.self_attention
            if c is None or c.get("stale"):
                assert key is not None
                xk = ops.batch_major(key)                                     # [B,S,D]
                S = xk.shape[1]
                k = self.k_proj(xk).contiguous()
                v = self.v_proj(xk).contiguous()
                m = key_padding_mask.bool().contiguous() if key_padding_mask is not None and key_padding_mask.dim() > 0 else None
                if c is None:
                    c = {"k": k, "v": v, "len": S, "kpm": m, "static": True}
                    incremental_state[self._state_key()] = c
                else:                                                         # a new sequence in the same (static) buffers
                    assert c["k"].shape == k.shape and (c["kpm"] is None) == (m is None)
                    c["k"].copy_(k)
                    c["v"].copy_(v)
                    if m is not None:
                        c["kpm"].copy_(m)
                    c["stale"] = False
        else:
            assert self.self_attention, "incremental decoding: self-attention or static encoder-decoder attention"
            if kvq is not None:
                k_new, v_new = kvq[:, :D], kvq[:, D:2 * D]
            else:
                k_new, v_new = self.k_proj(xq), self.v_proj(xq)               # [B,D]
            cur = key_padding_mask if key_padding_mask is not None and key_padding_mask.dim() > 0 else None
            if c is None:
                cap = int(incremental_state.get("__capacity__", 64))          # (StepDecoder: the whole sequence up front)
                c = {"k": k_new.new_zeros(bsz, cap, D), "v": k_new.new_zeros(bsz, cap, D), "len": 0, "kpm": None}
                incremental_state[self._state_key()] = c
            n = c["len"]
            c["k"], c["v"] = self._grow(c["k"], n + 1), self._grow(c["v"], n + 1)
            c["k"][:, n].copy_(k_new)
            c["v"][:, n].copy_(v_new)
            # key padding across steps (multihead_attention.py:356-391): absent masks count as "not padded"
            if cur is not None or c["kpm"] is not None:
                if c["kpm"] is None:
                    c["kpm"] = torch.zeros(bsz, c["k"].shape[1], dtype=torch.bool, device=query.device)
                c["kpm"] = self._grow(c["kpm"], n + 1)
                if cur is not None:
                    c["kpm"][:, n] = cur.reshape(bsz).bool()
                else:
                    c["kpm"][:, n] = False
            c["len"] = n + 1
        S = c["len"]
        if isinstance(attn_bias, ops.SharedBias):
            raise NotImplementedError("incremental decoding takes the new position's bias ROW as a [B*A, 1, S] tensor")
        bias = attn_bias if torch.is_tensor(attn_bias) else None
        if bias is not None:
            bias = bias.reshape(bsz * H, S)
        out, probs = K.attn_decode(q, c["k"], c["v"], S, H, self.scaling, bias=bias, kpm=c["kpm"], c_attn=self.c_attn,
                                   need_probs=need_weights or need_head_weights)
        out = self.out_proj(out).view(1, bsz, D)
        attn_weights = None
        if need_weights or need_head_weights:
            if need_head_weights:
                attn_weights = probs.view(bsz, H, 1, S).transpose(1, 0)
            else:
                attn_weights = K.mean_heads(probs.view(bsz * H, 1, S), bsz, H)
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
