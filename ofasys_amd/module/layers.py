"""Leaf modules with the reference's names, constructor signatures and state-dict keys, computing through the HIP
kernels (reference: module/layer_norm.py:27-32, module/layer.py:8-23, module/initialize.py:10-40, module/droppath.py).

They subclass the torch containers (nn.Linear / nn.Embedding / nn.LayerNorm) only for parameter bookkeeping, so that
`init_bert_params`' isinstance dispatch and checkpoint interchange behave exactly as in the reference; `forward` is
replaced by the gfx950 path and refuses CPU tensors (no fallback).
"""
import torch
import torch.nn as nn

from .. import ops


class OfaLayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def fork(self, x):
        """(residual, LayerNorm(x)) for the pre-LN pattern `residual = x; x = LN(x)`: one autograd node whose backward adds
        the residual-branch gradient inside the LayerNorm kernel."""
        return ops.layer_norm_fork(x, self.weight, self.bias, self.eps)


def LayerNorm(normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
    """module/layer_norm.py:27-32 (the apex branch is irrelevant here)."""
    assert elementwise_affine, "the hot path only uses affine LayerNorm"
    return OfaLayerNorm(normalized_shape, eps, elementwise_affine)


class OfaLinear(nn.Linear):
    def forward(self, x, alpha=1.0, skip_bias_grad=False):
        """skip_bias_grad: the caller guarantees that a downstream `ops.residual_join(..., x_bias=self.bias)` produces
        the bias gradient (column sums of the output gradient) inside its backward kernel."""
        return ops.linear(x, self.weight, self.bias, alpha, skip_bias_grad)


def Linear(in_features, out_features, bias=True):
    """module/layer.py:18-23."""
    m = OfaLinear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.0)
    return m


class OfaEmbedding(nn.Embedding):
    def forward(self, ids):
        return ops.embedding(ids, self.weight, self.padding_idx)


def Embedding(num_embeddings, embedding_dim, padding_idx=None, zero_init=False):
    """module/layer.py:8-15."""
    m = OfaEmbedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    if zero_init:
        nn.init.constant_(m.weight, 0)
    return m


class Dropout(nn.Module):
    """module/dropout.py counterpart.  Standalone use goes through the dropout kernel; the layers fuse it with the
    residual add instead (ops.dropout_add)."""

    def __init__(self, p, module_name=None):
        super().__init__()
        self.p = p
        self.module_name = module_name
        self.apply_during_inference = False

    def forward(self, x, inplace: bool = False):
        if self.p > 0 and (self.training or self.apply_during_inference):
            return ops.dropout_add(x, None, self.p, True)
        return x


class DropPath(nn.Module):
    """module/droppath.py:13-60 (per-sample stochastic depth; the reference's default rate is 0 = identity)."""

    def __init__(self, drop_prob: float = 0.0, batch_axis: int = 0, scale_by_keep: bool = True):
        super().__init__()
        if drop_prob < 0 or drop_prob > 1:
            raise ValueError("droppath probability has to be between 0 and 1, but got {}".format(drop_prob))
        self.drop_prob = float(drop_prob)
        self.batch_axis = batch_axis
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        if self.batch_axis >= x.ndim:
            raise ValueError("droppath batch_axis has to be less than input.ndim, but got {} >= {}".format(self.batch_axis, x.ndim))
        if x.ndim != 3 or self.batch_axis not in (0, 1):
            raise NotImplementedError("DropPath is implemented for [T,B,C] / [B,T,C] activations")
        return ops.drop_path(x, self.drop_prob, self.batch_axis, self.scale_by_keep)

    def extra_repr(self):
        return "p={}".format(self.drop_prob)


def init_bert_params(module):
    """module/initialize.py:10-40: N(0, 0.02) for Linear / Embedding / MHA q,k,v weights, drawn on CPU then copied."""
    from .multihead_attention import MultiheadAttention

    def normal_(data):
        data.copy_(data.cpu().normal_(mean=0.0, std=0.02).to(data.device))

    if isinstance(module, nn.Linear):
        normal_(module.weight.data)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        normal_(module.weight.data)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    if isinstance(module, MultiheadAttention):
        normal_(module.q_proj.weight.data)
        normal_(module.k_proj.weight.data)
        normal_(module.v_proj.weight.data)
