"""Ragged row packing: run the encoder-decoder stack on the NON-PAD positions only.

The reference pads every sample of a micro-batch to the longest one, computes all positions and masks the padding
(preprocessor/utils.py:75-113 collate_tokens, model/transformer.py:110-112 pad zeroing, multihead_attention.py:319-326 key-padding
-inf); 11% of the encoder rows and 37% of the decoder rows of the cfg-2 batch are such padding.  The metric counts non-pad positions
(SURVEY.md section 8d), and no non-pad output depends on a padded position, so this build may drop them:

    [B, T, D] padded rows --gather--> [R, D] packed rows   (sample b = rows off_b .. off_b + len_b - 1, off_b a multiple of 8,
                                                            R a multiple of `bucket`; the filler rows are zero and inert)
    every row-wise kernel (LayerNorm, GEMMs, GELU, residual joins, criterion) simply sees fewer rows;
    attention takes the segment table {q_off, q_len, k_off, k_len} per sample (csrc/attention.hip, ragged mode).

The default (biased-attention) configuration packs too since round 3: the position bias is no longer a dense [B,A,T,T] tensor laid
out by padded position but ONE [A,T,T] matrix per layer shared by the batch (ops.SharedBias: batch-invariant positions), which the
attention kernels index by the position INSIDE the sample; that needs packed row r of a sample to BE padded position r, i.e. each
sample's valid positions must be a prefix of its padded row (one ragged slot, at the end -- image + text, video + text, text alone;
`enc_prefix` / `dec_prefix`).  R is rounded up to `bucket` rows so that a
handful of hipGraphs cover all batches of a length distribution.  Host side: the plan is built from HOST masks / lengths (no device
sync), shipped with the batch, and is part of the step's static inputs.
"""
from dataclasses import dataclass
from typing import Optional

import torch

ALIGN = 8          # every sample starts on an 8-row boundary (16-byte aligned fp32 lse / delta rows in the attention kernels)


@dataclass
class Segments:
    """Segment table of one attention call over packed rows.  table: int32 [B, 4] = (q_off, q_len, k_off, k_len)."""
    table: torch.Tensor
    batch: int
    rows_q: int
    rows_k: int
    max_q: int
    max_k: int

    def to(self, device):
        return Segments(self.table.to(device, non_blocking=True), self.batch, self.rows_q, self.rows_k, self.max_q, self.max_k)


@dataclass
class PackPlan:
    """Everything the packed forward needs.  *_index: int64 [rows] -- packed row r reads padded row index[r] of the [B*T, D]
    adaptor output (-1: filler row, zero)."""
    enc_index: torch.Tensor
    dec_index: torch.Tensor
    enc_inverse: torch.Tensor      # int64 [B*Ts]: padded row -> packed row (-1: padding), the gather's backward
    dec_inverse: torch.Tensor
    enc_self: Segments
    dec_self: Segments
    cross: Segments
    enc_tokens: int            # non-pad positions (the metric's numerator)
    dec_tokens: int
    enc_prefix: bool = True    # every sample's valid positions are the PREFIX 0 .. len-1 of its padded row (one ragged slot, at the
    dec_prefix: bool = True    # end): packed row r of a sample is then padded position r, which the rel-pos ids are indexed by

    def to(self, device):
        mv = lambda t: t.to(device, non_blocking=True)                       # noqa: E731
        return PackPlan(mv(self.enc_index), mv(self.dec_index), mv(self.enc_inverse), mv(self.dec_inverse),
                        self.enc_self.to(device), self.dec_self.to(device), self.cross.to(device), self.enc_tokens, self.dec_tokens,
                        self.enc_prefix, self.dec_prefix)

    def tensors(self):
        return [self.enc_index, self.dec_index, self.enc_inverse, self.dec_inverse, self.enc_self.table, self.dec_self.table,
                self.cross.table]

    def structure(self):
        """The static part (shapes / launch bounds): two plans with equal structure can share one captured graph."""
        return (self.enc_index.numel(), self.dec_index.numel(), self.enc_self.batch, self.enc_self.max_q, self.dec_self.max_q)


def _layout(mask: torch.Tensor, bucket: int):
    """mask: host bool [B, T], True = padding.  -> (index int64 [R], offsets, lengths)."""
    B, T = mask.shape
    keep = ~mask
    lengths = keep.sum(1).tolist()
    offs, rows = [], 0
    for n in lengths:
        offs.append(rows)
        rows += (n + ALIGN - 1) // ALIGN * ALIGN
    R = max(bucket, (rows + bucket - 1) // bucket * bucket)
    index = torch.full((R,), -1, dtype=torch.int64)
    for b in range(B):
        index[offs[b]:offs[b] + lengths[b]] = torch.nonzero(keep[b]).squeeze(1) + b * T
    inverse = torch.full((B * T,), -1, dtype=torch.int64)
    valid = index >= 0
    inverse[index[valid]] = torch.nonzero(valid).squeeze(1)
    prefix = all(bool(keep[b, :lengths[b]].all()) for b in range(B))
    return index, inverse, offs, lengths, R, prefix


def build_pack_plan(enc_pad_mask: torch.Tensor, dec_pad_mask: torch.Tensor, bucket: int = 256,
                    dec_bucket: Optional[int] = None) -> PackPlan:
    """enc_pad_mask [B, Ts], dec_pad_mask [B, Tt]: HOST bool tensors, True where the position is padding (for right-padded
    text `tokens.eq(pad)`, all-False for image patches).  The valid positions of a sample need not be a prefix (several ragged
    slots concatenated): they are packed in order."""
    assert enc_pad_mask.device.type == "cpu" and dec_pad_mask.device.type == "cpu" and enc_pad_mask.shape[0] == dec_pad_mask.shape[0]
    B = enc_pad_mask.shape[0]
    ei, einv, eo, el, Re, epre = _layout(enc_pad_mask.bool(), bucket)
    di, dinv, do_, dl, Rd, dpre = _layout(dec_pad_mask.bool(), dec_bucket or bucket)

    def table(qo, ql, ko, kl):
        return torch.tensor([[qo[b], ql[b], ko[b], kl[b]] for b in range(B)], dtype=torch.int32)
    pad_to = lambda n: (n + 31) // 32 * 32                                   # noqa: E731  launch bound, coarse on purpose
    mq_e, mq_d = pad_to(max(el)), pad_to(max(dl))
    return PackPlan(ei, di, einv, dinv,
                    Segments(table(eo, el, eo, el), B, Re, Re, mq_e, mq_e),
                    Segments(table(do_, dl, do_, dl), B, Rd, Rd, mq_d, mq_d),
                    Segments(table(do_, dl, eo, el), B, Rd, Re, mq_d, mq_e),
                    int(sum(el)), int(sum(dl)), epre, dpre)


def causal_tag(device):
    """What MultiheadAttention looks for to apply causality inside the kernel (`_ofa_causal`); no T x T mask exists in packed mode."""
    t = torch.empty(0, device=device)
    t._ofa_causal = True
    return t
