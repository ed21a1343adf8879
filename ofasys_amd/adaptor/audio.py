"""AudioFbankAdaptor, source branch (reference: adaptor/audio.py:188-325; Conv2dSubsampling4, module/subsample.py:11-63):
fbank [B,T,80] -> Conv2d(1,D,3,2)+ReLU -> Conv2d(D,D,3,2)+ReLU -> Linear(D*19, D); learned positions; padding mask from
the subsampled lengths; 1-D log-bucket relative-position bias (bucket table [4096,4096], 2*max_position-1 rows).

The whole parameter set of the reference adaptor is mirrored (decoder prenet / postnet / feat_proj / eos_proj / mask_emb)
so checkpoints interchange; the target-side branch (fbank generation, TTS) is outside the train-step hot path and raises
NotImplementedError, as does is_transformer_layers (a branch the reference itself cannot construct).  The time / channel mask
DRAWS (get_mask_indices, :401-450) belong to the speech-pretraining criterion, which hands them over in the slot value."""
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from .. import ops
from ..configure import register_config
from ..module import Embedding
from ..preprocessor import Dictionary, ModalityType, Slot
from .base import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig
from .text import make_token_bucket_position

DEFAULT_MAX_WAV_POSITIONS = 4096


def make_audio_bucket_position(bucket_size, max_position=DEFAULT_MAX_WAV_POSITIONS):
    """adaptor/audio.py:50-60 -- the text adaptor's log-bucket rule on a 4096-position grid (integer, bit-exact)."""
    return make_token_bucket_position(bucket_size, max_position)


@dataclass
class AudioFbankAdaptorConfig(BaseAdaptorConfig):
    output_frame_dim: int = field(default=80, metadata={"help": "output_frame_dim"})
    n_frames_per_step: int = field(default=1, metadata={"help": "n_frames_per_step"})
    is_transformer_layers: bool = field(default=False, metadata={"help": "whether encoder prenet have transformer net"})
    prenet_layers: int = field(default=2, metadata={"help": "prenet layers"})
    prenet_dim: int = field(default=256, metadata={"help": "prenet dim"})
    prenet_dropout: float = field(default=0.5, metadata={"help": "prenet dropout"})
    postnet_conv_dim: int = field(default=512, metadata={"help": "postnet_conv_dim"})
    postnet_conv_kernel_size: int = field(default=5, metadata={"help": "postnet_conv_kernel_size"})
    postnet_layers: int = field(default=5, metadata={"help": "postnet_layers"})
    postnet_dropout: float = field(default=0.5, metadata={"help": "postnet_dropout"})
    use_mask: bool = field(default=False, metadata={"help": "use mask"})
    mask_prob: float = field(default=0.65, metadata={"help": "probability of replacing a token with mask"})
    mask_channel_prob: float = field(default=0.0, metadata={"help": "probability of replacing a feature with 0"})
    mask_channel_before: bool = False        # zero the drawn channels before (True) or after (False) the mask_emb rows are written


class Conv2dSubsampling4(nn.Module):
    """module/subsample.py:11-63 on the gfx950 convolution stack (im2col + MFMA GEMM with the bias in the epilogue)."""

    def __init__(self, idim: int, odim: int):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, odim, 3, 2), nn.ReLU(), nn.Conv2d(odim, odim, 3, 2), nn.ReLU())
        self.out = nn.Sequential(nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim))
        self.subsampling_rate = 4
        self.right_context = 6

    def get_out_seq_lens_tensor(self, in_seq_lens_tensor):
        out = in_seq_lens_tensor.clone()
        for _ in range(2):
            out = ((out.float() - 1) / 2 + 1).floor().long()
        return out

    def forward(self, x: torch.Tensor, x_length: torch.Tensor):
        B, T, Fd = x.shape
        c1, c2 = self.conv[0], self.conv[2]
        img = x.unsqueeze(1)                                                   # [B, 1, T, F] (NCHW, C = 1)
        h, T1, F1 = ops.conv2d(img, c1.weight, c1.bias, B, T, Fd, 2, 0, nchw=True)
        h = ops.relu(h)
        h, T2, F2 = ops.conv2d(h, c2.weight, c2.bias, B, T1, F1, 2, 0)
        h = ops.relu(h)
        C = h.shape[-1]
        # rows are (b, t, f) x c; the reference flattens each frame channel-major: view(b, t, c*f)  (:60-61)
        h = h.view(B, T2, F2, C).permute(0, 1, 3, 2).reshape(B, T2, C * F2)
        lin = self.out[0]
        return ops.linear(h, lin.weight, lin.bias), self.get_out_seq_lens_tensor(x_length)


class Prenet(nn.Module):                                                       # adaptor/audio.py:721-732 (decoder side)
    def __init__(self, in_dim, n_layers, n_units, dropout):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.Sequential(nn.Linear(in_dim if i == 0 else n_units, n_units), nn.ReLU()) for i in range(n_layers))
        self.dropout = dropout


class Postnet(nn.Module):                                                      # adaptor/audio.py:735-763 (decoder side)
    def __init__(self, in_dim, n_channels, kernel_size, n_layers, dropout):
        super().__init__()
        self.convolutions = nn.ModuleList()
        assert kernel_size % 2 == 1
        for i in range(n_layers):
            cur_layers = ([nn.Conv1d(in_dim if i == 0 else n_channels, n_channels if i < n_layers - 1 else in_dim,
                                     kernel_size=kernel_size, padding=((kernel_size - 1) // 2)),
                           nn.BatchNorm1d(n_channels if i < n_layers - 1 else in_dim)]
                          + ([nn.Tanh()] if i < n_layers - 1 else []) + [nn.Dropout(dropout)])
            nn.init.xavier_uniform_(cur_layers[0].weight, torch.nn.init.calculate_gain("tanh" if i < n_layers - 1 else "linear"))
            self.convolutions.append(nn.Sequential(*cur_layers))


@register_config("ofasys.adaptor", "audio_fbank", AudioFbankAdaptorConfig)
class AudioFbankAdaptor(BaseAdaptor):
    pos_batch_invariant = True          # positions are arange- / grid-derived: identical for every batch row

    def __init__(self, embed_tokens: Embedding, dictionary: Dictionary, is_src: bool, general_adaptor,
                 cfg: AudioFbankAdaptorConfig):
        super().__init__(embed_tokens, dictionary, is_src, general_adaptor, cfg)
        if cfg.is_transformer_layers:
            # adaptor/audio.py:208-217, 338-345 read cfg.encoder_config (no config class defines it) and import
            # ofasys.model.transformer_layer (no such module): the reference cannot construct this branch either
            raise NotImplementedError("audio_fbank.is_transformer_layers: the reference's own branch does not construct "
                                      "(cfg.encoder_config / ofasys.model.transformer_layer do not exist); default False")
        self.audio_bucket_size = cfg.max_position
        self.out_dim = cfg.output_frame_dim * cfg.n_frames_per_step
        self.subsample = Conv2dSubsampling4(self.out_dim, cfg.embed_dim)
        self.is_transformer_layers = False
        self.prenet = nn.Sequential(Prenet(self.out_dim, cfg.prenet_layers, cfg.prenet_dim, cfg.prenet_dropout),
                                    nn.Linear(cfg.prenet_dim, cfg.embed_dim))
        self.embed_audio_positions = Embedding(cfg.max_position, cfg.embed_dim)
        audio_num_rel_dis = 2 * self.audio_bucket_size - 1
        audio_rp_bucket = make_audio_bucket_position(self.audio_bucket_size)
        num_rel_pos_tables = 1 if self.cfg.share_attn_bias else self.num_layers
        self.audio_rel_pos_table_list = nn.ModuleList(
            [Embedding(audio_num_rel_dis, cfg.num_attention_heads, zero_init=True) for _ in range(num_rel_pos_tables)])
        self.register_buffer("audio_rp_bucket", audio_rp_bucket)
        self.n_frames_per_step = cfg.n_frames_per_step
        self.feat_proj = nn.Linear(cfg.embed_dim, self.out_dim)
        self.eos_proj = nn.Linear(cfg.embed_dim, 1)
        self.postnet = Postnet(self.out_dim, cfg.postnet_conv_dim, cfg.postnet_conv_kernel_size, cfg.postnet_layers,
                               cfg.postnet_dropout)
        self.use_mask = cfg.use_mask
        self.mask_emb = nn.Parameter(torch.FloatTensor(cfg.embed_dim).uniform_())
        self.mask_prob = cfg.mask_prob
        self.mask_channel_prob = cfg.mask_channel_prob
        self.mask_channel_before = cfg.mask_channel_before

    def get_rel_pos_bias(self, batch_size, seq_length, idx, **kwargs):
        if seq_length > self.audio_rp_bucket.size(0):                  # the reference fails on the size mismatch (slicing clamps)
            raise ValueError(f"sequence length {seq_length} exceeds the {self.audio_rp_bucket.size(0)} positions of audio_rp_bucket")
        rp_bucket = ops.cached_index(self, ("audio", seq_length), lambda: self.audio_rp_bucket[:seq_length, :seq_length].contiguous())
        return ops.embedding(rp_bucket, self.audio_rel_pos_table_list[idx].weight, plan_key=("audio", ops.owner_token(self)))

    def forward(self, slot: Slot, **kwargs) -> AdaptorOutput:
        assert slot.modality == ModalityType.AUDIO
        if not slot.is_src:
            raise NotImplementedError("target-side fbank slots (speech generation) are outside the train-step hot path")
        fbank = slot.value["fbank"]
        fbank_lengths = slot.value["fbank_lengths"]
        mask_indices = slot.value.get("mask_indices", None)
        feature, feature_length = self.subsample(fbank, fbank_lengths)
        T2 = feature.shape[1]
        # adaptor/audio.py:303-310 builds this row by row on the host (one device sync per row); same mask, no sync:
        # positions >= the subsampled length are padding
        padding_mask = torch.arange(T2, device=feature.device)[None, :] >= feature_length[:, None]
        pos = torch.arange(T2, device=feature.device)[None, :]
        pos_embed = self.embed_audio_positions(pos).expand(feature.shape[0], -1, -1)       # one lookup, batch-shared (ops.shared_rows)
        if (slot.has_attr("use_mask") or self.use_mask) and mask_indices is not None:      # apply_mask, :452-466
            mch = None
            if self.mask_channel_prob > 0:          # [B, C] channel draws of get_mask_indices: those channels are zeroed over all T
                if slot.value.get("mask_channel_indices") is None:
                    raise ValueError("audio_fbank.mask_channel_prob > 0: the slot value needs 'mask_channel_indices' ([B, C] bool, "
                                     "adaptor/audio.py:433-446) next to 'mask_indices'")
                mch = slot.value["mask_channel_indices"].to(feature.device).unsqueeze(1)
            if mch is not None and self.mask_channel_before:
                feature = feature.masked_fill(mch, 0.0)
            if self.mask_prob > 0:
                m = mask_indices.to(feature.device).unsqueeze(-1)
                feature = torch.where(m, self.mask_emb.to(feature.dtype).view(1, 1, -1), feature)
            if mch is not None and not self.mask_channel_before:
                feature = feature.masked_fill(mch, 0.0)
        return AdaptorOutput(feature, padding_mask, pos_embed, [])

    def forward_output(self, x, extra, slot, **kwargs):
        raise NotImplementedError("fbank output head (feat_proj / eos_proj / postnet) is outside the train-step hot path")
