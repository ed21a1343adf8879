"""VideoImageSequenceAdaptor (reference: adaptor/video_image_sequence.py:43-221): every frame of a clip goes through the
image_resnet adaptor's backbone and projection; position = image position + learned frame position; a frame whose mean
|x| is exactly 0 is padding; the attention bias is frame rel-pos (1-D log buckets) (+) image rel-pos (2-D), i.e.
bias[(f,p),(f',p')] = frame[f,f'] + image[p,p'].

The reference materialises that bias as [B, A, F*P, F*P] per layer (1.97 GB per layer at B=32, 8 frames of 196 patches);
it does not depend on the batch row, so here it is built once as [F*P, F*P, A] values and handed to the bias assembly as
the usual batch-expanded view."""
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from .. import ops
from ..configure import ConfigStore, register_config
from ..module import Embedding
from ..preprocessor import Dictionary, ModalityType, Slot
from .base import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig
from .text import make_token_bucket_position


@dataclass
class VideoImageSequenceAdaptorConfig(BaseAdaptorConfig):
    token_bucket_size: int = field(default=256, metadata={"help": "token bucket size"})


def make_video_bucket_position(bucket_size, max_position=8192):
    """video_image_sequence.py:51-61 -- the same log-bucket rule as the text adaptor's table."""
    return make_token_bucket_position(bucket_size, max_position)


@register_config("ofasys.adaptor", "video_image_sequence", VideoImageSequenceAdaptorConfig)
class VideoImageSequenceAdaptor(BaseAdaptor):
    pos_batch_invariant = True          # positions are arange- / grid-derived: identical for every batch row

    def __init__(self, embed_tokens: Embedding, dictionary: Dictionary, is_src: bool, general_adaptor,
                 cfg: VideoImageSequenceAdaptorConfig):
        super().__init__(embed_tokens, dictionary, is_src, general_adaptor, cfg)
        self.embed_frame_positions = Embedding(1024 + 1, cfg.embed_dim, zero_init=True)
        video_num_rel_dis = 2 * cfg.token_bucket_size - 1
        video_rp_bucket = make_video_bucket_position(cfg.token_bucket_size, 1024)
        num_rel_pos_tables = 1 if self.cfg.share_attn_bias else self.num_layers
        self.video_rel_pos_table_list = nn.ModuleList(
            [Embedding(video_num_rel_dis, cfg.num_attention_heads, zero_init=True) for _ in range(num_rel_pos_tables)])
        self.register_buffer("video_rp_bucket", video_rp_bucket)
        if "image_resnet" not in self.general_adaptor.name2adaptor and self.is_src:      # :84-96
            ga = self.general_adaptor
            ga.name2adaptor["image_resnet"] = ConfigStore().get("ofasys.adaptor", "image_resnet").target(
                embed_tokens, dictionary, is_src, ga, getattr(ga.cfg.adaptor, "image_resnet"))
            setattr(ga, "image_resnet", ga.name2adaptor["image_resnet"])

    def get_image_resnet_adaptor(self):
        a = self.general_adaptor.name2adaptor["image_resnet"]
        assert a is not None
        return a

    def get_rel_pos_bias(self, batch_size, seq_length, idx, **kwargs):
        if seq_length > self.video_rp_bucket.size(0):                  # the reference fails on the size mismatch (slicing clamps)
            raise ValueError(f"sequence length {seq_length} exceeds the {self.video_rp_bucket.size(0)} positions of video_rp_bucket")
        rp_bucket = ops.cached_index(self, ("video", seq_length), lambda: self.video_rp_bucket[:seq_length, :seq_length].contiguous())
        return ops.embedding(rp_bucket, self.video_rel_pos_table_list[idx].weight, plan_key=("video", ops.owner_token(self)))        # [F,F,A]

    def get_clip_videos_info(self, clip_videos: torch.Tensor):
        """video_image_sequence.py:111-154.  clip_videos: [B, 3, F, H, W]."""
        ira = self.get_image_resnet_adaptor()
        device = clip_videos.device
        clips = clip_videos.transpose(1, 2)                                               # [B, F, 3, H, W]
        B, Fr = clips.size(0), clips.size(1)
        rows, h, w = ira.embed_images(clips.reshape(-1, clips.size(2), clips.size(3), clips.size(4)))
        ira._hw = (int(h), int(w))
        P = h * w
        T = P * Fr
        video_embed = rows.view(B, T, rows.shape[-1])                                     # rows are (b, f, h, w) ordered
        pad = clips.reshape(B, Fr, -1).abs().mean(dim=-1) == 0.0                          # :131-133 (frame padding rule)
        video_padding_mask = pad.unsqueeze(-1).expand(B, Fr, P).reshape(B, T)
        image_position_idx = (torch.arange(w, device=device).unsqueeze(0).expand(h, w)
                              + torch.arange(h, device=device).unsqueeze(1) * ira.cfg.image_bucket_size + 1).view(-1)
        frame_position_idx = torch.arange(Fr, device=device) + 1
        # positions are the same for every clip: [1, ...] lookups, the sum built once and expanded (stride 0) to the batch
        image_pos_embed = ira.embed_image_positions(image_position_idx[None, :])
        frame_pos_embed = self.embed_frame_positions(frame_position_idx[None, :])
        video_pos_embed = (image_pos_embed.unsqueeze(1) + frame_pos_embed.unsqueeze(2)).reshape(1, T, -1).expand(B, -1, -1)
        return video_embed, T, video_padding_mask, image_position_idx, video_pos_embed

    def forward(self, slot: Slot, **kwargs) -> AdaptorOutput:
        assert slot.modality == ModalityType.VIDEO
        ira = self.get_image_resnet_adaptor()
        video_embed, T, mask, image_position_idx, pos_embed = self.get_clip_videos_info(slot.value)
        video_embed = ira.image_proj(video_embed)
        batch_size, seq_length = video_embed.size()[:2]
        P = image_position_idx.size(-1)
        Fr = seq_length // P
        self_attn_bias = []
        if self.cfg.use_self_attn_bias:
            for idx in range(self.num_layers):
                vi = ira.get_rel_pos_bias(batch_size, P, idx, image_position_ids=image_position_idx)   # [P,P,A]
                vf = self.get_rel_pos_bias(batch_size, Fr, idx)                                        # [F,F,A]
                # :187-204 adds the two broadcast views into [F P, F P, A] values (59 MB per layer at 8 x 196 positions) and expands them
                # over the batch; here the two tables travel on as they are (ops.OuterRelPos): the bias assembly reads them, the
                # gradient is summed straight into them
                self_attn_bias.append(ops.LazyRelPosBias(ops.OuterRelPos(vf, vi), batch_size))
        else:
            self_attn_bias = [None] * self.num_layers
        return AdaptorOutput(video_embed, mask, pos_embed, self_attn_bias)

    def upgrade_state_dict_named(self, state_dict, name):
        if name == "encoder.adaptor.video_image_sequence":                                # :209-221
            resnet_prefix = name.replace("video_image_sequence", "image_resnet")
            for key in ("layernorm_embedding.weight", "layernorm_embedding.bias", "layernorm_position.weight",
                        "layernorm_position.bias", "type_embedding.weight"):
                full_key = f"{name}.{key}"
                if full_key not in state_dict:
                    state_dict[full_key] = state_dict[f"{resnet_prefix}.{key}"].clone()
