"""ImageResnetAdaptor (reference: adaptor/image_resnet.py:25-202) -- the default IMAGE adaptor: ResNet-{50,101,152}
stride-16 features -> Linear(1024, D); 2-D position ids `w + h*bucket + 1`; 2-D relative-position bias by a double gather
into the integer table `image_rp_bucket` and one Embedding table per layer."""
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..configure import ChoiceEnum, register_config
from ..module import Embedding, Linear
from ..module.resnet import resnet50_backbone, resnet101_backbone, resnet152_backbone
from ..preprocessor import Dictionary, ModalityType, Slot
from .base import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig


def make_image_bucket_position(bucket_size, num_relative_distance):
    """The reference's [bucket_size^2 + 1, bucket_size^2 + 1] int64 table of 2-D relative-position ids (adaptor/image_resnet.py:25-40; a
    state-dict buffer, pinned by the CRC in tests/golden/base_resnet*.npz).  Written from its meaning: grid cell p = (y, x) has id
    1 + y * bucket_size + x; the entry for a pair of cells is the row-major index of their offset (dy, dx) in the
    (2 * bucket_size - 1)^2 window of possible offsets; id 0 is the reserved class-token slot, whose row, column and corner take the
    last three ids of the table."""
    span = 2 * bucket_size - 1
    ys, xs = np.divmod(np.arange(bucket_size * bucket_size, dtype=np.int64), bucket_size)
    dy = ys[:, None] - ys[None, :] + (bucket_size - 1)
    dx = xs[:, None] - xs[None, :] + (bucket_size - 1)
    table = np.empty((bucket_size * bucket_size + 1,) * 2, dtype=np.int64)
    table[1:, 1:] = dy * span + dx
    table[0, :] = num_relative_distance - 3
    table[:, 0] = num_relative_distance - 2
    table[0, 0] = num_relative_distance - 1
    return torch.from_numpy(table)


@dataclass
class ImageResnetAdaptorConfig(BaseAdaptorConfig):
    resnet_type: ChoiceEnum(["resnet50", "resnet101", "resnet152"]) = field(default="resnet152", metadata={"help": "resnet type"})
    resnet_drop_path_rate: float = field(default=0.0, metadata={"help": "resnet drop path rate"})
    sync_bn: bool = field(default=False, metadata={"help": "sync batchnorm"})
    freeze_resnet: bool = field(default=False, metadata={"help": "freeze resnet"})
    image_bucket_size: int = field(default=42, metadata={"help": "image bucket size"})
    pretrained_ckpt_path: str = field(default="", metadata={"help": "path of pretrained ckpt"})


def _sync_batch_norm_2d(out_chan, momentum=0.1, eps=1e-3):
    """module/layer.py:26-27 SynBatchNorm2d."""
    bn = nn.BatchNorm2d(out_chan, momentum=momentum, eps=eps)
    bn._ofa_sync = True
    return bn


@register_config("ofasys.adaptor", "image_resnet", ImageResnetAdaptorConfig)
class ImageResnetAdaptor(BaseAdaptor):
    pos_batch_invariant = True          # positions are arange- / grid-derived: identical for every batch row

    def __init__(self, embed_tokens: Embedding, dictionary: Dictionary, is_src: bool, general_adaptor,
                 cfg: ImageResnetAdaptorConfig):
        super().__init__(embed_tokens, dictionary, is_src, general_adaptor, cfg)
        if cfg.pretrained_ckpt_path:
            raise NotImplementedError("pretrained_ckpt_path: load the state dict through model.load_state_dict instead")
        self.embed_image_positions = Embedding(cfg.image_bucket_size ** 2 + 1, cfg.embed_dim)
        backbone = {"resnet50": resnet50_backbone, "resnet101": resnet101_backbone, "resnet152": resnet152_backbone}[cfg.resnet_type]
        # image_resnet.py:87-90: norm_layer = SynBatchNorm2d when cfg.sync_bn (module/layer.py:26-27: nn.SyncBatchNorm converted from
        # nn.BatchNorm2d(momentum=0.1, eps=1e-3) -- note the eps, 100x the plain layer's -- same parameters, buffers and state-dict
        # keys).  Here the modules stay nn.BatchNorm2d holders carrying a flag: in training ops.batch_norm all-reduces each layer's
        # per-channel sums over the default process group (ops.SyncBatchNormFn)
        self.embed_images = backbone(norm_layer=_sync_batch_norm_2d if cfg.sync_bn else None, drop_path_rate=cfg.resnet_drop_path_rate)
        self.image_proj = Linear(1024, cfg.embed_dim)
        image_num_rel_dis = (2 * cfg.image_bucket_size - 1) * (2 * cfg.image_bucket_size - 1) + 3
        image_rp_bucket = make_image_bucket_position(cfg.image_bucket_size, image_num_rel_dis)
        num_rel_pos_tables = 1 if self.cfg.share_attn_bias else self.num_layers
        self.image_rel_pos_table_list = nn.ModuleList(
            [Embedding(image_num_rel_dis, cfg.num_attention_heads, zero_init=True) for _ in range(num_rel_pos_tables)])
        self.register_buffer("image_rp_bucket", image_rp_bucket)

    def train(self, mode=True):                                               # image_resnet.py:107-114
        super().train(mode)
        if self.cfg.freeze_resnet:
            for m in self.embed_images.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
                    m.weight.requires_grad = False
                    m.bias.requires_grad = False
        return self

    def get_rel_pos_bias(self, batch_size, seq_length, idx, **kwargs):
        """[T,T,A] values = table_idx[ bucket[ids][:, ids] ]  (image_resnet.py:116-128).  The reference gathers per batch
        row and returns [B,A,T,T]; every row of `image_position_ids` is the same arange-derived vector, so the values are
        computed once and handed to the bias assembly as the usual batch-expanded view."""
        ids = kwargs["image_position_ids"]
        hw = getattr(self, "_hw", None)                                        # integer double gather, bit-exact; the same for every layer
        gather = lambda: self.image_rp_bucket[ids][:, ids].contiguous()        # noqa: E731   (and every step: ids come from (h, w))
        rp_bucket = ops.cached_index(self, ("image", hw, int(ids.numel())), gather) if hw is not None else gather()
        # (the position ids are arange-derived from the feature map's (h, w): with the shape, that identifies the lookup)
        return ops.embedding(rp_bucket, self.image_rel_pos_table_list[idx].weight, plan_key=("image", ops.owner_token(self), getattr(self, "_hw", None)))

    def get_patch_images_info(self, patch_images):
        """image_resnet.py:130-164 -> (embed rows [B, h*w, 1024], n, mask, position ids [T], pos_embed [B, T, D])."""
        device = patch_images.device
        B = patch_images.size(0)
        rows, h, w = self.embed_images(patch_images)
        self._hw = (int(h), int(w))
        n = h * w
        image_embed = rows.view(B, n, rows.shape[-1])
        image_padding_mask = torch.zeros((B, n), dtype=torch.bool, device=device)
        idx = (torch.arange(w, device=device).unsqueeze(0).expand(h, w)
               + torch.arange(h, device=device).unsqueeze(1) * self.cfg.image_bucket_size + 1).view(-1)
        image_pos_embed = self.embed_image_positions(idx[None, :]).expand(B, -1, -1)        # one lookup, batch-shared (ops.shared_rows)
        return image_embed, n, image_padding_mask, idx, image_pos_embed

    def forward(self, slot: Slot, **kwargs) -> AdaptorOutput:
        assert slot.modality == ModalityType.IMAGE
        image_embed, n, mask, position_ids, pos_embed = self.get_patch_images_info(slot.value)
        image_embed = self.image_proj(image_embed)
        batch_size, seq_length = image_embed.size()[:2]
        self_attn_bias = []
        if self.cfg.use_self_attn_bias:
            num_rel_pos_tables = 1 if self.cfg.share_attn_bias else self.num_layers
            for idx in range(num_rel_pos_tables):
                values = self.get_rel_pos_bias(batch_size, seq_length, idx, image_position_ids=position_ids)
                self_attn_bias.append(self.expand_rel_pos_bias(values, batch_size))
        return AdaptorOutput(image_embed, mask, pos_embed, self_attn_bias)
