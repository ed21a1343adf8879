"""ImagePatchEmbedAdaptor (reference: adaptor/image_patch_embed.py:14-80): non-overlapping p x p patch conv embed
(im2col + MFMA GEMM), learned [CLS] token, learned patch positions, all-False mask, no attention bias."""
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from .. import ops
from ..configure import register_config
from ..module import Embedding
from ..preprocessor import Dictionary, ModalityType, Slot
from .base import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig


@dataclass
class ImagePatchEmbedAdaptorConfig(BaseAdaptorConfig):
    image_size_width: int = 224
    image_size_height: int = 224
    patch_size_width: int = 14
    patch_size_height: int = 14
    embed_dim: int = 768
    add_cls_token: bool = True


@register_config("ofasys.adaptor", "image_patch_embed", ImagePatchEmbedAdaptorConfig)
class ImagePatchEmbedAdaptor(BaseAdaptor):
    pos_batch_invariant = True          # positions are arange- / grid-derived: identical for every batch row

    def __init__(self, embed_tokens: Embedding, dictionary: Dictionary, is_src: bool, general_adaptor,
                 cfg: ImagePatchEmbedAdaptorConfig):
        super().__init__(embed_tokens, dictionary, is_src, general_adaptor, cfg)
        image_size = (cfg.image_size_height, cfg.image_size_width)
        patch_size = (cfg.patch_size_height, cfg.patch_size_width)
        assert patch_size[0] == patch_size[1], "square patches only"
        num_patches = (image_size[1] // patch_size[1]) * (image_size[0] // patch_size[0])
        self.image_size, self.patch_size, self.num_patches = image_size, patch_size, num_patches
        self.embed_image_positions = Embedding(num_patches + 1 if cfg.add_cls_token else num_patches, cfg.embed_dim)
        if cfg.add_cls_token:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        # parameters named/shaped like nn.Conv2d(3, D, k=p, s=p) for checkpoint interchange
        self.proj = nn.Conv2d(3, cfg.embed_dim, kernel_size=patch_size, stride=patch_size)

    def _ofa_plain_conv_weights(self):
        """4-D weights the flat parameter arena must keep in torch's contiguous order (trainer.FlatParams._view): `proj.weight` is
        read by ops.PatchEmbedFn as a [D, C*p*p] view, not by the im2col convolution."""
        return (self.proj.weight,)

    def forward(self, slot: Slot, **kwargs) -> AdaptorOutput:
        assert slot.modality == ModalityType.IMAGE
        image: torch.Tensor = slot.value
        batch_size, _, height, width = image.shape
        assert height == self.image_size[0] and width == self.image_size[1], \
            f"Input image size ({height}*{width}) doesn't match model ({self.image_size[0]}*{self.image_size[1]})."
        # [B, (1 +) N, D]: the class token (:71-73 concatenates it) is written by the projection op itself
        x = ops.patch_embed(image, self.proj.weight, self.proj.bias, self.patch_size[0],
                            cls_token=self.cls_token if self.cfg.add_cls_token else None)
        n = x.size(1)
        mask = ops.cached_index(self, ("nomask", batch_size, n), lambda: torch.zeros((batch_size, n), dtype=torch.bool, device=image.device))
        pos = ops.cached_index(self, ("arange", n), lambda: torch.arange(n, dtype=torch.long, device=image.device).unsqueeze(0))
        # one lookup of the n positions, expanded (stride 0) to the batch: ops.shared_rows
        return AdaptorOutput(x, mask, self.embed_image_positions(pos).expand(batch_size, -1, -1), None)
