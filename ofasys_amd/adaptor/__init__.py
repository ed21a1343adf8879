from .base import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig
from . import text, image_patch_embed, image_resnet, video_image_sequence, audio  # noqa: F401  (registers the adaptors)
from .general import OFAAdaptorConfig, OFAGeneralAdaptor, default_adaptor

__all__ = ["AdaptorOutput", "BaseAdaptor", "BaseAdaptorConfig", "OFAAdaptorConfig", "OFAGeneralAdaptor", "default_adaptor"]
