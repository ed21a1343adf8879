"""ofasys_amd: MI355X-native (gfx950) implementation of the OFASys unified encoder-decoder hot path.

Public surface mirrors the reference's `ofasys/__init__.py:28-65` for the path in scope:
    from ofasys_amd import Task, Trainer, GeneralistModel, Instruction, Slot, ModalityType
Everything computes through libofasys_amd.so (include/ofasys_amd.h); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .preprocessor import Dictionary, Instruction, ModalityType, Slot  # noqa: E402,F401
from .configure import ConfigStore, register_config  # noqa: E402,F401
from .adaptor import AdaptorOutput, BaseAdaptor, BaseAdaptorConfig, OFAGeneralAdaptor  # noqa: E402,F401
from .model import GeneralistModel, GeneralistModelConfig  # noqa: E402,F401

from .task import Task, TaskConfig  # noqa: E402,F401
from .engine import Trainer, TrainerConfig  # noqa: E402,F401

__all__ = ["Task", "TaskConfig", "Trainer", "TrainerConfig", "Instruction",
           "GeneralistModel", "GeneralistModelConfig", "Slot", "ModalityType", "Dictionary", "register_config",
           "ConfigStore", "AdaptorOutput", "BaseAdaptor", "BaseAdaptorConfig", "OFAGeneralAdaptor"]
