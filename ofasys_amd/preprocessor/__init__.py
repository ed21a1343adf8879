from .collate import (CollateOutput, DefaultBoxPreprocess, DefaultTextPreprocess, GeneralPreprocess, Instruction,
                      TensorPreprocess, collate_others, collate_tokens, group_by_predicator, to_device)
from .dictionary import Dictionary
from .instruction import ModalityType, Slot

__all__ = ["Dictionary", "ModalityType", "Slot", "Instruction", "CollateOutput", "GeneralPreprocess", "DefaultTextPreprocess",
           "DefaultBoxPreprocess", "TensorPreprocess", "collate_tokens", "collate_others", "group_by_predicator", "to_device"]
