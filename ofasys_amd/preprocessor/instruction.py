"""Carrier types of the hot path (reference: ofasys/__init__.py:29-45 ModalityType, preprocessor/instruction.py:29-107 Slot).
The Instruction parser / tokenizers / collators stay out of scope: fixtures and the
bench start at the Slot level, which is exactly the model's input."""
from dataclasses import dataclass
from enum import Enum
from typing import Any, List, Optional


class ModalityType(Enum):
    TEXT = 1
    IMAGE = 2
    BOX = 3
    AUDIO = 4
    MOTION = 5
    PHONE = 6
    VIDEO = 7
    STRUCT = 8
    CATEGORY = 9

    @classmethod
    def parse(cls, mark):
        for mod in ModalityType:
            if mark == mod.name:
                return cls(mod.value)
        return None


@dataclass
class Slot:
    """One modality span of an instruction: E-slot (is_src) or D-slot."""
    modality: ModalityType
    is_src: bool
    value: Optional[Any]

    global_position: Optional[int] = None
    column_name: Optional[str] = None
    attributes: Optional[List[str]] = None

    preprocess: Optional[str] = None
    is_plaintext: bool = False
    split: str = "train"
    decoder_plain_with_loss: bool = False

    def __post_init__(self):
        if self.column_name is None:
            self.column_name = str(self.global_position)
        if self.attributes is not None and isinstance(self.attributes, str):
            self.attributes = self.attributes.split(",")

    def has_attr(self, attr_key: str) -> bool:
        if self.attributes is None:
            return False
        return any(a == attr_key or a.startswith(attr_key + "=") for a in self.attributes)

    def get_attr(self, attr_key: str, class_factory: type = None):
        if self.attributes is None:
            return None
        for attr in self.attributes:
            if attr.startswith(attr_key + "="):
                val = attr[len(attr_key) + 1:]
                return class_factory(val) if class_factory is not None else val
        return None

    def attr2kwargs(self):
        kwargs = {}
        for attr in self.attributes or []:
            if "=" in attr:
                k, v = attr.split("=", 1)
            else:
                k, v = attr, True
            kwargs[k] = v
        return kwargs

    @staticmethod
    def get_target_slot_from_slots(slots: List):
        return [s for s in slots if not s.is_src][-1]

    @staticmethod
    def get_target_slot_from_sample(sample):
        return Slot.get_target_slot_from_slots(sample["net_input"]["slots"])
