"""Carrier types of the hot path (reference: ofasys/__init__.py:29-45 ModalityType, preprocessor/instruction.py:29-107 Slot)
and the instruction template (preprocessor/instruction.py:110-279 Instruction): "... [MOD:column,attr,...] ... -> ... [MOD:column]"
parsed into source (E-) and target (D-) slots; plain text between the bracketed slots becomes TEXT slots."""
import copy
import re
from collections import Counter
from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, List, Optional, Sequence, Union


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


# "[MOD]", "[MOD:column]", "[MOD:column,attr,key=value,...]"  (preprocessor/instruction.py:110-113)
_SLOT_RE = re.compile(r"\[(" + "|".join(m.name for m in ModalityType) + r")(?::([_A-Za-z0-9]+))?(?:,([_A-Za-z0-9,.=]+))?\]")

_HELP = ('an instruction reads "<source side> -> <target side>" with exactly one "->"; each side mixes plain text and slots written '
         '[MODALITY:column_name,attribute,key=value], MODALITY one of ' + ", ".join(m.name for m in ModalityType))


class Instruction:
    """Template of one task sample (preprocessor/instruction.py:116-279).

    Instruction("[IMAGE:img] what does the image describe? -> [TEXT:cap]") parses the template; `.format(*args, **kwargs)`
    returns a filled copy (positional values fill the empty slots in order, keyword values by column name, left-over keywords
    land in `.others`).  The slot-level form Instruction(slots, template, others) -- what collation consumes -- is accepted too.
    """

    def __init__(self, template: Union[str, Sequence[Slot]], split: Union[str, None] = "train",
                 decoder_plain_with_loss: Union[bool, Dict, None] = False, *, slots: Optional[List[Slot]] = None,
                 others: Optional[Dict[str, Any]] = None):
        if not isinstance(template, str):                    # slot-level construction: Instruction(slots, template="", others={})
            self.slots = list(template)
            self.template = split if isinstance(split, str) and split not in ("train", "valid", "test") else ""
            self.split = "train"
            self.decoder_plain_with_loss = False
            self.others = dict(decoder_plain_with_loss) if isinstance(decoder_plain_with_loss, dict) else dict(others or {})
            return
        template = template.strip()
        if template.count("->") != 1:
            raise ValueError(_HELP)
        source, target = (part.strip() for part in template.split("->"))
        self.template, self.split = template, split
        self.decoder_plain_with_loss = bool(decoder_plain_with_loss)
        self.slots: List[Slot] = list(slots) if slots is not None else []
        if slots is None:
            self._parse_side(source, True)
            self._parse_side(target, False)
        self.others = dict(others or {})

    # ------------------------------------------------------------------ parsing
    def _plain(self, text, is_src):
        text = text.strip()
        if text:
            self.slots.append(Slot(ModalityType.TEXT, is_src, text, global_position=len(self.slots), is_plaintext=True,
                                   split=self.split, decoder_plain_with_loss=self.decoder_plain_with_loss))

    def _parse_side(self, side: str, is_src: bool):
        cursor = 0
        for m in _SLOT_RE.finditer(side):
            self._plain(side[cursor:m.start()], is_src)
            mod, column, attrs = m.groups()
            self.slots.append(Slot(ModalityType.parse(mod), is_src, None, global_position=len(self.slots), column_name=column,
                                   attributes=attrs, is_plaintext=False, split=self.split,
                                   decoder_plain_with_loss=self.decoder_plain_with_loss))
            cursor = m.end()
        self._plain(side[cursor:], is_src)

    # ------------------------------------------------------------------ use
    def __str__(self):
        words, on_source = [], True
        for slot in self.slots:
            if on_source and not slot.is_src:
                words.append("->")
                on_source = False
            words.append(str(slot.value))
        return " ".join(words)

    def get_slot_names(self) -> List[str]:
        return [slot.column_name for slot in self.slots if slot.value is None]

    def format(self, *args, **kwargs):
        """A filled deep copy.  A column used by several slots is filled everywhere from one value; a missing SOURCE value is
        an error, a missing target value stays None (inference)."""
        ist = copy.deepcopy(self)
        args = list(args)
        remaining = Counter(s.column_name for s in ist.slots if not s.is_plaintext)
        for slot in ist.slots:
            if slot.value is not None:
                continue
            if args:
                slot.value = args.pop(0)
                remaining[slot.column_name] -= 1
                if remaining[slot.column_name] != 0:          # later slots of the same column reuse this value
                    kwargs[slot.column_name] = slot.value
            else:
                slot.value = kwargs.get(slot.column_name)
                if slot.value is None and slot.is_src:
                    raise ValueError(f"Expect filling slot ({slot.column_name}) but missing")
                remaining[slot.column_name] -= 1
        if args:
            raise ValueError(f"Unexpect args ({args})")
        ist.others = kwargs
        return ist
