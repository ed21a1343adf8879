from .dictionary import Dictionary
from .instruction import ModalityType, Slot

__all__ = ["Dictionary", "ModalityType", "Slot"]
