"""`Task`: one instruction + its datasets + preprocessing + criterion settings (reference: task/base.py:157-460, 586-615,
839-865; exported as `ofasys.Task`, ofasys/__init__.py:65).

    task = Task(name="caption", instruction="[IMAGE:image] what does the image describe? -> [TEXT:caption]", micro_batch_size=4)
    task.add_dataset(rows, "train")            # any sequence of dict rows: datasets.Dataset, list of dicts, ...

What stays out: file / OSS readers, multi-worker prefetch, metrics, generators, checkpoint state.  The
batch iterator is a plain in-process loop over the bound dataset: micro-batches of `micro_batch_size` rows, each rank of a
data-parallel job reading its own contiguous shard (io/reader/dataset.py:49-53), reshuffled per epoch with a seeded permutation.
"""
import random
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Set, Union

import numpy as np
import torch

from .preprocessor.collate import (BoxPreprocessConfig, DefaultBoxPreprocess, DefaultTextPreprocess, GeneralPreprocess,
                                   PreprocessConfig, TensorPreprocess, TextPreprocessConfig, default_preprocess)
from .preprocessor.instruction import Instruction, ModalityType, Slot

# adaptor/general.py:36-46
default_adaptor = {
    ModalityType.TEXT: "text", ModalityType.IMAGE: "image_resnet", ModalityType.BOX: "text", ModalityType.AUDIO: "audio_fbank",
    ModalityType.PHONE: "text", ModalityType.VIDEO: "video_image_sequence", ModalityType.MOTION: "text",
    ModalityType.STRUCT: "text", ModalityType.CATEGORY: "text",
}


@dataclass
class DatasetConfig:                                  # task/base.py:48-120 (fields the in-process iterator uses)
    micro_batch_size: int = 1
    update_freq: int = 1
    shuffle: bool = True
    seed: int = 1


@dataclass
class InstructionConfig:                              # task/base.py:123-137
    template: Optional[str] = None
    decoder_plain_with_loss: bool = False


@dataclass
class CriterionConfig:                                # engine/criterion/label_smoothed_cross_entropy.py:27-59 / cross_entropy.py
    label_smoothing: float = 0.0
    drop_worst_ratio: float = 0.0


@dataclass
class ImagePreprocessConfig(PreprocessConfig):        # preprocessor/default/image.py (resize + mean/std 0.5 normalisation)
    patch_image_size: int = 224
    mean: float = 0.5
    std: float = 0.5


@dataclass
class TaskConfig:                                     # task/base.py:157-189
    dataset: DatasetConfig = field(default_factory=DatasetConfig)
    instruction: InstructionConfig = field(default_factory=InstructionConfig)
    criterion: CriterionConfig = field(default_factory=CriterionConfig)
    text: TextPreprocessConfig = field(default_factory=TextPreprocessConfig)
    box: BoxPreprocessConfig = field(default_factory=BoxPreprocessConfig)
    image: ImagePreprocessConfig = field(default_factory=ImagePreprocessConfig)
    max_src_length: int = 128
    max_tgt_length: int = 30
    constraint_range: Optional[str] = None
    _name: Optional[str] = None

    def update(self, **kwargs):
        if "name" in kwargs:
            self._name = kwargs["name"]
        if "instruction" in kwargs:
            self.instruction.template = kwargs["instruction"]
        if "micro_batch_size" in kwargs:
            self.dataset.micro_batch_size = kwargs["micro_batch_size"]


def parse_template(template: Union[None, str, List[str]]) -> Optional[List[str]]:
    """A task may carry several alternative templates: a list, or one string with '||' between them."""
    if template is None:
        return None
    if isinstance(template, str):
        return [t.strip() for t in template.split("||") if t.strip()]
    return list(template)


class ImageTensorPreprocess(TensorPreprocess):
    """IMAGE / VIDEO columns that are already arrays: CHW float tensors pass through; HWC uint8 arrays (or PIL images) are
    scaled to [0,1], resized (bilinear) to patch_image_size and normalised with mean = std = 0.5.  File / URL decoding is out
    of scope."""

    def map(self, slot: Slot) -> Slot:
        v = slot.value
        if hasattr(v, "convert") and not torch.is_tensor(v):          # PIL.Image
            v = np.asarray(v.convert("RGB"))
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if not torch.is_tensor(v):
            raise ValueError(f"image slot '{slot.column_name}': expected a tensor / array / PIL image, got {type(v)} "
                             "(decoding files or URLs is outside this package)")
        if v.dtype == torch.uint8:
            if v.dim() == 3 and v.shape[-1] in (1, 3):
                v = v.permute(2, 0, 1)
            v = v.float() / 255.0
            size = self.cfg.patch_image_size
            if v.shape[-2:] != (size, size):
                v = torch.nn.functional.interpolate(v[None], size=(size, size), mode="bilinear", align_corners=False)[0]
            v = (v - self.cfg.mean) / self.cfg.std
        slot.value = v.float()
        return slot


class Task:
    def __init__(self, cfg: TaskConfig = None, **kwargs):
        self.cfg = TaskConfig() if cfg is None else cfg
        self.cfg.update(**kwargs)
        self.datasets: Dict[str, Any] = {}
        self.templates = parse_template(self.cfg.instruction.template)
        self.target_modality = self.infer_target_modality(self.templates[0]) if self.templates else None
        self.global_dict = None
        self.general_preprocess: Optional[GeneralPreprocess] = None
        self._iters = {}

    # ------------------------------------------------------------------ identity / datasets (task/base.py:256-273)
    @property
    def name(self):
        return self.cfg._name or self.__class__.__name__

    def add_dataset(self, dataset, split="train"):
        assert self.datasets.get(split) is None, f"{split} dataset already exists in task {self.name}"
        self.datasets[split] = dataset

    def add_train_dataset(self, dataset):
        self.add_dataset(dataset, "train")

    def add_valid_dataset(self, dataset):
        self.add_dataset(dataset, "valid")

    def add_test_dataset(self, dataset):
        self.add_dataset(dataset, "test")

    def infer_target_modality(self, instruction: Union[str, Instruction]):
        if not isinstance(instruction, Instruction):
            instruction = Instruction(instruction)
        return Slot.get_target_slot_from_slots(instruction.slots).modality

    # ------------------------------------------------------------------ set-up (task/base.py:218-231, 328-338)
    def initialize(self, global_dict, **kwargs):
        """Registers this task's symbols in the shared dictionary ('<text>_i', '<mask>', '<bin>_k') and builds its preprocessors."""
        self.global_dict = global_dict
        if "<text>_0" not in global_dict:
            from .preprocessor.tokenizer import N_GPT2
            for i in range(N_GPT2):
                global_dict.add_symbol(f"<text>_{i}")
        global_dict.add_symbol("<mask>")
        self.cfg.text.max_src_length, self.cfg.text.max_tgt_length = self.cfg.max_src_length, self.cfg.max_tgt_length
        name2pre = {"text": DefaultTextPreprocess(global_dict, self.cfg.text), "box": DefaultBoxPreprocess(global_dict, self.cfg.box)}
        for name, mod in (("image", ModalityType.IMAGE), ("video", ModalityType.VIDEO)):
            name2pre[name] = ImageTensorPreprocess(global_dict, self.cfg.image, mod)
        name2pre["audio"] = TensorPreprocess(global_dict, PreprocessConfig(), ModalityType.AUDIO)
        name2pre["table"] = name2pre["phone"] = name2pre["text"]     # STRUCT / PHONE columns arrive as token ids (text adaptor)
        self.general_preprocess = GeneralPreprocess(global_dict, name2pre)

    @classmethod
    def upgrade_model_adaptor_cfg(cls, tasks, model_cfg):
        """Activate exactly the adaptors the tasks' instructions need (task/base.py:228-231, 839-865)."""
        for name in collect_adaptor_name_from_tasks(tasks):
            getattr(model_cfg.adaptor, name).is_active = True

    # ------------------------------------------------------------------ rows -> samples (task/base.py:291-326, 389-394)
    def preprocess(self, data: Dict[str, Any], split: str) -> Dict[str, Any]:
        return data

    def build_instruction(self, data: Dict[str, Any], split: str) -> Instruction:
        template = random.choice(self.templates) if len(self.templates) > 1 else self.templates[0]
        return Instruction(template, split=split, decoder_plain_with_loss=self.cfg.instruction.decoder_plain_with_loss).format(**data)

    def preprocess_data_and_instruction(self, data, split):
        data = self.preprocess(data, split)
        if data is None:
            return None
        return self.general_preprocess(self.build_instruction(data, split))

    def collate(self, rows: List[Dict[str, Any]], split="train") -> Dict:
        samples = [s for s in (self.preprocess_data_and_instruction(dict(r), split) for r in rows) if s is not None]
        return self.general_preprocess.collate(samples)

    # ------------------------------------------------------------------ iteration (task/base.py:396-449, in process)
    def _batches(self, split, rank, world):
        ds = self.datasets[split]
        n = len(ds)
        per = n // world if n >= world else n                       # contiguous per-rank shard (io/reader/dataset.py:49-53)
        lo = (rank * per) % max(n, 1)
        bs = self.cfg.dataset.micro_batch_size
        epoch = 0
        while True:
            order = list(range(lo, lo + per))
            if split == "train" and self.cfg.dataset.shuffle:
                np.random.default_rng(self.cfg.dataset.seed + epoch).shuffle(order)
            for i in range(0, len(order) - bs + 1 if len(order) >= bs else 1, bs):
                idx = order[i:i + bs]
                yield self.collate([ds[int(j) % n] for j in idx], split)
            epoch += 1

    def init_data_iterator(self, split="train", rank=0, world=1):
        self._iters[split] = self._batches(split, rank, world)

    def get_sample(self, split="train"):
        if split not in self._iters:
            self.init_data_iterator(split)
        return next(self._iters[split])


def collect_adaptor_name_from_tasks(tasks) -> Set[str]:
    names = set()
    for task in tasks:
        for template in task.templates:
            for slot in Instruction(template).slots:
                names.add(slot.get_attr("adaptor") if slot.has_attr("adaptor") else default_adaptor[slot.modality])
    return names
