"""CPU: the user-facing surface of scripts/trainer_api.py -- Task(name, instruction, micro_batch_size), add_dataset, the
instruction -> slots -> collated micro-batch path, adaptor activation from the instructions, the lr schedule -- on synthetic
in-memory datasets (no GPU work: the step itself is covered by tests/test_trainer_api_gpu.py)."""
import numpy as np
import pytest
import torch

from ofasys_amd import Dictionary, GeneralistModel, ModalityType, Task, Trainer
from ofasys_amd.engine import polynomial_decay_lr
from ofasys_amd.task import collect_adaptor_name_from_tasks


def caption_rows(n):
    g = torch.Generator().manual_seed(0)
    return [{"image_url": torch.randn(3, 224, 224, generator=g), "caption": f"a synthetic caption number {i} of a card"} for i in range(n)]


def cola_rows(n):
    return [{"sentence": f"the {i}th book was written by a very careful author .", "label": i % 2, "idx": i} for i in range(n)]


def make_tasks():
    task1 = Task(name="caption", instruction="[IMAGE:image_url] what does the image describe? -> [TEXT:caption]", micro_batch_size=4)
    task2 = Task(name="text_infilling",
                 instruction='what is the complete text of " [TEXT:sentence,mask_ratio=0.3] "? -> [TEXT:sentence]', micro_batch_size=2)
    task1.add_dataset(caption_rows(9), "train")
    task2.add_dataset(cola_rows(7), "train")
    return task1, task2


def test_task_surface_and_batches():
    task1, task2 = make_tasks()
    assert task1.name == "caption" and task1.cfg.dataset.micro_batch_size == 4 and task1.target_modality == ModalityType.TEXT
    with pytest.raises(AssertionError):
        task1.add_dataset([], "train")                           # a split is bound once (task/base.py:262-264)
    assert collect_adaptor_name_from_tasks([task1, task2]) == {"image_resnet", "text"}
    d = Dictionary()
    for t in (task1, task2):
        t.initialize(d)
    V = len(d)
    assert d.index("<mask>") == V - 1001 or "<bin>_0" in d       # <text>_i, <mask>, then the box bins
    s1 = task1.get_sample("train")
    img, src = s1["net_input"]["slots"][0], s1["net_input"]["slots"][1]
    assert img.modality == ModalityType.IMAGE and img.value.shape == (4, 3, 224, 224)
    assert src.modality == ModalityType.TEXT and src.is_src and src.value.shape[0] == 4
    assert (src.value[:, 0] == d.bos()).all() and (src.value == d.eos()).sum() == 4      # bos ... eos around the grouped plain text
    prev = s1["net_input"]["slots"][2]
    assert not prev.is_src and prev.value.shape == s1["target"].shape
    # teacher forcing: target = prev shifted left (text.py:278-313)
    for r in range(4):
        n = int(s1["target"][r].ne(d.pad()).sum())
        assert torch.equal(prev.value[r, 1:n], s1["target"][r, :n - 1]) and s1["target"][r, n - 1] == d.eos()
    assert s1["ntokens"] == int(s1["target"].ne(d.pad()).sum()) and s1["nsentences"] == 4
    s2 = task2.get_sample("train")
    src2 = s2["net_input"]["slots"][0].value
    assert src2.shape[0] == 2 and (src2 == d.index("<mask>")).sum() > 0                 # mask_ratio=0.3 noised the source copy
    assert (s2["target"] == d.index("<mask>")).sum() == 0                                 # ... but not the target
    assert len(s2["label"]) == 2 and set(np.asarray(s2["label"]).tolist()) <= {0, 1}      # unused columns ride along (`others`)
    # a second epoch keeps producing batches; valid split without noise
    for _ in range(8):
        task2.get_sample("train")
    task2.add_dataset(cola_rows(4), "valid")
    sv = task2.get_sample("valid")
    assert (sv["net_input"]["slots"][0].value == d.index("<mask>")).sum() == 0


def test_adaptor_activation_and_model_build():
    task1, task2 = make_tasks()
    model = GeneralistModel()
    d = Dictionary()
    for t in (task1, task2):
        t.initialize(d)
    Task.upgrade_model_adaptor_cfg([task1, task2], model.cfg)
    assert model.cfg.adaptor.image_resnet.is_active and model.cfg.adaptor.text.is_active
    assert not model.cfg.adaptor.audio_fbank.is_active
    t3 = Task(name="patch", instruction="[IMAGE:img,adaptor=image_patch_embed] -> [TEXT:cap]", micro_batch_size=1)
    assert collect_adaptor_name_from_tasks([t3]) == {"image_patch_embed", "text"}


def test_trainer_config_and_schedule():
    tr = Trainer(max_update=100, lr=2e-4, clip_norm=0.5, fp32=True, weight_decay=0.0)
    assert tr.cfg.optimization.max_update == 100 and tr.cfg.optimization.lr == [2e-4] and tr.cfg.common.fp32
    with pytest.raises(TypeError):
        Trainer(no_such_option=1)
    # the learning rate of every update, recorded from the reference's own scheduler driven as its trainer drives it
    # (oracle/gen_lr_golden.py -> tests/golden/lr_schedule.json)
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_schedule.json")))
    for name, c in gold.items():
        got = [polynomial_decay_lr(done, c["lr"], c["total"], c["warmup_ratio"], c["end"], c["power"]) for done in range(len(c["lrs"]))]
        assert got == pytest.approx(c["lrs"], rel=1e-12, abs=1e-18), name
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            tr.fit(GeneralistModel(), list(make_tasks()))
