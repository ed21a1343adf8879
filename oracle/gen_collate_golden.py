"""Generate tests/golden/collate.npz by RUNNING THE REFERENCE's collation functions (build container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_collate_golden.py
The reference's `collate_tokens`, `DefaultTextPreprocess.map / group_map / collate`, `DefaultBoxPreprocess.map` and
`GeneralPreprocess.collate` (preprocessor/utils.py:75-113, default/text.py:100-313, default/box.py:101-110,
general.py:79-144) are executed on seeded synthetic samples; inputs are described by (seed, shape) recipes that the test
re-creates, the integer outputs are stored.  The text / box preprocessors are instantiated WITHOUT their constructors
(which download BPE vocabularies): only the attributes the collation code reads are set.  Only data is stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.collate_cases import (CASES, TOKEN_CASES, BOX_CASES, N_TEXT, make_samples, make_token_lists)  # noqa: E402
from oracle.ref_import import install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "collate.npz")


def ref_dictionary():
    from ofasys.preprocessor.dictionary import Dictionary
    d = Dictionary()
    for i in range(N_TEXT):
        d.add_symbol(f"<text>_{i}")
    d.add_symbol("<mask>")
    return d


def ref_preprocessors(d, case):
    from ofasys import ModalityType
    from ofasys.preprocessor.default.base import PreprocessConfig
    from ofasys.preprocessor.default.box import BoxPreprocessConfig, DefaultBoxPreprocess
    from ofasys.preprocessor.default.image import DefaultImagePreprocess
    from ofasys.preprocessor.default.text import DefaultTextPreprocess, TextPreprocessConfig
    from ofasys.utils.trie import Trie
    text = object.__new__(DefaultTextPreprocess)
    cfg = TextPreprocessConfig()
    cfg.max_src_length, cfg.max_tgt_length = case.get("max_src_length", 1024), case.get("max_tgt_length", 1024)
    cfg.pad_to_multiple = case.get("pad_to_multiple", 1)
    text.global_dict, text.cfg, text._modality_type, text._sanity_check = d, cfg, ModalityType.TEXT, False
    text.dict_text_start, text.dict_text_end = d.get_start_end_idx("<text>")
    text.dict_text_end += 1
    text.constraint_trie = None
    if case.get("closed_set"):
        text.constraint_trie = Trie(d.eos())
        for ans in case["closed_set"]:
            text.constraint_trie.insert([d.bos()] + [4 + t for t in ans] + [d.eos()])
    box = object.__new__(DefaultBoxPreprocess)
    bcfg = BoxPreprocessConfig()
    box.global_dict, box.cfg, box._modality_type, box._sanity_check = d, bcfg, ModalityType.BOX, False
    box.num_bins, box.max_image_size = bcfg.box_dict_size, bcfg.max_image_size
    for i in range(box.num_bins):
        d.add_symbol("<bin>_{}".format(i))
    box.dict_start, box.dict_end = d.get_start_end_idx("<bin>")
    box.constraint_trie = None
    box.instruction_map = lambda ist: ist        # the image/box co-transform decodes image files: out of scope, boxes arrive resized
    image = object.__new__(DefaultImagePreprocess)
    image.global_dict, image.cfg, image._modality_type, image._sanity_check = d, PreprocessConfig(), ModalityType.IMAGE, False
    image.map = lambda slot: slot                # image decoding / augmentation is out of scope: tensors arrive ready
    return {"text": text, "box": box, "image": image}


def main():
    install()
    import ofasys  # noqa: F401
    from ofasys import ModalityType
    from ofasys.preprocessor import Slot
    from ofasys.preprocessor.general import GeneralPreprocess
    from ofasys.preprocessor.instruction import Instruction
    from ofasys.preprocessor.utils import collate_tokens
    out = {}
    for name, spec in TOKEN_CASES.items():
        vals = make_token_lists(spec)
        kw = dict(spec["kwargs"])
        out[f"tok.{name}"] = collate_tokens(vals, **kw).numpy()
    for name, case in CASES.items():
        d = ref_dictionary()
        pres = ref_preprocessors(d, case)
        gp = object.__new__(GeneralPreprocess)
        gp.global_dict, gp.name2pre = d, pres
        samples = []
        for raw in make_samples(case):
            slots = [Slot(ModalityType[m], is_src, v, global_position=i, attributes=attrs, split=case.get("split", "train"),
                          is_plaintext=plain)
                     for i, (m, is_src, v, attrs, plain) in enumerate(raw)]
            ist = object.__new__(Instruction)
            ist.slots, ist.others, ist.template = slots, {"uid": len(samples)}, case["template"]
            samples.append(GeneralPreprocess.__call__(gp, ist))
        out[f"{name}.n_slots"] = np.array([len(samples[0].slots)])
        for si, s in enumerate(samples):                                     # grouped (per-sample) values
            for gi, slot in enumerate(s.slots):
                if isinstance(slot.value, dict):
                    for k, v in slot.value.items():
                        if v is not None:
                            out[f"{name}.group.{si}.{gi}.{k}"] = v.numpy()
        res = GeneralPreprocess.collate(gp, samples)
        for gi, slot in enumerate(res["net_input"]["slots"]):
            out[f"{name}.net_input.{gi}"] = slot.value.numpy()
            out[f"{name}.net_input.{gi}.is_src"] = np.array([int(slot.is_src)])
        for gi, slot in enumerate(res["net_target"]["slots"]):
            out[f"{name}.net_target.{gi}"] = slot.value.numpy()
        for k in ("target", "prefix_tokens", "constraint_masks"):
            if k in res and res[k] is not None:
                out[f"{name}.extra.{k}"] = res[k].numpy()
        out[f"{name}.extra.ntokens"] = np.array([res["ntokens"]])
        out[f"{name}.extra.dict_range"] = np.array([res["dict_start"], res["dict_end"]])
        out[f"{name}.uid"] = np.asarray(res["uid"])
    d = ref_dictionary()
    box = ref_preprocessors(d, {})["box"]
    for name, coords in BOX_CASES.items():
        s = Slot(ModalityType.BOX, True, torch.tensor([coords], dtype=torch.float32), global_position=0)
        toks = box.map(s).value
        out[f"box.{name}"] = toks.numpy()
        out[f"box.{name}.decode"] = box.decode(torch.cat([toks, torch.tensor([d.eos()])]), 0.5, 2.0).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
