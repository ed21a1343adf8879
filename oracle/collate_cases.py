"""Seeded collation cases shared by oracle/gen_collate_golden.py (runs the REFERENCE on them, build container only) and
tests/test_collate_cpu.py (runs ofasys_amd.preprocessor on the same inputs).  TEST INFRASTRUCTURE."""
import numpy as np
import torch

N_TEXT = 300            # <text>_0 .. <text>_299 after the 4 specials, then <mask>, then 1000 <bin>_k (added by the box preprocessor)


def _rng(seed):
    return np.random.Generator(np.random.Philox(key=seed))


# ---- collate_tokens: {"seed", "lengths", "dim2", "kwargs"}
TOKEN_CASES = {
    "ragged": dict(seed=1, lengths=[5, 1, 9, 3], dim2=None, kwargs=dict(pad_idx=1, eos_idx=2)),
    "left_pad": dict(seed=2, lengths=[4, 7, 2], dim2=None, kwargs=dict(pad_idx=1, eos_idx=2, left_pad=True)),
    "move_eos": dict(seed=3, lengths=[6, 3, 8], dim2=None, kwargs=dict(pad_idx=1, eos_idx=2, move_eos_to_beginning=True)),
    "move_eos_last": dict(seed=4, lengths=[6, 3], dim2=None, kwargs=dict(pad_idx=1, eos_idx=None, move_eos_to_beginning=True, left_pad=True)),
    "multiple8": dict(seed=5, lengths=[9, 16, 3], dim2=None, kwargs=dict(pad_idx=1, pad_to_multiple=8)),
    "to_length": dict(seed=6, lengths=[2, 5], dim2=None, kwargs=dict(pad_idx=0, pad_to_length=12)),
    "masks2d": dict(seed=7, lengths=[3, 6, 1], dim2=10, kwargs=dict(pad_idx=0)),
    "single": dict(seed=8, lengths=[4], dim2=None, kwargs=dict(pad_idx=1)),
    "exact_multiple": dict(seed=9, lengths=[8, 4], dim2=None, kwargs=dict(pad_idx=1, pad_to_multiple=4)),
}


def make_token_lists(spec):
    g = _rng(spec["seed"])
    out = []
    for n in spec["lengths"]:
        if spec["dim2"]:
            out.append(torch.from_numpy(g.integers(0, 2, (n, spec["dim2"])).astype(bool)))
        else:
            out.append(torch.from_numpy(g.integers(4, 300, (n,)).astype(np.int64)))
    return out


# ---- instruction-level cases.  A sample is a list of raw slots (modality, is_src, value, attributes, is_plaintext);
# text values are np.int64 arrays of dictionary-relative ids (the reference maps id w to the symbol '<text>_w'),
# box values are [1,4] float tensors, image values float tensors.
CASES = {
    # caption-like: [IMAGE] + two adjacent source text slots (merged) -> one target text slot
    "caption": dict(seed=11, batch=3, template="[IMAGE:img] what does the image describe? -> [TEXT:cap]",
                    layout=[("IMAGE", True, ("img", (3, 8, 8)), None, False), ("TEXT", True, ("tok", 2, 6), None, True),
                            ("TEXT", True, ("tok", 1, 5), None, False), ("TEXT", False, ("tok", 2, 9), None, False)]),
    # grounding-like: text + box on the source side are ONE text group (box groups under TEXT); target = plaintext
    # (no loss, becomes the prefix at inference: split = test) followed by a box
    "grounding": dict(seed=12, batch=4, split="test", template="which region ... [TEXT] [BOX] -> region: [BOX]",
                      layout=[("TEXT", True, ("tok", 3, 7), None, False), ("BOX", True, ("box",), None, False),
                              ("TEXT", False, ("tok", 2, 2), None, True), ("BOX", False, ("box",), None, False)]),
    # truncation + pad_to_multiple + disable_auto_boseos on the source side, max_length attribute on the target
    "truncate": dict(seed=13, batch=3, max_src_length=6, max_tgt_length=5, pad_to_multiple=4, template="t",
                     layout=[("TEXT", True, ("tok", 5, 12), ["disable_auto_boseos"], False),
                             ("TEXT", False, ("tok", 6, 11), ["max_length=7"], False)]),
    # closed answer set: constraint masks from the trie, shifted by one at collation; no_loss slot in front of it
    "closed_set": dict(seed=14, batch=3, template="c", closed_set=[[10, 11], [10, 12, 13], [20]],
                       answers=[[10, 11], [20], [10, 12, 13]],
                       layout=[("TEXT", True, ("tok", 2, 4), None, False), ("TEXT", False, ("tok", 1, 3), ["no_loss"], False),
                               ("TEXT", False, ("answer",), ["closed_set"], False)]),
}

BOX_CASES = {"plain": [12.0, 40.5, 300.25, 511.0], "half_even": [0.2562562562, 0.7687687687, 256.0, 512.0],
             "zero": [0.0, 0.0, 0.0, 0.0], "ties": [1.2812812812, 2.3063063063, 3.3313313313, 4.3563563563]}


def make_samples(case):
    g = _rng(case["seed"])
    samples = []
    for b in range(case["batch"]):
        raw = []
        for mod, is_src, spec, attrs, plain in case["layout"]:
            if spec[0] == "tok":
                n = int(g.integers(spec[1], spec[2] + 1))
                v = g.integers(0, N_TEXT, (n,)).astype(np.int64)
            elif spec[0] == "answer":
                v = np.asarray(case["answers"][b], dtype=np.int64)
            elif spec[0] == "box":
                v = torch.from_numpy((g.random((1, 4)) * 512).astype(np.float32))
            else:
                v = torch.from_numpy(g.standard_normal(spec[1]).astype(np.float32))
            raw.append((mod, is_src, v, list(attrs) if attrs else None, plain))
        samples.append(raw)
    return samples
