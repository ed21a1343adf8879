"""CPU: ofasys_amd.Instruction (template parser + format) against the parse recorded from the reference's Instruction
(tests/golden/instruction_parse.json, oracle/gen_instruction_golden.py; preprocessor/instruction.py:116-279)."""
import json
import os

import pytest

from oracle import instruction_cases as IC
from ofasys_amd import Instruction, ModalityType, Slot

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "instruction_parse.json")))


def test_parse_matches_reference():
    it = iter(G["parse"])
    for t in IC.TEMPLATES:
        for split, dpl in IC.SPLITS:
            want = next(it)
            ist = Instruction(t, split=split, decoder_plain_with_loss=dpl)
            assert [IC.slot_record(s) for s in ist.slots] == want["slots"], t
            assert ist.get_slot_names() == want["names"] and str(ist) == want["str"]


def test_format_matches_reference():
    for (ti, args, kw), want in zip(IC.FORMATS, G["format"]):
        base = Instruction(IC.TEMPLATES[ti])
        f = base.format(*args, **dict(kw))
        assert [s.value for s in f.slots] == want["values"] and f.others == want["others"] and str(f) == want["str"]
        assert all(s.value is None for s in base.slots if not s.is_plaintext)          # format returns a copy


def test_errors_match_reference():
    for t, want in zip(IC.BAD, G["bad"]):
        assert want == "ValueError"
        with pytest.raises(ValueError):
            Instruction(t)
    with pytest.raises(ValueError) as e:
        Instruction(IC.TEMPLATES[0]).format(caption="only the target")
    assert str(e.value) == G["missing_source"]
    with pytest.raises(ValueError) as e:
        Instruction(IC.TEMPLATES[0]).format("a", "b", "c")
    assert str(e.value) == G["extra_args"]


def test_slot_level_construction():
    ist = Instruction([Slot(ModalityType.TEXT, True, 1)], "tmpl", {"uid": 1})
    assert ist.template == "tmpl" and ist.others == {"uid": 1} and len(ist.slots) == 1
