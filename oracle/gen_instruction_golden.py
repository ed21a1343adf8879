"""Generate tests/golden/instruction_parse.json by RUNNING THE REFERENCE's Instruction (preprocessor/instruction.py:116-279)
on oracle/instruction_cases.py (build container only).  TEST INFRASTRUCTURE: only data is stored."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import instruction_cases as IC  # noqa: E402
from oracle.ref_import import install  # noqa: E402


def main():
    install()
    from ofasys.preprocessor import Instruction
    out = {"parse": [], "format": [], "bad": []}
    for t in IC.TEMPLATES:
        for split, dpl in IC.SPLITS:
            ist = Instruction(t, split=split, decoder_plain_with_loss=dpl)
            out["parse"].append({"template": t, "split": split, "decoder_plain_with_loss": dpl,
                                 "slots": [IC.slot_record(s) for s in ist.slots], "names": ist.get_slot_names(), "str": str(ist)})
    for ti, args, kw in IC.FORMATS:
        f = Instruction(IC.TEMPLATES[ti]).format(*args, **dict(kw))
        out["format"].append({"values": [s.value for s in f.slots], "others": f.others, "str": str(f)})
    for t in IC.BAD:
        try:
            Instruction(t)
            out["bad"].append("ok")
        except ValueError:
            out["bad"].append("ValueError")
    try:
        Instruction(IC.TEMPLATES[0]).format(caption="only the target")
        out["missing_source"] = "ok"
    except ValueError as e:
        out["missing_source"] = str(e)
    try:
        Instruction(IC.TEMPLATES[0]).format("a", "b", "c")
        out["extra_args"] = "ok"
    except ValueError as e:
        out["extra_args"] = str(e)
    path = os.path.join(ROOT, "tests", "golden", "instruction_parse.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
