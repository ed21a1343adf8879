"""Line-level similarity of our files against reference files (difflib ratio on stripped, comment-free lines)."""
import difflib, sys, re
def norm(path):
    out = []
    for l in open(path, errors="ignore"):
        l = l.split("#")[0].strip()
        if l and not l.startswith(('"""', "'''")):
            out.append(re.sub(r"\s+", " ", l))
    return out
pairs = [
 ("ofasys_amd/preprocessor/collate.py", ["preprocessor/utils.py", "preprocessor/default/text.py", "preprocessor/general.py", "preprocessor/default/box.py", "preprocessor/default/base.py"]),
 ("ofasys_amd/module/multihead_attention.py", ["module/multihead_attention.py"]),
 ("ofasys_amd/module/transformer_layer.py", ["module/transformer_layer.py"]),
 ("ofasys_amd/model/transformer.py", ["model/transformer.py"]),
 ("ofasys_amd/preprocessor/dictionary.py", ["preprocessor/dictionary.py"]),
 ("ofasys_amd/preprocessor/instruction.py", ["preprocessor/instruction.py"]),
 ("oracle/restate.py", ["module/multihead_attention.py", "model/transformer.py"]),
]
for mine, refs in pairs:
    a = norm("/root/repo/" + mine)
    for r in refs:
        b = norm("/root/reference/ofasys/" + r)
        sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
        same = sum(m.size for m in sm.get_matching_blocks())
        print(f"{mine:45s} vs {r:38s} ratio {sm.ratio():.2f}  matching lines {same}/{len(a)} of mine, {same}/{len(b)} of ref")
