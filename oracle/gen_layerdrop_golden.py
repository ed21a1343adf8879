"""Generate tests/golden/layerdrop.json by ITERATING THE REFERENCE's LayerDropModuleList (module/layer_drop.py:13-41) in training and
evaluation mode for a few (seed, p, number of layers) and recording which layer indices each iteration yields -- two consecutive
iterations per seed, so the fixture also pins how many random numbers one iteration consumes.  Build container only.
TEST INFRASTRUCTURE: only data is stored."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_import import install  # noqa: E402

CASES = [(seed, p, n) for seed in (0, 7, 11) for p, n in ((0.5, 6), (0.2, 12), (0.9, 4), (0.0, 6))]


def main():
    install()
    import torch
    import torch.nn as nn
    from ofasys.module.layer_drop import LayerDropModuleList
    out = {"__meta__": {"torch": torch.__version__, "how": "oracle/gen_layerdrop_golden.py"}, "cases": []}
    for seed, p, n in CASES:
        layers = LayerDropModuleList(p, [nn.Identity() for _ in range(n)])
        index = {id(m): i for i, m in enumerate(nn.ModuleList.__iter__(layers))}
        layers.train()
        torch.manual_seed(seed)
        first = [index[id(m)] for m in layers]
        second = [index[id(m)] for m in layers]
        layers.eval()
        evaluated = [index[id(m)] for m in layers]
        out["cases"].append({"seed": seed, "p": p, "n": n, "train_first": first, "train_second": second, "eval": evaluated})
    path = os.path.join(ROOT, "tests", "golden", "layerdrop.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
