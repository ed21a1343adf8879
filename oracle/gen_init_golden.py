"""Generate tests/golden/init_stats.json by BUILDING THE REFERENCE's GeneralistModel (torch.manual_seed(SEED); GeneralistModel();
initialize(dictionary) -> apply(init_bert_params), model/ofa.py:380) for the configurations of oracle/init_cases.py and recording,
per state-dict entry, shape / mean / std / |max| and a digest of the bytes.  Build container only.  TEST INFRASTRUCTURE: only
data is stored."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import init_cases as IC  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402


def one_case(name):
    c = IC.CASES[name]
    m, _ = build_reference_model(c["arch"], IC.VOCAB_EXTRA, set(c["active"]), c["overrides"], c["adaptor_overrides"], seed=IC.SEED)
    sd = m.state_dict()
    rec = {k: IC.tensor_record(v) for k, v in sd.items()}
    zeros = sum(1 for r in rec.values() if r["absmax"] == 0.0)
    print(name, len(sd), "entries;", zeros, "all-zero;", sum(v.numel() for v in sd.values()), "elements", file=sys.stderr)
    return rec


def main():
    import subprocess
    import torch
    if len(sys.argv) > 1:                         # child: one case, JSON on stdout
        json.dump(one_case(sys.argv[1]), sys.stdout)
        return
    out = {"__meta__": {"torch": torch.__version__, "seed": IC.SEED, "vocab_extra": IC.VOCAB_EXTRA,
                        "how": "oracle/gen_init_golden.py: reference GeneralistModel.initialize on CPU, ONE PROCESS PER CASE (the "
                               "reference's adaptor configs are shared dataclass defaults: a second model built in the same "
                               "process inherits the first one's layer counts, SURVEY.md section 5)"}}
    for name in IC.CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, check=True)
        sys.stderr.write(r.stderr.splitlines()[-1] + "\n")
        out[name] = json.loads(r.stdout[r.stdout.index("{"):])
    path = os.path.join(ROOT, "tests", "golden", "init_stats.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
