"""TEST INFRASTRUCTURE (build container only): the REFERENCE's own bf16-vs-fp32 logit gap on a golden case -- the yard-stick for
the bf16 tolerances in tests/test_model_gpu.py.  Usage: python oracle/ref_bf16_gap.py <case>
Measured: tiny_text 1.0e-2, tiny_resnet 4.7e-2, tiny_video 3.0e-2 (max |diff| / max |logit|)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import recipe
from oracle.ref_import import build_reference_model, install
from oracle.cases import CASES, EXTRA_CASES, VOCAB_EXTRA, make_value, make_target
name = sys.argv[1]
case = CASES.get(name) or EXTRA_CASES[name]
install()
import ofasys
from ofasys import ModalityType
from ofasys.preprocessor import Slot
model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
recipe.fill_state(model.state_dict())
model.eval()
if case.get("train"): model.train()
if "drop_seed" in case:                    # stochastic depth: both dtypes replay the golden's recorded draws
    import numpy as np
    from oracle.ref_import import force_drop_path_draws
    force_drop_path_draws(model, np.load(f"/root/repo/tests/golden/{name}.npz")["droppath_keep"])
def run(dtype):
    slots = []
    for mod, is_src, spec, attrs in case["slots"]:
        v = make_value(spec, len(d))
        if isinstance(v, dict): v = {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in v.items()}
        elif v.is_floating_point(): v = v.to(dtype)
        slots.append(Slot(ModalityType[mod], is_src, v, attributes=attrs))
    with torch.no_grad():
        return model(slots)[0].float()
a = run(torch.float32)
recipe.fill_state(model.state_dict())          # reset BN running stats
model.to(torch.bfloat16)
b = run(torch.bfloat16)
print(name, "ref bf16 vs fp32: max abs / max |logit| =", float((a - b).abs().max() / a.abs().max()))
