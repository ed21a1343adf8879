"""TEST INFRASTRUCTURE (build container only): the REFERENCE's own bf16-vs-fp32 gap of the GRADIENT NORMS inside the ResNet
backbone of a golden case -- the yard-stick for the bf16 gradient tolerance of those parameters in tests/test_model_gpu.py.
Usage: python oracle/ref_bf16_grad_gap.py tiny_video
Measured (torch CPU, bf16 parameters and activations): see the table the script prints; summary in DESIGN.md section 2."""
import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import recipe
from oracle.ref_import import build_reference_model, install
from oracle.cases import CASES, VOCAB_EXTRA, make_value, make_target
name = sys.argv[1]
case = CASES[name]
install()
import ofasys  # noqa
from ofasys import ModalityType
from ofasys.preprocessor import Slot
from ofasys.engine.criterion.cross_entropy import nll_loss


def run(dtype):
    model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
    recipe.fill_state(model.state_dict())
    model.eval()
    if case.get("train"):
        model.train()
    if "drop_seed" in case:                # stochastic depth: both dtypes replay the golden's recorded draws
        import numpy as np
        from oracle.ref_import import force_drop_path_draws
        force_drop_path_draws(model, np.load(f"/root/repo/tests/golden/{name}.npz")["droppath_keep"])
    model.to(dtype)
    slots, prev = [], None
    for mod, is_src, spec, attrs in case["slots"]:
        v = make_value(spec, len(d))
        if isinstance(v, dict):
            v = {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in v.items()}
        elif v.is_floating_point():
            v = v.to(dtype)
        slots.append(Slot(ModalityType[mod], is_src, v, attributes=attrs))
        if not is_src:
            prev = v
    target = make_target(prev)
    out = model(slots)
    lprobs = model.get_normalized_probs(out, log_probs=True)
    loss = nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=d.pad(), reduce=True)
    loss.backward()
    return {k: float(p.grad.double().norm()) for k, p in model.named_parameters() if p.grad is not None}


a = run(torch.float32)
b = run(torch.bfloat16)
worst = 0.0
for k in a:
    if "embed_images" in k and a[k] > 0:
        r = b[k] / a[k]
        worst = max(worst, abs(r - 1))
        if "layer1.0" in k or k.endswith("embed_images.conv1.weight") or "bn1" in k.split("embed_images.")[1][:4]:
            print(f"{k.split('embed_images.')[1]:36s} bf16/fp32 grad-norm ratio {r:6.3f}")
print(name, "worst |ratio - 1| over the backbone parameters:", round(worst, 3))
