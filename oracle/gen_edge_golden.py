"""Generate tests/golden/edge_shapes.npz by RUNNING THE REFERENCE (tiny arch, text -> text, default flags, eval mode) on the edge
shapes of oracle/edge_cases.py: logits, loss, sample size and every parameter's gradient norm per shape.  Build container only.
TEST INFRASTRUCTURE: only data is stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edge_cases as EC, recipe  # noqa: E402
from oracle.cases import CASES, VOCAB_EXTRA  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402


def main():
    install()
    import ofasys  # noqa: F401
    from ofasys import ModalityType
    from ofasys.preprocessor import Slot
    from ofasys.engine.criterion.cross_entropy import nll_loss
    case = CASES["tiny_text"]
    model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
    recipe.fill_state(model.state_dict())
    model.eval()
    torch.set_num_threads(8)
    out = {}
    names = sorted(k for k, _ in model.named_parameters())
    out["grad_norm_keys"] = np.array(names)
    for shape in EC.SHAPES:
        src, prev, target = EC.inputs(shape)
        logits, extra = model([Slot(ModalityType.TEXT, True, src), Slot(ModalityType.TEXT, False, prev)])
        lprobs = model.get_normalized_probs((logits, extra), log_probs=True)
        loss = nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=d.pad(), reduce=True)
        model.zero_grad()
        loss.backward()
        params = dict(model.named_parameters())
        k = EC.key(shape)
        out[k + ".logits"] = logits.detach().numpy()
        out[k + ".loss"] = loss.detach().reshape(1).numpy()
        out[k + ".target"] = target.numpy()
        out[k + ".grad_norms"] = np.array([float(params[n].grad.double().norm()) if params[n].grad is not None else -1.0 for n in names])
        print(k, "loss", float(loss), "finite", bool(torch.isfinite(logits).all()))
    src, prev = EC.max_position_inputs()                   # forward only, as the GPU test does
    with torch.no_grad():
        logits, extra = model([Slot(ModalityType.TEXT, True, src), Slot(ModalityType.TEXT, False, prev)])
    out["maxpos.logits"] = logits.reshape(-1)[::EC.MAX_POS_STRIDE].numpy()
    out["maxpos.attn"] = extra["attn"][0].reshape(-1)[::EC.MAX_POS_STRIDE].numpy()
    out["maxpos.logits_absmax"] = np.array([float(logits.abs().max())])
    print("maxpos", tuple(logits.shape), "finite", bool(torch.isfinite(logits).all()))
    path = os.path.join(ROOT, "tests", "golden", "edge_shapes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
