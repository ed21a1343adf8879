"""TEST INFRASTRUCTURE (build container only): how the CPU oracle (oracle/restate.py, bench.py's cpu_baseline leg, kind "port")
compares in SPEED with the reference itself on the same host, same threads, same inputs: forward + loss + backward of the cfg-2
shape (OFA-base, image_patch_embed 224 x 224 + 191 text -> 64 target positions) at a bounded batch.
Usage: python oracle/ref_cpu_timing.py [batch] [threads]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, restate  # noqa: E402
from oracle.cases import CASES, VOCAB_EXTRA, make_target  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402
from tests.golden_util import oracle_cfg  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(32, os.cpu_count() or 8)
torch.set_num_threads(threads)
case = CASES["base_patch"]
install()
import ofasys  # noqa: E402,F401
from ofasys import ModalityType  # noqa: E402
from ofasys.preprocessor import Slot  # noqa: E402
from ofasys.engine.criterion.cross_entropy import nll_loss  # noqa: E402

model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
recipe.fill_state(model.state_dict())
model.eval()
V = len(d)
img = recipe.floats("timing.image", (B, 3, 224, 224))
src = recipe.tokens("timing.src", (B, 191), V, [191 - 7 * (i % 5) for i in range(B)])
prev = recipe.tokens("timing.prev", (B, 64), V, [64 - 5 * (i % 4) for i in range(B)], bos=0)
target = make_target(prev)
ntok = int((src != 1).sum() + (prev != 1).sum()) + 257 * B


def ref_step():
    slots = [Slot(ModalityType.IMAGE, True, img, attributes=["adaptor=image_patch_embed"]), Slot(ModalityType.TEXT, True, src),
             Slot(ModalityType.TEXT, False, prev)]
    out = model(slots)
    lprobs = model.get_normalized_probs(out, log_probs=True)
    loss = nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=d.pad(), reduce=True)
    model.zero_grad()
    loss.backward()
    return float(loss)


state = {k: v.detach().clone() for k, v in model.state_dict().items()}
state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
params = [v.requires_grad_(True) for k, v in state.items() if v.is_floating_point() and not k.endswith("version")
          and not k.startswith("decoder.adaptor.embed_tokens")]
cfg = oracle_cfg(case)


def oracle_step():
    oslots = [restate.OSlot("IMAGE", True, img, ["adaptor=image_patch_embed"]), restate.OSlot("TEXT", True, src, None),
              restate.OSlot("TEXT", False, prev, None)]
    logits, _ = restate.model_forward(state, cfg, oslots)
    loss, _ = restate.cross_entropy(logits, target)
    for p in params:
        p.grad = None
    loss.backward()
    return float(loss)


def clock(fn, n=3):
    fn()
    t = time.perf_counter()
    for _ in range(n):
        last = fn()
    return (time.perf_counter() - t) / n, last


tr, lr = clock(ref_step)
to, lo = clock(oracle_step)
print(f"cfg-2 shape, batch {B}, {threads} threads, {ntok} non-pad positions per step (forward + loss + backward, fp32)")
print(f"reference (ofasys, torch CPU): {tr:.2f} s/step = {ntok / tr:.0f} tokens/s   loss {lr:.4f}")
print(f"oracle port (oracle/restate.py): {to:.2f} s/step = {ntok / to:.0f} tokens/s   loss {lo:.4f}")
print(f"port / reference step time: {to / tr:.2f}")
