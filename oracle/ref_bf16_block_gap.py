"""TEST INFRASTRUCTURE (build container only): the REFERENCE's own bf16 error of ONE ResNet bottleneck's backward.

For every checkpointed block of a golden case (`block_grads`, tests/golden/<case>.npz: dL/d(output) and dL/d(input) recorded from
the reference's fp32 run) the reference is run in bf16 (torch CPU, bf16 parameters and activations), the golden fp32 dL/d(output) is
injected -- rounded to bf16 -- into the backward of THAT block alone, and the resulting dL/d(input) is compared with the golden
fp32 one ("chain" gap: the block's input already carries the bf16 forward's drift, and with 32 values per channel
the BatchNorm / ReLU gates make that a large number).  Second figure ("matched"): the bf16 block against a float copy of the same
block on the SAME bf16 input and gradient -- the error the block's own bf16 arithmetic adds.  The per-block figures (max |diff| / max |ref|, and the norm ratio) is what a correct bf16 implementation of the block pays
for bf16 activations and weights; tests/test_model_gpu.py::test_resnet_block_backward_bf16_pinned_per_bottleneck holds this
build's bf16 kernels to 3x it ("a 30 % error in a bf16 conv gradient passes" under the whole-trunk bounds).

Usage: python oracle/ref_bf16_block_gap.py tiny_resnet   ->  tests/golden/resnet_block_bf16_gap.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
from oracle import recipe  # noqa: E402
from oracle.cases import CASES, VOCAB_EXTRA, make_value  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_resnet"
case = CASES[name]
install()
import ofasys  # noqa: E402,F401
from ofasys import ModalityType  # noqa: E402
from ofasys.preprocessor import Slot  # noqa: E402

g = np.load(f"/root/repo/tests/golden/{name}.npz")
model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
recipe.fill_state(model.state_dict())
model.eval()
if case.get("train"):
    model.train()
model.to(torch.bfloat16)
backbone = dict(model.named_modules())["encoder.adaptor.image_resnet.embed_images"]
io = {}
hooks = []
for lname in ("layer1", "layer2", "layer3"):
    for bi, blk in enumerate(getattr(backbone, lname)):
        def keep(m, i, o, key=f"{lname}.{bi}"):
            io[key] = (i[0], o)
        hooks.append(blk.register_forward_hook(keep))
slots = []
for mod, is_src, spec, attrs in case["slots"]:
    v = make_value(spec, len(d))
    if torch.is_tensor(v) and v.is_floating_point():
        v = v.to(torch.bfloat16)
    slots.append(Slot(ModalityType[mod], is_src, v, attributes=attrs))
out = model(slots)
res = {}
for k in case["block_grads"]:
    x, y = io[k]
    dy = torch.from_numpy(g[f"blockgrad.{k}.dy"]).to(torch.bfloat16)
    (dx,) = torch.autograd.grad(y, x, dy, retain_graph=True)
    want = torch.from_numpy(g[f"blockgrad.{k}.dx"]).double()
    got = dx.double()
    res[k] = {"max_rel": float((got - want).abs().max() / want.abs().max()),
              "norm_rel": float((got - want).norm() / want.norm()),
              "norm_ratio": float(got.norm() / want.norm())}
    # ... and the WITHIN-block part of that error: the same block (a float copy: bf16-valued weights, fp32 arithmetic) on the SAME bf16
    # input and the same injected gradient -- what remains is the bf16 rounding of the block's own intermediate activations
    import copy
    blk = dict(backbone.named_modules())[k]
    blk32 = copy.deepcopy(blk).float()
    blk32.train(blk.training)
    x32 = x.detach().float().requires_grad_(True)
    y32 = blk32(x32)
    (dx32,) = torch.autograd.grad(y32, x32, dy.float())
    dx32 = dx32.double()
    res[k]["matched_max_rel"] = float((got - dx32).abs().max() / dx32.abs().max())
    res[k]["matched_norm_rel"] = float((got - dx32).norm() / dx32.norm())
    print(f"{k:10s} reference bf16 block backward vs the golden fp32 run: max {res[k]['max_rel']:.3e}  rel-norm of the difference "
          f"{res[k]['norm_rel']:.3e}  norm ratio {res[k]['norm_ratio']:.4f};  vs the fp32 block on the SAME bf16 input: max "
          f"{res[k]['matched_max_rel']:.3e}  rel-norm {res[k]['matched_norm_rel']:.3e}")
path = "/root/repo/tests/golden/resnet_block_bf16_gap.json"
allres = json.load(open(path)) if os.path.exists(path) else {}
allres[name] = res
json.dump(allres, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
