import sys, json, torch
sys.path.insert(0, '.')
from oracle.cases import CASES
from tests.golden_util import case_inputs, load_golden
from tests.model_util import build_model, make_slots
from ofasys_amd import ops
name = "tiny_video"
case = CASES[name]; g = load_golden(name)
model, d = build_model(case, "cuda", torch.bfloat16)
model.train()
vals, target = case_inputs(case)
logits, extra, enc = model(make_slots(vals, "cuda", torch.bfloat16), return_encoder_out=True)
loss = ops.cross_entropy_sum(logits, target.cuda(), d.pad())
loss.backward()
gn = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]))
out = {}
for k, p in model.named_parameters():
    if "embed_images" in k and p.grad is not None:
        out[k] = (float(p.grad.double().norm()), float(gn.get(k, -1)))
json.dump(out, open(sys.argv[1], "w"))
