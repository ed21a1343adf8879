"""How far apart are two fp32 implementations of the SAME ResNet-50 backbone backward (torch CPU vs torch GPU/MIOpen)
on the tiny_resnet geometry (B=2, 64x64: BatchNorm statistics over as few as 32 values per channel)?"""
import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import recipe, restate
from tests.golden_util import load_golden, oracle_cfg
from oracle.cases import CASES
case = CASES["tiny_resnet"]; g = load_golden("tiny_resnet")
cfg = oracle_cfg(case)
p = "encoder.adaptor.image_resnet.embed_images"
keys = []
for item in g["state_keys"]:
    key, shape, _ = str(item).split("|")
    if key.startswith(p):
        keys.append((key, tuple(int(x) for x in shape.strip("()").split(",") if x.strip())))
def run(dev):
    state = {}
    for k, sh in keys:
        if k.endswith("num_batches_tracked"):
            state[k] = torch.zeros((), dtype=torch.long, device=dev)
        else:
            state[k] = recipe.value_for(k, sh).to(dev)
            if not k.endswith(("running_mean", "running_var")):
                state[k].requires_grad_(True)
    img = recipe.floats("input.image", (2, 3, 64, 64)).to(dev)
    feat = restate.resnet_backbone(state, p, img, cfg)
    w = recipe.floats("probe", tuple(feat.shape)).to(dev)
    (feat * w).sum().backward()
    return feat.detach().cpu(), {k: v.grad.detach().cpu() for k, v in state.items() if v.requires_grad}
fc, gc = run("cpu")
fg, gg = run("cuda")
print("features rel", float((fc - fg).abs().max() / fc.abs().max()))
for k in [p + ".conv1.weight", p + ".layer1.0.downsample.0.weight", p + ".layer2.1.conv2.weight", p + ".layer3.5.bn3.weight", p + ".bn1.bias"]:
    print(k[-40:], "max rel", float((gc[k] - gg[k]).abs().max() / gc[k].abs().max()), "norm rel", float((gc[k].norm() - gg[k].norm()).abs() / gc[k].norm()))
