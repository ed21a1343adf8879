"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only; needs /root/reference).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden.py
The reference is imported through oracle/ref_import.py (third-party stubs), its parameters are
overwritten with the shared recipe (oracle/recipe.py), it is run in eval mode (dropout off: bitwise
dropout parity with a fused kernel is not a goal) on recipe inputs, and inputs'
descriptions + outputs + gradients are stored.  Only data is stored -- no reference source.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

from oracle.cases import CASES, ORACLE_CASES, VOCAB_EXTRA, make_value, make_target  # noqa: E402


def run_case(name, case):
    install()
    import ofasys  # noqa: F401
    from ofasys import ModalityType
    from ofasys.preprocessor import Slot
    from ofasys.engine.criterion.cross_entropy import nll_loss

    model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
    recipe.fill_state(model.state_dict())
    if case.get("half"):               # (modal_ffn: the reference forces fp16 activations; the whole run is in half, on CPU)
        model.half()
    model.eval()
    if case.get("train"):          # dropout is 0 in such cases: train mode only switches BatchNorm to batch statistics
        model.train()
    V = len(d)
    slots, prev = [], None
    for mod, is_src, spec, attrs in case["slots"]:
        v = make_value(spec, V)
        slots.append(Slot(ModalityType[mod], is_src, v, attributes=attrs))
        if not is_src:
            prev = v
    target = make_target(prev)

    rec = {}
    hooks = []
    for l, layer in enumerate(model.encoder.layers):
        hooks.append(layer.register_forward_hook(lambda m, i, o, l=l: rec.__setitem__(f"enc_layer{l}", o[0].detach())))
    for l, layer in enumerate(model.decoder.layers):
        hooks.append(layer.register_forward_hook(lambda m, i, o, l=l: rec.__setitem__(f"dec_layer{l}", o[0].detach())))

    def enc_adaptor_hook(m, i, o):
        rec["enc_embed"] = o[0].detach().clone()
        rec["enc_pos_embed"] = o[2].detach().clone()
        if o[3] is not None:
            rec["enc_bias0"] = o[3][0].detach().clone()
            rec["enc_bias_last"] = o[3][-1].detach().clone()
    hooks.append(model.encoder.adaptor.register_forward_hook(enc_adaptor_hook))

    # per-bottleneck gradient checkpoints of the ResNet backbone (cases with "block_grads"): dL/d(output) and dL/d(input) of
    # a few blocks in full -- the parity test feeds the reference's dL/d(output) into ITS OWN block backward, which pins every
    # block's backward arithmetic at 1e-3 although the 16-block BatchNorm chain as a whole is ill-conditioned -- and the norm of
    # dL/d(output) of every block
    blk_io = {}
    if case.get("block_grads"):
        backbone = model.encoder.adaptor.adaptors["image_resnet"].embed_images if hasattr(model.encoder.adaptor, "adaptors") else None
        if backbone is None:
            backbone = dict(model.named_modules())["encoder.adaptor.image_resnet.embed_images"]
        for lname in ("layer1", "layer2", "layer3"):
            for bi, blk in enumerate(getattr(backbone, lname)):
                def keep(m, i, o, key=f"{lname}.{bi}"):
                    if i[0].requires_grad:
                        i[0].retain_grad()
                    o.retain_grad()
                    blk_io[key] = (i[0], o)
                hooks.append(blk.register_forward_hook(keep))

    # stochastic depth (cases with "drop_seed"): the per-sample keep decisions of every active DropPath call are recorded in call
    # order -- they are random INPUTS of the run, which the oracle and the HIP model replay
    keep_rows = []
    if "drop_seed" in case:
        for m in model.modules():
            if type(m).__name__ == "DropPath" and m.drop_prob > 0.0:
                hooks.append(m.register_forward_hook(
                    lambda m, i, o: keep_rows.append((o.detach().reshape(o.shape[0], -1).abs().amax(1) > 0).float())))
        torch.manual_seed(case["drop_seed"])

    logits, extra, enc_out = model(slots, return_encoder_out=True)
    lprobs = model.get_normalized_probs((logits, extra), log_probs=True)
    loss = nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=d.pad(), reduce=True)
    model.zero_grad()
    loss.backward()
    for h in hooks:
        h.remove()

    out = {
        "logits": logits.detach(), "loss": loss.detach().reshape(1), "target": target,
        "sample_size": torch.tensor([int(target.ne(1).sum())]),
        "attn": extra["attn"][0].detach(), "encoder_out": enc_out["encoder_out"][0].detach(),
        "encoder_padding_mask": enc_out["encoder_padding_mask"][0].to(torch.uint8),
    }
    big = case["arch"] in ("base", "large")
    for k, v in rec.items():
        if big and v.numel() > 200000:
            v = v.reshape(-1)[::97]          # strided sample keeps the file small; the test applies the same stride
        out["rec." + k] = v
    if big:
        out["encoder_out"] = out["encoder_out"].reshape(-1)[::97]
    gn = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            gn[k] = -1.0
        else:
            gn[k] = float(p.grad.double().norm())
    out["grad_norm_keys"] = np.array(sorted(gn.keys()))
    out["grad_norms"] = np.array([gn[k] for k in sorted(gn.keys())], dtype=np.float64)
    params = dict(model.named_parameters())
    for k in case["full_grads"]:
        if params[k].grad is not None:     # (an unused parameter -- e.g. the shared fc1 of a modal_ffn layer -- has no gradient: norm -1 above)
            out["grad." + k] = params[k].grad.detach()
    if keep_rows:
        out["droppath_keep"] = torch.stack(keep_rows)
        print("  drop-path keep draws:", [[int(v) for v in r] for r in keep_rows])
    for k in case.get("buffers", []):
        out["buffer." + k] = model.state_dict()[k].detach().clone()
    if blk_io:
        names = sorted(blk_io, key=lambda k: (int(k[5]), int(k.split(".")[1])))
        out["blockgrad_keys"] = np.array(names)
        out["blockgrad_norms"] = np.array([float(blk_io[k][1].grad.double().norm()) for k in names])
        for k in case["block_grads"]:
            x, y = blk_io[k]
            out["blockgrad." + k + ".dy"] = y.grad.detach().clone()          # [B, C, h, w]
            out["blockgrad." + k + ".dx"] = x.grad.detach().clone()
    out["state_keys"] = np.array([f"{k}|{tuple(v.shape)}|{str(v.dtype)}" for k, v in model.state_dict().items()])
    # integer buffers are part of the bit-exact contract
    if "encoder.adaptor.image_resnet.image_rp_bucket" in model.state_dict():
        out["image_rp_bucket_crc"] = np.array([zlib_crc(model.state_dict()["encoder.adaptor.image_resnet.image_rp_bucket"])])
    out["token_rp_bucket_crc"] = np.array([zlib_crc(model.state_dict()["encoder.adaptor.text.token_rp_bucket"])])
    out["token_rp_bucket_corner"] = model.state_dict()["encoder.adaptor.text.token_rp_bucket"][:300:7, :300:7].clone()
    out = {k: (v.float() if torch.is_tensor(v) and v.dtype == torch.float16 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print(name, "loss", float(loss), "logits", tuple(logits.shape), "file KB",
          os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024)


def zlib_crc(t):
    import zlib
    return zlib.crc32(t.contiguous().numpy().tobytes())


def softmax_vectors():
    """Golden vectors for the fused-softmax entry points.  The CUDA extensions cannot be
    built here (no nvcc); their semantics are pinned by the reference's own torch fallback
    FusedScaleMaskSoftmax.forward_torch_softmax (fused_kernels/fused_softmax.py:187-202)."""
    install()
    from ofasys.module.fused_kernels.fused_softmax import FusedScaleMaskSoftmax
    x = recipe.floats("softmax.x", (2, 3, 8, 40), 2.0)
    m = FusedScaleMaskSoftmax(False, False, "pad", True, lambda a, mask: a.masked_fill(mask.bool(), -10000.0), True, 0.37)
    y = m.forward_torch_softmax(x, None)
    mask = (recipe.floats("softmax.mask", (2, 1, 8, 40)) > 0.8)
    ym = m.forward_torch_softmax(x, mask)
    np.savez_compressed(os.path.join(OUT, "fused_softmax.npz"), x=x.numpy(), scale=np.array([0.37], dtype=np.float32),
                        y=y.numpy(), mask=mask.numpy().astype(np.uint8), y_masked=ym.numpy())
    print("fused_softmax ok")


def softmax_bwd_vectors():
    """Golden vectors for the BACKWARD entry points of the three fused-softmax extensions and for their fp16 dtype
    (scaled_masked_softmax.cpp:60-77, scaled_upper_triang_masked_softmax.cpp:49-64, scaled_softmax_cuda.cu:84-117): autograd
    through the reference's own forward_torch_softmax (fused_kernels/fused_softmax.py:187-202), in fp32 and with fp16 inputs
    (input_in_fp16, softmax_in_fp32 -- the configuration the reference routes to the kernels, multihead_attention.py:83-91)."""
    install()
    from ofasys.module.fused_kernels.fused_softmax import FusedScaleMaskSoftmax
    fill = lambda a, mask: a.masked_fill(mask.bool(), -10000.0)        # noqa: E731
    scale = 0.37
    b, np_, sq = 2, 3, 40
    x = recipe.floats("softmax_bwd.x", (b, np_, sq, sq), 2.0)
    dy = recipe.floats("softmax_bwd.dy", (b, np_, sq, sq), 1.0)
    mask = recipe.floats("softmax_bwd.mask", (b, 1, sq, sq)) > 0.8
    causal = torch.triu(torch.ones(sq, sq, dtype=torch.bool), 1).expand(b, 1, sq, sq)
    out = {"x": x.numpy(), "dy": dy.numpy(), "mask": mask.numpy().astype(np.uint8), "scale": np.array([scale], dtype=np.float32)}
    for tag, half in (("f32", False), ("f16", True)):
        m = FusedScaleMaskSoftmax(half, False, "pad", True, fill, True, scale)
        for name, mk in (("plain", None), ("masked", mask), ("causal", causal)):
            xi = (x.half() if half else x.clone()).requires_grad_(True)
            y = m.forward_torch_softmax(xi, mk)
            y.backward(dy.half() if half else dy)
            out[f"{tag}.{name}.y"] = y.detach().numpy()
            out[f"{tag}.{name}.dx"] = xi.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "fused_softmax_bwd.npz"), **out)
    print("fused_softmax_bwd ok")


def ls_ce_vectors():
    """Golden vectors of the label-smoothed criterion: the reference's own label_smoothed_nll_loss
    (engine/criterion/label_smoothed_cross_entropy.py:62-94) on log-probs prepared exactly as get_constraint_masks /
    get_lprobs_and_target / compute_loss do (:140-189; those are methods of a criterion that needs a task object)."""
    install()
    import math
    from ofasys.engine.criterion.label_smoothed_cross_entropy import label_smoothed_nll_loss
    rows, V, pad = 14, 44, 1
    logits = recipe.floats("lsce.logits", (rows, V), 2.0)
    target = recipe.tokens("lsce.target", (rows,), V)
    target[3] = pad
    target[9] = pad
    smask = recipe.floats("lsce.cmask", (rows, V)) > -0.8
    out = {"logits": logits.numpy(), "target": target.numpy(), "sample_mask": smask.numpy().astype(np.uint8)}
    for name, eps, crange, use_smask, dw in [("plain", 0.1, None, False, 0.0), ("range", 0.1, (10, 30), False, 0.0),
                                              ("range_mask", 0.2, (8, 40), True, 0.0), ("drop", 0.1, None, False, 0.25)]:
        x = logits.clone().requires_grad_(True)
        tg = target.clone()
        if crange is not None:                       # keep targets inside the allowed set, as real constrained tasks do
            tg = torch.where(tg == pad, tg, crange[0] + (tg % (crange[1] - crange[0])))
        cm = None
        if crange is not None:
            cm = torch.ones(x.shape, dtype=torch.bool)
            cm[..., 4:crange[0]] = 0
            cm[..., crange[1]:] = 0
            if use_smask:
                sm = smask.clone()
                sm[torch.arange(rows), tg] = True    # the target itself is always allowed
                cm = torch.logical_and(sm, cm)
                out[name + ".sample_mask"] = sm.numpy().astype(np.uint8)
        xin = x if cm is None else x.masked_fill(~cm, -math.inf)
        lprobs = torch.log_softmax(xin.float(), dim=-1)
        keep = tg != pad
        loss, nll, ntok = label_smoothed_nll_loss(lprobs[keep], tg[keep], eps, update_num=10, drop_worst_ratio=dw,
                                                  drop_worst_after=0, constraint_masks=None if cm is None else cm[keep])
        loss.backward()
        out.update({name + ".target": tg.numpy(), name + ".loss": loss.detach().numpy().reshape(1),
                    name + ".nll": nll.detach().numpy().reshape(1), name + ".ntokens": np.array([ntok]),
                    name + ".dlogits": x.grad.numpy(), name + ".cfg": np.array([eps, -1 if crange is None else crange[0],
                                                                                -1 if crange is None else crange[1], dw])})
    np.savez_compressed(os.path.join(OUT, "ls_cross_entropy.npz"), **out)
    print("ls_cross_entropy ok")


def box_vectors():
    """Integer <bin> indices (bit-exact contract) via the reference's arithmetic
    preprocessor/default/box.py:101-110."""
    g = np.random.Generator(np.random.Philox(key=7))
    coords = g.uniform(0, 512, size=(64, 4)).astype(np.float32)
    max_image_size, num_bins = 512, 1000
    # the reference's own expression (box.py:103-106) on float32 tensors; include exact .5 ties
    coords[0] = [0.0, 512.0, 256.0, 0.2562562]
    coords[1] = np.array([k * 512.0 / 999.0 for k in (0.5, 1.5, 2.5, 3.5)], dtype=np.float32)
    bins = np.array([[int((torch.tensor(c) / max_image_size * (num_bins - 1)).round()) for c in row] for row in coords],
                    dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "box_bins.npz"), coords=coords, bins=bins,
                        max_image_size=np.array([max_image_size]), num_bins=np.array([num_bins]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) == 2 and sys.argv[1] == "--softmax-bwd":
        softmax_bwd_vectors()
        sys.exit(0)
    if len(sys.argv) == 3 and sys.argv[1] == "--case":
        run_case(sys.argv[2], {**CASES, **ORACLE_CASES}[sys.argv[2]])
        sys.exit(0)
    manifest_only = len(sys.argv) == 2 and sys.argv[1] == "--manifest"     # rewrite MANIFEST.json after single --case runs
    # one fresh process per case: the reference's dataclass defaults are shared mutable instances, so adaptor
    # configs (embed_dim, layers, ...) leak from one model to the next inside a process
    import subprocess
    if not manifest_only:
        for name in {**CASES, **ORACLE_CASES}:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--case", name], check=True)
        softmax_vectors()
        softmax_bwd_vectors()
        ls_ce_vectors()
        box_vectors()
    manifest = {
        "generator": "oracle/gen_golden.py",
        "torch": torch.__version__, "numpy": np.__version__,
        "reference_tree_sha1": hashlib.sha1("".join(sorted(
            f"{r}/{f}" for r, _, fs in os.walk("/root/reference/ofasys") for f in fs if f.endswith(".py"))).encode()).hexdigest(),
        "cases": {k: {kk: (sorted(vv) if isinstance(vv, set) else vv) for kk, vv in v.items() if kk != "slots"} for k, v in {**CASES, **ORACLE_CASES}.items()},
    }
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, default=str)
