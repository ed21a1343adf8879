"""GPU: the HIP-backed GeneralistModel (forward + loss + backward) against golden vectors produced by running the
reference (tests/golden/*.npz).  fp32 must match within 1e-3 rel (BASELINE.json north_star); bf16 within 2x the
reference's own bf16-vs-fp32 gap (5.3e-3 of max |logit|, mean rel 1e-2)."""
import numpy as np
import pytest
import torch

from oracle.cases import CASES
from tests.golden_util import case_inputs, drop_keep_rows, load_golden, rel_err, replay_drop_path
from tests.model_util import build_model, make_slots

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3
BF16_TOL = 2e-2
# 2x the reference's own bf16-vs-fp32 gap where that is larger (oracle/ref_bf16_gap.py: 4.7e-2 through the 50-layer
# BatchNorm backbone of tiny_resnet, 3.0e-2 on tiny_video, 1.0e-2 on tiny_text)
# ... and 2.1e-2 through the 24 layers of OFA-large (large_multislot)
# ... 4.2e-2 on tiny_resnet_droppath (same backbone, four images, the reference's recorded stochastic-depth draws replayed)
# ... 1.6e-2 / 1.5e-2 / 1.6e-2 without the layer's extra LayerNorms and head scales (tiny_text_noscale) / with post-LN layers
# (tiny_text_postln) / with one shared rel-pos table, attn_scale_factor 1.5 and no adaptor LayerNorms (tiny_multislot_shared)
BF16_TOL_CASE = {"tiny_resnet": 1e-1, "tiny_resnet_droppath": 1e-1, "tiny_video": 6e-2, "large_multislot": 4.2e-2,
                 "tiny_text_noscale": 3.2e-2, "tiny_text_postln": 3e-2, "tiny_multislot_shared": 3.2e-2}
# fp32 gradients INSIDE the ResNet backbone: 16 bottlenecks of conv / BatchNorm over as few as 32 values per channel /
# ReLU make the backward chain ill-conditioned -- torch's own CPU and GPU (MIOpen) fp32 implementations of this exact
# backbone differ by 0.9% element-wise / 0.06% in norm (tools/bn_noise.py, run on the MI355X box); this build differs
# from the CPU reference by <= 2.4% / 0.1%.  Outputs (logits 4e-6) and every gradient outside the backbone keep 1e-3.
# bf16 gradient NORMS inside the backbone: the reference's own bf16-vs-fp32 gap there (oracle/ref_bf16_grad_gap.py, torch
# CPU) is 41.6% on tiny_resnet and 22.8% on tiny_video (worst parameter: the stem's bn1) -- a one-ulp change of a single
# conv output (e.g. a different split-K plan) moves the stem gradients of THIS build by 10 points as well.  Bound = 1.25x that
# gap; every parameter outside the backbone keeps 2.5 * tol.
BF16_BACKBONE_GRAD_GAP = {"tiny_resnet": 0.416, "tiny_video": 0.228, "tiny_resnet_droppath": 0.297}
FP32_GRAD_TOL_DEEP = 5e-3
FP32_GRAD_ELEM_TOL_DEEP = 4e-2


def _run(name, dtype):
    from ofasys_amd import ops
    case = CASES[name]
    g = load_golden(name)
    model, d = build_model(case, DEV, dtype)
    model.eval()
    if case.get("train"):                      # dropout == 0 in such cases: only BatchNorm changes behaviour
        model.train()
    vals, target = case_inputs(case)
    slots = make_slots(vals, DEV, dtype)
    with replay_drop_path(drop_keep_rows(g)):          # stochastic-depth cases: the reference's recorded per-sample draws
        logits, extra, enc = model(slots, return_encoder_out=True)
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    return g, model, logits, extra, enc, loss


@pytest.mark.parametrize("name", [n for n in CASES if not CASES[n].get("half")])   # (half cases: tests/test_fp16_gpu.py)
def test_fp32_matches_reference(name):
    g, model, logits, extra, enc, loss = _run(name, torch.float32)
    assert logits.shape == tuple(g["logits"].shape)
    print(f"MEASURED {name} fp32: logits {rel_err(logits.detach().cpu(), g['logits']):.2e} (bound {FP32_TOL:.0e}), "
          f"loss {rel_err(loss.detach().cpu(), g['loss'][0]):.2e}")                    # shown by pytest -s; quoted in DESIGN.md section 2
    assert rel_err(logits.detach().cpu(), g["logits"]) < FP32_TOL
    assert rel_err(loss.detach().cpu(), g["loss"][0]) < FP32_TOL
    assert rel_err(extra["attn"][0].cpu(), g["attn"]) < FP32_TOL
    big = CASES[name]["arch"] in ("base", "large")
    e = enc["encoder_out"][0].detach().cpu().contiguous()
    assert rel_err(e.reshape(-1)[::97] if big else e, g["encoder_out"]) < FP32_TOL
    assert np.array_equal(enc["encoder_padding_mask"][0].cpu().numpy().astype(np.uint8), g["encoder_padding_mask"])
    params = dict(model.named_parameters())
    gn = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]))
    scale = max(gn.values())
    for k, want in gn.items():
        if k == "decoder.adaptor.embed_tokens.weight":
            continue
        p = params[k]
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        if want < 0:
            assert got == 0.0, k
        else:
            tol = FP32_GRAD_TOL_DEEP if ".embed_images." in k else FP32_TOL
            assert abs(got - want) <= tol * want + 1e-6 * scale, (k, got, want)
    for k in g:
        if k.startswith("grad."):
            got, want = params[k[5:]].grad.cpu().double(), torch.from_numpy(g[k]).double()
            # (k_proj-like biases have mathematically zero gradient: compare against the global gradient scale too)
            tol = FP32_GRAD_ELEM_TOL_DEEP if ".embed_images." in k else FP32_TOL
            assert float((got - want).abs().max()) <= tol * float(want.abs().max()) + 1e-7 * scale, k
        if k.startswith("buffer."):                # BatchNorm running statistics / step counters after the forward
            assert rel_err(model.state_dict()[k[7:]].double().cpu(), g[k].astype(np.float64)) < FP32_TOL, k
    if "image_rp_bucket_crc" in g:
        import zlib
        b = model.state_dict()["encoder.adaptor.image_resnet.image_rp_bucket"].cpu().contiguous().numpy()
        assert zlib.crc32(b.tobytes()) == int(g["image_rp_bucket_crc"][0])


@pytest.mark.parametrize("name", [n for n in CASES if not CASES[n].get("half")])   # (half cases: tests/test_fp16_gpu.py)
def test_bf16_matches_reference(name):
    g, model, logits, extra, enc, loss = _run(name, torch.bfloat16)
    assert logits.dtype == torch.bfloat16
    tol = BF16_TOL_CASE.get(name, BF16_TOL)
    print(f"MEASURED {name} bf16: logits {rel_err(logits.detach().float().cpu(), g['logits']):.2e} (bound {tol:.1e}), "
          f"loss {rel_err(loss.detach().float().cpu(), g['loss'][0]):.2e}")
    assert rel_err(logits.detach().float().cpu(), g["logits"]) < tol
    assert rel_err(loss.detach().float().cpu(), g["loss"][0]) < tol
    assert rel_err(extra["attn"][0].float().cpu(), g["attn"]) < 2 * tol
    params = dict(model.named_parameters())
    gn = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]))
    scale = max(gn.values())
    bad = []
    for k, want in gn.items():
        if k == "decoder.adaptor.embed_tokens.weight" or want < 0:
            continue
        got = float(params[k].grad.double().norm())
        rel = 2.5 * tol
        if ".embed_images." in k and name in BF16_BACKBONE_GRAD_GAP:
            rel = max(rel, 1.25 * BF16_BACKBONE_GRAD_GAP[name])
        if abs(got - want) > rel * want + 2e-3 * scale:
            bad.append((k, got, want))
    assert not bad, bad[:8]


def _rows(t):
    """reference [B, C, h, w] -> the NHWC rows [B*h*w, C] this build's blocks exchange"""
    B, C, h, w = t.shape
    return torch.from_numpy(np.ascontiguousarray(np.transpose(t, (0, 2, 3, 1)).reshape(B * h * w, C)))


def test_resnet_block_backward_pinned_per_bottleneck():
    """the backward through 16 conv / BatchNorm / ReLU bottlenecks is ill-conditioned as a CHAIN (errors of
    a block are amplified by every BatchNorm in front of it), which is why whole-backbone gradients carry a loose bound.  The
    arithmetic of each block is well-conditioned: feed the REFERENCE's dL/d(output) of a bottleneck (golden `blockgrad.*.dy`)
    into this build's backward of that block alone and compare dL/d(input) with the reference's -- 1e-3, the north-star
    tolerance -- for the last block, the stride-2 blocks with a downsample branch, and the first block.  Then the accumulated
    error of dL/d(output) at every block: 1e-3 through layer3 (the six blocks next to the loss), growing towards the stem."""
    case = CASES["tiny_resnet"]
    g = load_golden("tiny_resnet")
    model, d = build_model(case, DEV, torch.float32)
    model.train()
    backbone = model.encoder.adaptor.image_resnet.embed_images
    io = {}
    hooks = []
    for lname in ("layer1", "layer2", "layer3"):
        for bi, blk in enumerate(getattr(backbone, lname)):
            def keep(m, i, o, key=f"{lname}.{bi}"):
                o[0].retain_grad()
                io[key] = (i[0][0], o[0])
            hooks.append(blk.register_forward_hook(keep))
    from ofasys_amd import ops
    vals, target = case_inputs(case)
    logits = model(make_slots(vals, DEV))[0]
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    local = {}
    for k in case["block_grads"]:
        x, y = io[k]
        dy = _rows(g[f"blockgrad.{k}.dy"]).to(DEV)
        (dx,) = torch.autograd.grad(y, x, dy, retain_graph=True)
        want = _rows(g[f"blockgrad.{k}.dx"])
        local[k] = rel_err(dx.cpu(), want)
    for x, y in io.values():                   # (retain_grad also fires inside torch.autograd.grad: clear what the local runs left)
        x.grad = None
        y.grad = None
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    assert all(e < FP32_TOL for e in local.values()), local
    names = [str(k) for k in g["blockgrad_keys"]]
    acc = {k: abs(float(io[k][1].grad.double().norm()) - w) / w for k, w in zip(names, g["blockgrad_norms"])}
    assert all(acc[k] < FP32_TOL for k in names if k.startswith("layer3.")), acc
    assert all(e < FP32_GRAD_TOL_DEEP for e in acc.values()), acc


def test_resnet_block_backward_bf16_pinned_per_bottleneck():
    """in bf16 the trunk's gradients were only guarded by whole-trunk norm bounds (1.25 x the reference's own
    41.6 % gap) -- a 30 % error in ONE conv gradient would pass.  Per bottleneck, on the checkpointed blocks of the tiny_resnet
    golden, with the reference's recorded dL/d(output) injected (rounded to bf16) into this build's bf16 backward of that block alone:

      matched  dL/d(input) against the ORACLE's fp32 bottleneck (oracle/restate.py:_bottleneck) fed THE SAME bf16 input, bf16-valued
               weights and gradient: only the block's own bf16 arithmetic (conv GEMMs, BatchNorm statistics over 32-512 values per
               channel, ReLU gates) separates the two.  Bound: 2 x what the REFERENCE's bf16 block pays in the same experiment
               (tests/golden/resnet_block_bf16_gap.json, oracle/ref_bf16_block_gap.py: 4-9 % in norm);
      chain    against the golden fp32 dL/d(input) itself (the block input then carries the bf16 forward's drift too): 1.25 x the
               reference's own figure for that.  Measured on MI355X (round 4): matched 0.94 / 1.42 / 1.07 / 0.89 x the reference's own
               figure (layer3.5 / 3.0 / 2.0 / 1.0), chain 0.95 / 0.94 / 0.95 / 0.96 x.
    Printed next to the bounds (`pytest -s`)."""
    import json
    import os
    from oracle import restate
    from oracle.restate import OConfig
    from tests.golden_util import ARCH, ROOT
    gap = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet_block_bf16_gap.json")))["tiny_resnet"]
    case = CASES["tiny_resnet"]
    g = load_golden("tiny_resnet")
    model, d = build_model(case, DEV, torch.bfloat16)
    model.train()
    backbone = model.encoder.adaptor.image_resnet.embed_images
    io, hooks = {}, []
    for lname in ("layer1", "layer2", "layer3"):
        for bi, blk in enumerate(getattr(backbone, lname)):
            def keep(m, i, o, key=f"{lname}.{bi}"):
                io[key] = (i[0][0], o[0], i[0][1:] if len(i[0]) > 1 else None)
            hooks.append(blk.register_forward_hook(keep))
    vals, target = case_inputs(case)
    model(make_slots(vals, DEV, torch.bfloat16))
    for h in hooks:
        h.remove()
    state = {k: (v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    cfg = OConfig(**ARCH["tiny"], resnet_layers=(3, 4, 6), training=True)
    pre = "encoder.adaptor.image_resnet.embed_images."
    for k in case["block_grads"]:
        x, y, _ = io[k]
        dy4 = torch.from_numpy(g[f"blockgrad.{k}.dy"])                         # [B, C, h, w] fp32, the reference's
        dy = _rows(g[f"blockgrad.{k}.dy"]).to(torch.bfloat16).to(DEV)
        (dx,) = torch.autograd.grad(y, x, dy, retain_graph=True)
        torch.cuda.synchronize()
        dx = dx.float().cpu().double()
        # chain: the golden fp32 gradient of the block input
        want = _rows(g[f"blockgrad.{k}.dx"]).double()
        chain = float((dx - want).norm() / want.norm())
        # matched: the oracle's fp32 block on this build's own bf16 input (rows [B*h*w, C] -> [B, C, h, w])
        first = k.endswith(".0")
        B, hi, wi = io[k][2]
        stride = hi // dy4.shape[2]
        x4 = x.detach().float().cpu().view(B, hi, wi, -1).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        y4 = restate._bottleneck(state, pre + k, x4, cfg, stride, first)
        (dx4,) = torch.autograd.grad(y4, x4, dy4.to(torch.bfloat16).float())
        ref = dx4.permute(0, 2, 3, 1).reshape(-1, dx4.shape[1]).double()
        matched = float((dx - ref).norm() / ref.norm())
        print(f"MEASURED bf16 bottleneck {k}: matched {matched:.3e} (reference's own {gap[k]['matched_norm_rel']:.3e}, bound 2x); "
              f"chain {chain:.3e} (reference's own {gap[k]['norm_rel']:.3e}, bound 1.25x)")
        assert matched <= 2.0 * gap[k]["matched_norm_rel"], (k, matched, gap[k])
        assert chain <= 1.25 * gap[k]["norm_rel"], (k, chain, gap[k])


def test_token_bucket_buffer_bit_exact():
    import zlib
    g = load_golden("tiny_text")
    model, _ = build_model(CASES["tiny_text"])
    b = model.encoder.adaptor.text.token_rp_bucket
    assert zlib.crc32(b.contiguous().numpy().tobytes()) == int(g["token_rp_bucket_crc"][0])


def test_get_normalized_probs_and_train_mode():
    from ofasys_amd import ops
    case = CASES["tiny_text"]
    model, d = build_model(case, DEV, torch.bfloat16)
    vals, target = case_inputs(case)
    slots = make_slots(vals, DEV, torch.bfloat16)
    model.eval()
    out = model(slots)
    lp = model.get_normalized_probs(out, log_probs=True)
    assert lp.dtype == torch.float32
    ref = torch.log_softmax(out[0].float(), -1)
    assert rel_err(lp.cpu(), ref.cpu()) < 1e-5
    # train mode: dropout active, loss finite, grads finite, two steps differ (different Philox offsets)
    model.train()
    ops.manual_seed(7)
    l1 = ops.cross_entropy_sum(model(slots)[0], target.to(DEV), d.pad())
    l2 = ops.cross_entropy_sum(model(slots)[0], target.to(DEV), d.pad())
    l1.backward()
    assert torch.isfinite(l1) and torch.isfinite(l2) and float(l1) != float(l2)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    ops.manual_seed(7)
    l1b = ops.cross_entropy_sum(model(slots)[0], target.to(DEV), d.pad())
    assert float(l1b) == float(l1)


@pytest.mark.parametrize("name", ["tiny_text", "base_patch"])
def test_arena_sinks_match_autograd(name):
    """Train-step mode (flat gradient arena, packed k|v|q weights, backward kernels accumulating straight into the
    arena) must produce the same gradients as plain autograd accumulation, and two micro-batches must add up."""
    from ofasys_amd import ops
    from ofasys_amd.trainer import FlatParams
    case = CASES[name]
    vals, target = case_inputs(case)
    ref_model, d = build_model(case, DEV, torch.bfloat16)
    ref_model.eval()
    slots = make_slots(vals, DEV, torch.bfloat16)
    ops.cross_entropy_sum(ref_model(slots)[0], target.to(DEV), d.pad()).backward()
    want = {k: p.grad.float().clone() for k, p in ref_model.named_parameters() if p.grad is not None}
    model, _ = build_model(case, DEV, torch.bfloat16)
    model.eval()
    fp = FlatParams(model)
    packed = [m for m in model.modules() if getattr(m, "_pack", None)]
    assert len(packed) >= case_layers(case)            # every attention got zero-copy packed weights
    for rep in range(2):
        ops.cross_entropy_sum(model(slots)[0], target.to(DEV), d.pad()).backward()
    scale = max(float(v.abs().max()) for v in want.values())
    for k, p in model.named_parameters():
        if k not in want:
            continue
        got = p.grad.float() / 2                       # two identical micro-batches accumulated
        err = float((got - want[k]).abs().max())
        assert err <= 3e-2 * float(want[k].abs().max()) + 2e-3 * scale, (k, err)
    fp.zero_grad()
    assert float(fp.grad.float().abs().sum()) == 0.0


def case_layers(case):
    return {"tiny": 4 + 8, "base": 6 + 12, "large": 12 + 24}[case["arch"]]


def test_graph_replay_matches_eager_steps():
    """The captured (hipGraph) train step must walk the same trajectory as the eager one: fresh dropout masks on every
    replay (device-side Philox position), Adam bias corrections from the device step counter, lr changes picked up."""
    from ofasys_amd import ops
    from ofasys_amd.trainer import Trainer
    case = CASES["tiny_text"]
    vals, target = case_inputs(case)
    runs = []
    for use_graph in (False, True):
        model, d = build_model(case, DEV, torch.bfloat16)
        tr = Trainer(model, lr=1e-3, clip_norm=1.0, use_graph=use_graph, graph_warmup=1)
        slots = make_slots(vals, DEV, torch.bfloat16)
        batch = {"slots": slots, "target": target.to(DEV)}
        ops.manual_seed(123)
        losses = []
        for step in range(6):
            if step == 4:
                tr.lr = 5e-4
            out = tr.train_step([batch])
            losses.append(float(out["stats"][1]))
        torch.cuda.synchronize()
        runs.append((losses, tr.master.clone(), tr))
    (l0, m0, _), (l1, m1, tr1) = runs
    assert tr1.use_graph and any("graphs" in e for e in tr1._graphs.values())      # really captured and replayed
    assert len(set(l1)) == len(l1)                                                    # every replay drew new masks / moved
    assert l0 == l1
    assert torch.equal(m0, m1)


def test_train_step_arithmetic_two_tasks():
    """engine/trainer.py:747-884 + optim/adam.py:192-212 on a 2-task step: gradients of both micro-batches accumulate,
    are multiplied by 1/sum(sample_size), clipped to norm 1 (module/utils.py:342-384), then one Adam update.  The same
    arithmetic is restated on the CPU with the ORACLE model (fp32) and compared with Trainer.train_step (HIP, fp32)."""
    from oracle import restate
    from ofasys_amd.trainer import Trainer
    from tests.golden_util import oracle_cfg, oracle_slots, state_from_golden
    case_a, case_b = CASES["tiny_text"], CASES["tiny_multislot"]
    lr, b1, b2, eps, clip = 1e-2, 0.9, 0.999, 1e-8, 1.0
    # --- oracle side (CPU)
    state = state_from_golden(load_golden("tiny_text"))
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    params = {k: v.requires_grad_(True) for k, v in state.items()
              if v.is_floating_point() and not k.endswith("version") and not k.startswith("decoder.adaptor.embed_tokens")}
    n_total, loss_total = 0, 0.0
    for case in (case_a, case_b):
        vals, target = case_inputs(case)
        logits, _ = restate.model_forward(state, oracle_cfg(case), oracle_slots(vals))
        loss, n = restate.cross_entropy(logits, target)
        loss.backward()
        n_total += n
        loss_total += float(loss)
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) / n_total for k, p in params.items()}
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    coef = min(1.0, clip / (float(gnorm) + 1e-6))
    want = {}
    for k, p in params.items():                      # first Adam step: m = (1-b1) g, v = (1-b2) g^2
        g = grads[k] * coef
        m, v = (1 - b1) * g, (1 - b2) * g * g
        step_size = lr * (1 - b2) ** 0.5 / (1 - b1)
        want[k] = p.detach() - step_size * m / (v.sqrt() + eps)
    # --- HIP side
    model, d = build_model(case_a, DEV, torch.float32)
    tr = Trainer(model, lr=lr, betas=(b1, b2), eps=eps, clip_norm=clip)
    samples = []
    for case in (case_a, case_b):
        vals, target = case_inputs(case)
        samples.append({"slots": make_slots(vals, DEV), "target": target.to(DEV), "task": case["slots"][0][0]})
    model.eval()                                     # dropout off (Trainer switches to train mode: force p = 0 instead)
    for m_ in model.modules():
        if hasattr(m_, "p") and m_.__class__.__name__ == "Dropout":
            m_.p = 0.0
    out = tr.train_step(samples)
    torch.cuda.synchronize()
    assert int(out["stats"][0]) == n_total
    assert abs(float(out["stats"][1]) - loss_total) <= 1e-3 * loss_total
    assert abs(float(out["gnorm"]) - float(gnorm)) <= 2e-3 * float(gnorm)
    got = dict(model.named_parameters())
    gmax = max(float(g.abs().max()) for g in grads.values())
    # (1) the Adam moments are linear / quadratic in the scaled, clipped gradient: well conditioned
    offs = {id(p_): (o, p_.numel()) for p_, o in zip(tr.fp.params, tr.fp.offsets)}
    for k, p_ in got.items():
        if k not in grads or id(p_) not in offs:
            continue
        o, n = offs[id(p_)]
        g = (grads[k] * coef).reshape(-1)
        m_got, v_got = tr.exp_avg[o:o + n].cpu(), tr.exp_avg_sq[o:o + n].cpu()
        assert float((m_got - (1 - b1) * g).abs().max()) <= (1 - b1) * (2e-3 * float(g.abs().max()) + 1e-6 * gmax * coef), k
        assert float((v_got - (1 - b2) * g * g).abs().max()) <= (1 - b2) * (4e-3 * float((g * g).max()) + 1e-9 * (gmax * coef) ** 2), k
    # (2) the parameter update itself where it is well determined (first-step Adam is ~ lr*sign(g): elements whose gradient is
    #     fp32 noise -- e.g. the mathematically zero k_proj.bias gradients -- may legitimately flip sign)
    checked = 0
    for k, w in want.items():
        if k not in got:
            continue
        sel = (grads[k].abs() > 1e-3 * gmax)
        if not bool(sel.any()):
            continue
        upd_w = (w - params[k].detach())[sel]
        upd_g = (got[k].detach().cpu() - params[k].detach())[sel]
        assert float((upd_w - upd_g).abs().max()) <= 2e-2 * lr, k
        checked += int(sel.sum())
    assert checked > 1000


def test_two_graph_replay_path_of_data_parallel_steps():
    """dp_graph="split" (the fallback of the data-parallel step when collectives cannot be captured): TWO graphs
    (forward+backward, clip+Adam) around an eager all-reduce.  There is one GPU here, so the path is driven with the world size
    forced to 2 (the reducer itself stays single-rank): the trajectory must equal the eager one.  The default, one graph with
    the RCCL collectives captured, is covered by tests/test_configs_gpu.py::test_dp_step_graph_with_captured_rccl_collectives."""
    from ofasys_amd import ops
    from ofasys_amd.trainer import Trainer
    case = CASES["tiny_multislot"]
    vals, target = case_inputs(case)
    runs = []
    for two_graphs in (False, True):
        model, d = build_model(case, DEV, torch.bfloat16)
        tr = Trainer(model, lr=1e-3, clip_norm=1.0, use_graph=two_graphs, graph_warmup=1, dp_graph="split")
        if two_graphs:
            tr.world = 2
        batch = {"slots": make_slots(vals, DEV, torch.bfloat16), "target": target.to(DEV)}
        ops.manual_seed(5)
        losses = [float(tr.train_step([batch])["stats"][1]) for _ in range(5)]
        torch.cuda.synchronize()
        runs.append((losses, tr.master.clone(), tr))
    (l0, m0, _), (l1, m1, tr1) = runs
    entry = [e for e in tr1._graphs.values() if "graphs" in e]
    assert entry and len(entry[0]["graphs"]) == 2
    assert l0 == l1 and torch.equal(m0, m1)


def test_sharded_optimizer_step_on_one_rank_follows_the_default_step():
    """TrainStep(shard_optimizer=True) (reduce-scatter -> owners update their pieces of the arena -> all-gather; multi-rank arithmetic:
    tests/test_host_logic_cpu.py, 2 and 4 gloo ranks) on the one GPU there is: the rank owns every piece, so the step runs the
    piecewise sum of squares / Adam launches over the bucket pieces and tails and must follow the default step -- same losses, the
    master parameters within the fp32 rounding of a norm summed piece by piece -- eagerly and as a replayed graph."""
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    case = CASES["tiny_multislot"]
    vals, target = case_inputs(case)
    runs = []
    for shard, graph in ((False, False), (True, False), (True, True)):
        model, d = build_model(case, DEV, torch.bfloat16)
        tr = TrainStep(model, lr=1e-3, clip_norm=1.0, use_graph=graph, graph_warmup=1, shard_optimizer=shard, bucket_bytes=1 << 16)
        if shard:
            assert len(tr.reducer.buckets) > 3 and len(tr.reducer.owned_ranges()) >= len(tr.reducer.buckets)
            spans = sorted((lo, hi) for lo, hi, _ in tr.reducer.owned_ranges())
            assert spans[0][0] == 0 and spans[-1][1] == tr.fp.numel and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        batch = {"slots": make_slots(vals, DEV, torch.bfloat16), "target": target.to(DEV)}
        ops.manual_seed(5)
        losses, early = [], None
        for step in range(5):
            losses.append(float(tr.train_step([batch])["stats"][1]))
            if step == 1:
                early = tr.master.clone()
        torch.cuda.synchronize()
        runs.append((losses, tr.master.clone(), tr.fp.flat.float().clone(), early))
    (l0, m0, p0, e0), (l1, m1, p1, e1), (l2, m2, p2, e2) = runs
    assert l1 == l2 and torch.equal(m1, m2) and torch.equal(p1, p2)                 # eager == replayed graph, bit for bit
    # the two forms differ by the fp32 rounding of the squared norm (one sum over the arena / a sum of per-piece sums): the first update
    # is identical, the second within an ulp of the clip coefficient; after that this tiny model at lr = 1e-3 amplifies the last bit
    # (measured: 5e-4 after three updates, 0.2 % of the fourth loss) -- the later steps are held to "the same trajectory", not to bits
    assert l0[:3] == l1[:3]
    assert float((e0 - e1).abs().max()) <= 1e-6
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-2 * abs(a)
    assert float((m0 - m1).abs().max()) <= 5 * 1e-3                                 # five Adam steps of <= lr each


# ------------------------------------------------------------------------------------------------ incremental decoding
def _incremental_hip(dtype):
    """The scenario of oracle/incremental_case.py through the HIP model: encoder once, beams, KV-cache steps, reorder."""
    from oracle.cases import VOCAB_EXTRA
    from oracle.incremental_case import BEAM_ORDER, NEW_ORDER, REORDER_AT, STEPS, beam_prefix
    from ofasys_amd import ModalityType, Slot
    case = CASES["tiny_text"]
    model, d = build_model(case, DEV, dtype)
    model.eval()
    vals, _ = case_inputs(case)
    src = [s for s in make_slots(vals, DEV, dtype) if s.is_src]
    prev = beam_prefix(4 + VOCAB_EXTRA).to(DEV)
    with torch.no_grad():
        enc = model.encoder(src)
        enc = model.encoder.reorder_encoder_out(enc, torch.tensor(BEAM_ORDER, device=DEV))
        inc, logits, extra = {}, [], None
        for t in range(STEPS):
            out, extra = model.decoder([Slot(ModalityType.TEXT, False, prev[:, :t + 1])], encoder_out=enc, incremental_state=inc)
            assert out.shape[1] == 1
            logits.append(out[:, -1].float().clone())
            if t == REORDER_AT:
                order = torch.tensor(NEW_ORDER, device=DEV)
                model.decoder.reorder_incremental_state_scripting(inc, order)
                enc = model.encoder.reorder_encoder_out(enc, order)
                prev = prev.index_select(0, order)
        full, _ = model.decoder([Slot(ModalityType.TEXT, False, prev)], encoder_out=enc)
        buf = model.decoder.layers[0].self_attn._get_input_buffer(inc)
    torch.cuda.synchronize()
    return torch.stack(logits).cpu(), extra["attn"][0].float().cpu(), full[:, -1].float().cpu(), buf


def test_incremental_decoding_fp32_matches_reference():
    """KV-cache decoding with the decode attention kernel (csrc/attention_decode.hip) + beam reorder against the reference
    run step by step (oracle/gen_incremental_golden.py)."""
    gi = load_golden("tiny_text_incremental")
    logits, attn, full_last, buf = _incremental_hip(torch.float32)
    assert rel_err(logits, gi["logits"]) < FP32_TOL
    assert rel_err(attn, gi["attn"]) < FP32_TOL
    assert rel_err(full_last, gi["full_last"]) < FP32_TOL
    assert rel_err(logits[-1], full_last) < FP32_TOL
    # the cache in the reference's form: (bsz, heads, len, head_dim)
    assert tuple(buf["prev_key"].shape) == tuple(gi["prev_key_l0"].shape)
    assert rel_err(buf["prev_key"].cpu(), gi["prev_key_l0"]) < FP32_TOL
    assert rel_err(buf["prev_value"].cpu(), gi["prev_value_l0"]) < FP32_TOL


def test_incremental_decoding_bf16_matches_reference():
    gi = load_golden("tiny_text_incremental")
    logits, attn, full_last, _ = _incremental_hip(torch.bfloat16)
    assert rel_err(logits, gi["logits"]) < BF16_TOL
    assert rel_err(attn, gi["attn"]) < BF16_TOL
    assert rel_err(logits[-1], full_last) < BF16_TOL


def test_incremental_equals_teacher_forcing_at_base_size():
    """Size-independent property at the cfg-2 decoder size (OFA-base, bf16, 448 source positions): the logits of step t
    from the KV cache equal row t of the teacher-forced decoder pass."""
    import bench
    import argparse
    from ofasys_amd import ModalityType, Slot
    args = argparse.Namespace(arch="base", workload="cfg2", batch=4)
    model, d = bench.build(args, torch.device(DEV))
    model.eval()
    batch, _, _ = bench.make_batch(d, 4, 191, 24, 0, torch.device(DEV), "cfg2")
    src = [s for s in batch["slots"] if s.is_src]
    prev = batch["slots"][-1].value
    with torch.no_grad():
        enc = model.encoder(src)
        full, _ = model.decoder([Slot(ModalityType.TEXT, False, prev)], encoder_out=enc)
        inc = {}
        worst = 0.0
        for t in range(prev.shape[1]):
            out, _ = model.decoder([Slot(ModalityType.TEXT, False, prev[:, :t + 1])], encoder_out=enc, incremental_state=inc)
            # rows whose prefix already contains padding are not comparable position by position
            live = prev[:, t].ne(d.pad())
            if live.any():
                worst = max(worst, rel_err(out[live, -1].float().cpu(), full[live, t].float().cpu()))
    assert worst < BF16_TOL, worst


def test_maximum_positions_match_oracle():
    """Edge of the position tables: 1024 source and 1024 target tokens (max_source/target_positions, bucket tables of
    2*256-1 relative distances) through the tiny model, fp32, against the CPU oracle on the same recipe weights."""
    from oracle import recipe, restate
    from oracle.cases import VOCAB_EXTRA
    from oracle.restate import OSlot
    from tests.golden_util import oracle_cfg, state_from_golden
    from ofasys_amd import ModalityType, Slot
    torch.set_num_threads(8)
    case = CASES["tiny_text"]
    V = 4 + VOCAB_EXTRA
    src = recipe.tokens("input.max_src", (1, 1024), V, [1024])
    prev = recipe.tokens("input.max_prev", (1, 1024), V, [1024], bos=0)
    model, d = build_model(case, DEV, torch.float32)
    model.eval()
    with torch.no_grad():
        logits, extra = model([Slot(ModalityType.TEXT, True, src.to(DEV)), Slot(ModalityType.TEXT, False, prev.to(DEV))])
    state = state_from_golden(load_golden("tiny_text"))
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    with torch.no_grad():
        ref, rextra = restate.model_forward(state, oracle_cfg(case), [OSlot("TEXT", True, src, None), OSlot("TEXT", False, prev, None)])
    assert rel_err(logits.cpu(), ref) < FP32_TOL
    assert rel_err(extra["attn"][0].cpu(), rextra["attn"]) < FP32_TOL
    # one position more than the tables hold is refused, not wrapped
    too_long = torch.cat([src, src[:, :1]], 1).to(DEV)
    with pytest.raises(Exception):
        with torch.no_grad():
            model([Slot(ModalityType.TEXT, True, too_long), Slot(ModalityType.TEXT, False, prev.to(DEV))])
        torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Ts,Tt,slen,tlen", __import__("oracle.edge_cases", fromlist=["SHAPES"]).SHAPES)
def test_ragged_and_degenerate_shapes_match_oracle(B, Ts, Tt, slen, tlen, dtype):
    """Edge shapes of the text path (tiny model, biased attention; oracle/edge_cases.py: one token, a batch of one at odd lengths,
    single-token rows, an almost entirely padded row): logits, loss and every gradient norm against the CPU oracle run on the same
    recipe weights and inputs -- the oracle itself is pinned to the reference on exactly these inputs
    (tests/test_oracle_golden.py::test_edge_shapes_match_reference)."""
    from oracle import recipe, restate
    from oracle.cases import VOCAB_EXTRA, make_target
    from oracle.restate import OSlot
    from tests.golden_util import oracle_cfg, state_from_golden
    from ofasys_amd import ModalityType, Slot, ops
    torch.set_num_threads(8)
    case = CASES["tiny_text"]
    V = 4 + VOCAB_EXTRA
    src = recipe.tokens(f"input.edge_src{B}{Ts}", (B, Ts), V, slen)
    prev = recipe.tokens(f"input.edge_prev{B}{Tt}", (B, Tt), V, tlen, bos=0)
    target = make_target(prev)
    model, d = build_model(case, DEV, dtype)
    model.eval()
    logits, extra = model([Slot(ModalityType.TEXT, True, src.to(DEV)), Slot(ModalityType.TEXT, False, prev.to(DEV))])
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    state = state_from_golden(load_golden("tiny_text"))
    params = {k: v.requires_grad_(True) for k, v in state.items() if v.is_floating_point() and not k.endswith("version")}
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    ref, _ = restate.model_forward(state, oracle_cfg(case), [OSlot("TEXT", True, src, None), OSlot("TEXT", False, prev, None)])
    rloss, n = restate.cross_entropy(ref, target)
    rloss.backward()
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    live = target.ne(d.pad())                                         # rows of padded targets are not part of the contract
    assert rel_err(logits.detach().float().cpu()[live], ref.detach()[live]) < tol
    assert rel_err(loss.detach().float().cpu(), rloss.detach()) < tol
    mine = dict(model.named_parameters())
    scale = max(float(p.grad.norm()) for p in params.values() if p.grad is not None)
    for k, p in params.items():
        if k == "decoder.adaptor.embed_tokens.weight" or p.grad is None:
            continue
        want = float(p.grad.double().norm())
        got = float(mine[k].grad.double().norm()) if mine[k].grad is not None else 0.0
        assert abs(got - want) <= 2.5 * tol * want + 2e-3 * scale, (k, got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batch_of_one_image_patch_base_matches_oracle(dtype):
    """cfg-2 family at batch 1 (OFA-base, image_patch_embed + text -> text): the packed projections, the fused attention
    and the 192x256 / 128x128 GEMM plans at their smallest row counts, against the CPU oracle."""
    from oracle import recipe, restate
    from oracle.cases import VOCAB_EXTRA, make_target
    from oracle.restate import OSlot
    from tests.golden_util import oracle_cfg, state_from_golden
    from ofasys_amd import ModalityType, Slot, ops
    torch.set_num_threads(8)
    case = CASES["base_patch"]
    V = 4 + VOCAB_EXTRA
    img = recipe.floats("input.b1_image", (1, 3, 224, 224))
    src = recipe.tokens("input.b1_src", (1, 7), V, [7])
    prev = recipe.tokens("input.b1_prev", (1, 5), V, [5], bos=0)
    target = make_target(prev)
    attrs = ["adaptor=image_patch_embed"]
    model, d = build_model(case, DEV, dtype)
    model.eval()
    logits, _ = model([Slot(ModalityType.IMAGE, True, img.to(DEV).to(dtype), attributes=attrs),
                       Slot(ModalityType.TEXT, True, src.to(DEV)), Slot(ModalityType.TEXT, False, prev.to(DEV))])
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    loss.backward()
    torch.cuda.synchronize()
    state = state_from_golden(load_golden("base_patch"))
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    with torch.no_grad():
        ref, _ = restate.model_forward(state, oracle_cfg(case), [OSlot("IMAGE", True, img, attrs), OSlot("TEXT", True, src, None),
                                                               OSlot("TEXT", False, prev, None)])
        rloss, _ = restate.cross_entropy(ref, target)
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    assert rel_err(logits.detach().float().cpu(), ref) < tol
    assert rel_err(loss.detach().float().cpu(), rloss) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_step_decoder_graph_replay_equals_eager(dtype):
    """ofasys_amd.generator.StepDecoder: decoding through the per-length hipGraphs (sequence 0 runs eagerly, sequence 1
    records, sequence 2 replays) must give exactly the eager logits, for new batches of the same shape and across an
    in-place beam reorder."""
    from oracle import recipe
    from oracle.cases import VOCAB_EXTRA
    from ofasys_amd import ModalityType, Slot
    from ofasys_amd.generator import StepDecoder
    case = CASES["tiny_text"]
    V = 4 + VOCAB_EXTRA
    model, d = build_model(case, DEV, dtype)
    model.eval()
    steps, order = 6, torch.tensor([0, 0, 1, 1], device=DEV)
    new_order = torch.tensor([1, 0, 3, 3], device=DEV)

    def run(dec, seed):
        src = recipe.tokens(f"input.sd_src{seed}", (2, 9), V, [9, 6]).to(DEV)
        dec.begin([Slot(ModalityType.TEXT, True, src)], beam_order=order)
        nxt = torch.full((4,), d.bos(), dtype=torch.long, device=DEV)
        out = []
        for t in range(steps):
            logits = dec.step(nxt).float().clone()
            out.append(logits)
            nxt = logits.argmax(-1) + torch.tensor([0, 1, 0, 2], device=DEV)      # distinct beams
            nxt = nxt.clamp(max=V - 1)
            if t == 2:
                dec.reorder(new_order)
                nxt = nxt.index_select(0, new_order)
        return torch.stack(out)

    eager, graph = StepDecoder(model, 16, use_graph=False), StepDecoder(model, 16, use_graph=True)
    for seed in range(3):                                   # graph decoder: eager warm-up, capture, replay
        a, b = run(eager, seed), run(graph, seed)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (seed, float((a - b).abs().max()))
    assert len(graph._graphs) == steps


def test_drop_path_model_trains_and_is_identity_in_eval():
    """encode_drop_path_rate > 0 (module/droppath.py, model/transformer.py:58-59, 249-252): the layers leave the fused
    residual joins for the op-by-op order; eval mode is unaffected, train mode gives finite loss and gradients."""
    import copy
    from ofasys_amd import ops
    case = copy.deepcopy(CASES["tiny_text"])
    g = load_golden("tiny_text")
    case["overrides"] = dict(case["overrides"], encode_drop_path_rate=0.3, dropout=0.0)
    model, d = build_model(case, DEV, torch.float32)
    assert model.encoder.layers[-1].drop_path.drop_prob > 0 and model.decoder.layers[-1].drop_path.drop_prob > 0
    vals, target = case_inputs(case)
    model.eval()
    logits, _ = model(make_slots(vals, DEV))
    assert rel_err(logits.detach().cpu(), g["logits"]) < FP32_TOL          # identity in eval
    model.train()
    torch.manual_seed(0)
    logits, _ = model(make_slots(vals, DEV))
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    loss.backward()
    assert torch.isfinite(loss) and rel_err(logits.detach().cpu(), g["logits"]) > 1e-3     # paths were dropped
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layerdrop_keeps_the_reference_s_survivor_semantics(dtype):
    """encoder_layerdrop / decoder_layerdrop (module/layer_drop.py:13-41, model/transformer.py:53-54, 244-245): in training each layer
    survives with probability 1 - p, one uniform CPU draw per layer, and the stacks enumerate the SURVIVORS (a survivor's index picks
    its per-layer position bias).  With the draws replayed from the same seed, the model must equal a copy whose layer lists were cut
    down to the survivors by hand; evaluation keeps every layer; gradients reach the survivors only."""
    import copy
    from ofasys_amd import ops
    case = copy.deepcopy(CASES["tiny_text"])
    case["overrides"] = dict(case["overrides"], encoder_layerdrop=0.5, decoder_layerdrop=0.4, dropout=0.0)
    model, d = build_model(case, DEV, dtype)
    assert model.encoder.layerdrop == 0.5 and model.decoder.layerdrop == 0.4
    vals, target = case_inputs(case)
    g = load_golden("tiny_text")
    model.eval()
    if dtype == torch.float32:
        assert rel_err(model(make_slots(vals, DEV, dtype))[0].detach().cpu(), g["logits"]) < FP32_TOL      # eval: all layers
    seed = 11
    torch.manual_seed(seed)
    ne, nd = len(model.encoder.layers), len(model.decoder.layers)
    keep_e = [u > 0.5 for u in torch.empty(ne).uniform_().tolist()]
    keep_d = [u > 0.4 for u in torch.empty(nd).uniform_().tolist()]
    assert 0 < sum(keep_e) + sum(keep_d) < ne + nd                                                         # (something is dropped)
    ref = copy.deepcopy(model)
    ref.encoder.layers = torch.nn.ModuleList([l for l, k in zip(ref.encoder.layers, keep_e) if k])
    ref.decoder.layers = torch.nn.ModuleList([l for l, k in zip(ref.decoder.layers, keep_d) if k])
    ref.encoder.layerdrop = ref.decoder.layerdrop = 0.0
    model.train()
    ref.train()
    torch.manual_seed(seed)
    logits, _ = model(make_slots(vals, DEV, dtype))
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    loss.backward()
    rl, _ = ref(make_slots(vals, DEV, dtype))
    assert torch.equal(logits, rl)
    dropped = [f"encoder.layers.{i}." for i, k in enumerate(keep_e) if not k] + [f"decoder.layers.{i}." for i, k in enumerate(keep_d) if not k]
    for k, p in model.named_parameters():
        if any(k.startswith(pre) for pre in dropped):
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
        elif p.grad is not None:
            assert torch.isfinite(p.grad).all(), k


@pytest.mark.parametrize("what", ["freeze_encoder", "freeze_embedding", "freeze_resnet"])
def test_frozen_parameters_take_no_gradient_and_nothing_else_changes(what):
    """freeze_encoder (model/ofa.py:382-383: encoder.requires_grad_(False) -- the tied embedding is a child of the encoder's adaptor,
    so it freezes for the decoder too), freeze_encoder_embedding == freeze_decoder_embedding (adaptor/general.py:211, 218-219) and
    image_resnet.freeze_resnet (adaptor/image_resnet.py:107-114: BatchNorm in eval mode, affine parameters frozen): the forward and
    every remaining gradient equal the fixture of the unfrozen model; frozen tensors take no gradient and a train step (eager and
    replayed graph) leaves them bit-identical while the others move."""
    import copy
    from ofasys_amd import ops
    from ofasys_amd.trainer import Trainer
    name = "tiny_resnet" if what == "freeze_resnet" else "tiny_text"
    case = copy.deepcopy(CASES[name])
    if what == "freeze_encoder":
        case["overrides"] = dict(case["overrides"], freeze_encoder=True)
    elif what == "freeze_embedding":
        case["overrides"] = dict(case["overrides"], freeze_encoder_embedding=True, freeze_decoder_embedding=True)
    else:
        case["adaptor_overrides"] = {"image_resnet": dict(case["adaptor_overrides"]["image_resnet"], freeze_resnet=True)}
    g = load_golden(name)
    model, d = build_model(case, DEV, torch.float32)
    params = dict(model.named_parameters())
    if what == "freeze_encoder":
        frozen = {k for k in params if k.startswith("encoder.")} | {"decoder.adaptor.embed_tokens.weight"}
    elif what == "freeze_embedding":
        frozen = {"encoder.adaptor.embed_tokens.weight", "decoder.adaptor.embed_tokens.weight"}
    else:
        model.train()                                   # the adaptor's train() is what puts the trunk's BatchNorm into eval mode
        frozen = {k for k, p in params.items() if not p.requires_grad}
        assert frozen and all(".embed_images." in k and (".bn" in k or ".downsample.1." in k) for k in frozen), sorted(frozen)[:4]
    assert {k for k, p in params.items() if not p.requires_grad} == {k for k in frozen if k in params}
    vals, target = case_inputs(case)
    if what != "freeze_resnet":                         # (a frozen trunk normalises with its running statistics: no fixture for that)
        model.eval()
        logits, _ = model(make_slots(vals, DEV))
        loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
        loss.backward()
        assert rel_err(logits.detach().cpu(), g["logits"]) < FP32_TOL
        gn = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]))
        scale = max(gn.values())
        for k, p in params.items():
            if k in frozen:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            elif gn[k] >= 0:
                got = float(p.grad.double().norm())
                assert abs(got - gn[k]) <= FP32_TOL * gn[k] + 1e-6 * scale, (k, got, gn[k])
        model.zero_grad(set_to_none=True)
    # the update loop: frozen tensors stay bit-identical, the rest moves; the replayed graph walks the eager trajectory
    unused_of = {str(k): v < 0 for k, v in zip(g["grad_norm_keys"], g["grad_norms"])}
    del model, params
    runs = []
    for use_graph in (False, True):
        model, d = build_model(case, DEV, torch.float32)
        model.train()
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        buffers0 = {k: v.detach().clone() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        tr = Trainer(model, lr=1e-3, clip_norm=1.0, use_graph=use_graph, graph_warmup=1)
        batch = {"slots": make_slots(vals, DEV), "target": target.to(DEV)}
        ops.manual_seed(5)
        losses = [float(tr.train_step([batch])["stats"][1]) for _ in range(4)]
        torch.cuda.synchronize()
        assert all(v == v and v < 1e6 for v in losses), losses
        assert not use_graph or any("graphs" in e for e in tr._graphs.values())
        for k, p in model.named_parameters():
            same = torch.equal(p.detach(), before[k])
            unused = unused_of[k]                       # e.g. the decoder adaptor's type embedding (source slots only): zero gradient
            assert same == (k in frozen or unused), (k, same)
        if what == "freeze_resnet":                     # eval-mode BatchNorm: the running statistics do not move either
            sd = model.state_dict()
            assert buffers0 and all(torch.equal(sd[k], v) for k, v in buffers0.items())
        runs.append(losses)
        del tr, model
    assert runs[0] == runs[1]


def test_overfit_one_batch_loss_goes_down():
    """End-to-end sanity of the whole update loop (fused backward kernels + gradient arena + clip + Adam, captured step):
    40 steps on one tiny batch with dropout on must drive the per-token loss well below its starting value, with
    finite gradient norms throughout."""
    from ofasys_amd import ops
    from ofasys_amd.trainer import Trainer
    case = CASES["tiny_text"]
    vals, target = case_inputs(case)
    model, d = build_model(case, DEV, torch.bfloat16)
    tr = Trainer(model, lr=2e-3, clip_norm=1.0, use_graph=True, graph_warmup=1)
    batch = {"slots": make_slots(vals, DEV, torch.bfloat16), "target": target.to(DEV)}
    ops.manual_seed(7)
    losses, gnorms = [], []
    for _ in range(40):
        out = tr.train_step([batch])
        losses.append(float(out["stats"][1]) / float(out["stats"][0]))
        gnorms.append(float(out["gnorm"]))
    assert all(map(lambda v: v == v and v < 1e6, losses + gnorms)), (losses[-5:], gnorms[-5:])
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert sum(losses[-5:]) < sum(losses[:5])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_incremental_decoding_with_finished_beams_matches_reference(dtype):
    """Pad tokens inside the prefix (finished hypotheses): the cached key-padding mask reaches ofa_attn_decode on every
    later step; logits against the reference run step by step, the mask bit-exact in the reference's form."""
    from oracle.cases import VOCAB_EXTRA
    from oracle.incremental_case import BEAM_ORDER, STEPS, padded_prefix
    from ofasys_amd import ModalityType, Slot
    case = CASES["tiny_text"]
    gi = load_golden("tiny_text_incremental")
    model, d = build_model(case, DEV, dtype)
    model.eval()
    vals, _ = case_inputs(case)
    src = [s for s in make_slots(vals, DEV, dtype) if s.is_src]
    prev = padded_prefix(4 + VOCAB_EXTRA).to(DEV)
    with torch.no_grad():
        enc = model.encoder.reorder_encoder_out(model.encoder(src), torch.tensor(BEAM_ORDER, device=DEV))
        inc, logits = {}, []
        for t in range(STEPS):
            out, _ = model.decoder([Slot(ModalityType.TEXT, False, prev[:, :t + 1])], encoder_out=enc, incremental_state=inc)
            logits.append(out[:, -1].float().clone())
        buf = model.decoder.layers[1].self_attn._get_input_buffer(inc)
    assert rel_err(torch.stack(logits).cpu(), gi["logits_padded"]) < (FP32_TOL if dtype == torch.float32 else BF16_TOL)
    assert np.array_equal(buf["prev_key_padding_mask"].float().cpu().numpy(), gi["kpm_padded_l1"])
