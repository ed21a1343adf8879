"""GPU: parity AT THE BENCHMARKED SIZES (VERDICT r2 "what's weak" 2-3): the models `bench.py` measures, built by bench.build /
bench.make_batch themselves, stepped the way the bench steps them (ragged row packing, replayed hipGraph) and compared with the CPU
oracle (oracle/restate.py, pinned to the reference by tests/golden/*) on the SAME weights and inputs:

  cfg-2   base, image_patch_embed 257 + text <= 191 -> text <= 64, bf16, packed rows at bucket 512 / 256, one hipGraph;
          batch 8 instead of 32 bounds the CPU leg (~4 s) -- every kernel sees the bench's row buckets and tile plans
  cfg-4   base, video 8 x 224 x 224 -> 1568 + 32 positions, batch 1, fp32 (1e-3 tier) and bf16
  cfg-5   OFA-large with an image slot through the DEFAULT image adaptor (image_resnet, resnet152) -- large had only run text
  cfg-3   the two-task step with resnet101 (the BASELINE cfg-2b / cfg-3 trunk) instead of resnet50

Tolerances: fp32 1e-3 (north_star) outside the ResNet trunk, the trunk's documented bounds inside (tests/test_model_gpu.py);
bf16 2e-2 of max |logit| (2x the reference's own bf16-vs-fp32 gap) and 2.5x that on gradient norms; the cfg-2
step (same bf16 weights on both sides) is held to 2x / 4x what it measured: logits 1.2e-2, gradient norms 1e-2, loss 1e-4."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import recipe, restate  # noqa: E402
from oracle.cases import VOCAB_EXTRA, make_target  # noqa: E402
from oracle.restate import OConfig, OSlot  # noqa: E402
from tests.golden_util import ARCH, oracle_params, oracle_slots, oracle_state_for, rel_err  # noqa: E402
from tests.model_util import build_model, make_slots  # noqa: E402
from tests.test_configs_gpu import _arena_grads, _check_grads, _oracle_step, _tok  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")]
DEV = "cuda"
BF16_TOL = 2e-2
# cfg-2, the benchmarked step itself: the oracle runs on the model's OWN bf16 weights and inputs in fp32 arithmetic, so what is left
# is the bf16 rounding of the activations between kernels and the MFMA accumulation order.  Measured on MI355X
# (profiles/round3_parity_measured.txt): logits 6.1e-3 of max |logit|, loss 6.9e-6, worst gradient norm 2.4e-3, clip norm 4.7e-4.
# Bounds = 2x (logits) / 4x (gradient norms: 2.5 * CFG2_GRAD_TOL = 1e-2) what was measured -- not the generic bf16 tier's 2e-2 / 5e-2.
CFG2_LOGIT_TOL = 1.2e-2
CFG2_GRAD_TOL = 4e-3


def _state_from_model(model):
    """The oracle's state = the model's OWN parameters and buffers (bf16 values, held in fp32), reference key schema."""
    state = {}
    for k, v in model.state_dict().items():
        state[k] = v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    return state


def _measured(what, value):
    """`pytest -s` shows what a run actually measured next to the bound it is held to (DESIGN.md section 2 quotes these)."""
    print(f"MEASURED {what}: {value:.3e}")


def _bf16_grad_check(got, want, tol, what=""):
    scale = max(float(g.double().norm()) for g in want.values() if g is not None)
    bad, checked, worst = [], 0, 0.0
    for k, w in want.items():
        if k not in got or w is None:
            continue
        g, wn = float(got[k].double().norm()), float(w.double().norm())
        if wn > 1e-2 * scale:
            worst = max(worst, abs(g - wn) / wn)
        if abs(g - wn) > 2.5 * tol * wn + 2e-3 * scale:
            bad.append((k, g, wn))
        checked += 1
    _measured(f"{what} worst gradient-norm deviation (parameters above 1% of the largest norm, {checked} checked)", worst)
    assert checked > 100 and not bad, bad[:8]


@pytest.mark.parametrize("B", [8, 32])
def test_cfg2_benchmarked_step_packed_graph_vs_oracle(B):
    """bench.py's headline configuration, exactly as the bench builds and steps it, against the oracle.  B = 32 is the benchmarked
    batch itself: its 13 312-row encoder bucket, the 216-workgroup grouped weight-gradient launches and the eight-wave 192 x 256 /
    256 x 256 tile plans; the oracle walks it as four shards of 8 rows (the loss and the gradients of a batch
    are sums over its rows), which bounds the CPU leg to ~15 s."""
    import bench
    from ofasys_amd.trainer import TrainStep
    args = SimpleNamespace(arch="base", workload="cfg2", dtype="bf16", dropout=0.0)      # dropout 0: the oracle has none
    bench._HALF_NOW[0] = torch.bfloat16
    model, d = bench.build(args, torch.device(DEV))
    sample, ntok, (slens, tlens) = bench.make_batch(d, B, 191, 64, 0, torch.device(DEV), "cfg2", pack=True)
    plan = sample["pack"]
    assert plan.enc_index.numel() % 512 == 0 and plan.dec_index.numel() % 256 == 0          # the bench's row buckets
    # the oracle on the same weights (bf16 values in fp32 arithmetic) and the same inputs, padded as the reference pads
    state = _state_from_model(model)
    cfg = OConfig(**ARCH["base"], use_self_attn_bias=False, entangle_position_embedding=True,
                  adaptor_entangle={"text": True, "image_patch_embed": True})
    img, src, prev = (s.value.detach().cpu() for s in sample["slots"])
    target = sample["target"].cpu()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    params = oracle_params(state)
    ref_loss, n, shards = 0.0, 0, []
    for r0 in range(0, B, 8):                                    # (gradients accumulate in the parameters' .grad across the shards)
        rows = slice(r0, r0 + 8)
        oslots = [OSlot("IMAGE", True, img[rows].float(), ["adaptor=image_patch_embed"]), OSlot("TEXT", True, src[rows]),
                  OSlot("TEXT", False, prev[rows])]
        lg, _ = restate.model_forward(state, cfg, oslots)
        ls, ns = restate.cross_entropy(lg, target[rows])
        ls.backward()
        ref_loss, n = ref_loss + float(ls), n + int(ns)
        shards.append(lg.detach())
    want = {k: (None if p.grad is None else p.grad.detach()) for k, p in params.items()}
    ref_logits = torch.cat(shards, 0)

    # 1) the step: three train_steps -> the third one is a REPLAY of the captured graph (lr 0: same weights, same gradients)
    tr = TrainStep(model, lr=0.0, clip_norm=0.0, use_graph=True, graph_warmup=1)
    for _ in range(3):
        out = tr.train_step([sample])
    torch.cuda.synchronize()
    assert any("graphs" in e for e in tr._graphs.values()), "the step was not captured"
    assert int(out["stats"][0]) == n == sum(tlens)
    assert abs(float(out["stats"][1]) - float(ref_loss)) <= 1e-4 * float(ref_loss)          # measured 6.9e-6 (profiles/round3_parity_measured.txt)
    got = _arena_grads(tr, model)
    _measured(f"cfg-2 packed graph step (B = {B}): loss deviation", abs(float(out["stats"][1]) - float(ref_loss)) / float(ref_loss))
    _bf16_grad_check(got, want, CFG2_GRAD_TOL, f"cfg-2 packed graph step (B = {B}):")
    gn = np.sqrt(sum(float(g.double().pow(2).sum()) for g in want.values() if g is not None)) / n
    _measured(f"cfg-2 packed graph step (B = {B}): clip-norm deviation", abs(float(out["gnorm"]) - gn) / gn)
    assert abs(float(out["gnorm"]) - gn) <= 2e-3 * gn                                        # measured 4.7e-4
    # 2) the logits of the packed forward at every non-pad decoder position
    model.train()
    with torch.no_grad():
        logits = model(sample["slots"], pack=plan)[0].float().cpu()                           # [1, dec rows, V]
    idx = plan.dec_index.cpu()
    rows = torch.nonzero(idx >= 0).squeeze(1)
    assert rows.numel() == sum(tlens)
    ref_rows = ref_logits.reshape(-1, ref_logits.shape[-1])[idx[rows]]
    _measured(f"cfg-2 packed forward (B = {B}): logits max |diff| / max |logit|", rel_err(logits[0, rows], ref_rows))
    assert rel_err(logits[0, rows], ref_rows) < CFG2_LOGIT_TOL
    filler = torch.nonzero(idx < 0).squeeze(1)
    assert bool(torch.isfinite(logits[0, filler]).all())


CFG4 = {"arch": "base", "active": {"text", "video_image_sequence"}, "overrides": {"dropout": 0.0},
        "adaptor_overrides": {"image_resnet": {"resnet_type": "resnet101"}}}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cfg4_video_real_shape_vs_oracle(dtype):
    """cfg-4 at the benchmarked shape (8 frames of 224 x 224 through resnet101 -> 1568 positions + 32 text -> 32 target), batch 1
    with one all-zero (padding) frame, train-mode BatchNorm, against the oracle: loss, logits, every gradient norm."""
    from ofasys_amd import ops
    model, d = build_model(CFG4, DEV, dtype)
    model.train()
    video = recipe.floats("bench.cfg4.video", (1, 3, 8, 224, 224))
    video[0, :, 6] = 0.0
    src = _tok("bench.cfg4.src", (1, 32), [27])
    prev = _tok("bench.cfg4.prev", (1, 32), [32], bos=True)
    target = make_target(prev)
    vals = [("VIDEO", True, video, None), ("TEXT", True, src, None), ("TEXT", False, prev, None)]
    state = oracle_state_for(model)
    cfg = OConfig(**ARCH["base"], resnet_layers=(3, 4, 23), training=True)
    loss, n, want = _oracle_step(state, cfg, [(vals, target)])
    with torch.no_grad():
        ref_logits, _ = restate.model_forward(oracle_state_for(model), cfg, oracle_slots(vals))
    logits, extra, enc = model(make_slots(vals, DEV, dtype), return_encoder_out=True)
    assert enc["encoder_padding_mask"][0].shape == (1, 1568 + 32)
    got_loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    model.zero_grad()
    got_loss.backward()
    torch.cuda.synchronize()
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    if dtype == torch.float32:
        assert rel_err(logits.detach().cpu(), ref_logits) < 1e-3
        assert abs(float(got_loss) - loss) <= 1e-3 * loss
        _check_grads(got, want)
    else:
        tol = 6e-2                                                     # the video case's bf16 bound (tests/test_model_gpu.py)
        assert rel_err(logits.detach().float().cpu(), ref_logits) < tol
        assert abs(float(got_loss) - loss) <= tol * loss
        outside = {k: v for k, v in want.items() if ".embed_images." not in k}
        _bf16_grad_check(got, outside, tol, "cfg-4 real shape bf16 (outside the trunk):")
        for k, p in model.named_parameters():                           # the 101-layer trunk in bf16: finite, and alive at the stem
            if p.grad is not None:
                assert bool(torch.isfinite(p.grad.float()).all()), k
        stem = "encoder.adaptor.image_resnet.embed_images.conv1.weight"
        assert 0.3 < float(got[stem].double().norm()) / float(want[stem].double().norm()) < 3.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_large_with_default_image_adaptor_vs_oracle(dtype):
    """OFA-large (D = 1024, 16 heads, 12 + 12 layers) with an IMAGE slot through the default adaptor (image_resnet, the reference's
    default trunk resnet152, adaptor/image_resnet.py:45-47; 224 x 224 -> 196 positions, 2-D rel-pos bias per layer) + ragged text
    (oracle/cases.py EXTRA_CASES["large_image"]).  fp32 holds the north-star 1e-3 on logits / loss / every gradient outside the
    trunk; inside the 152-layer train-mode BatchNorm trunk at batch 2 the gradient norms get 5e-2 (50 layers: 2e-2, see
    tests/test_model_gpu.py on the conditioning).  bf16: the REFERENCE ITSELF differs from its own fp32 run by 27.6 % of max |logit|
    on these inputs (oracle/ref_bf16_gap.py large_image) -- the bound is 2x that, i.e. this leg only shows the MFMA path of the
    large architecture runs an image slot end to end without blowing up; the arithmetic is pinned by the fp32 leg."""
    from ofasys_amd import ops
    from oracle.cases import EXTRA_CASES
    from tests.golden_util import case_inputs
    case = EXTRA_CASES["large_image"]
    model, d = build_model(case, DEV, dtype)
    model.train()
    vals, target = case_inputs(case)
    cfg = OConfig(**ARCH["large"], resnet_layers=(3, 8, 36), training=True)
    loss, n, want = _oracle_step(oracle_state_for(model), cfg, [(vals, target)])
    with torch.no_grad():
        ref_logits, _ = restate.model_forward(oracle_state_for(model), cfg, oracle_slots(vals))
    logits = model(make_slots(vals, DEV, dtype))[0]
    got_loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    model.zero_grad()
    got_loss.backward()
    torch.cuda.synchronize()
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    if dtype == torch.float32:
        assert rel_err(logits.detach().cpu(), ref_logits) < 1e-3
        assert abs(float(got_loss) - loss) <= 1e-3 * loss
        _check_grads(got, want, backbone_tol=5e-2)
    else:
        tol = 2 * 0.2765
        assert rel_err(logits.detach().float().cpu(), ref_logits) < tol
        assert abs(float(got_loss) - loss) <= 0.1 * loss
        for k, p in model.named_parameters():
            if p.grad is not None:
                assert bool(torch.isfinite(p.grad.float()).all()), k


LARGE_AV = {"arch": "large", "active": {"text", "video_image_sequence", "audio_fbank"}, "overrides": {"dropout": 0.0},
            "adaptor_overrides": {}}


def test_large_with_video_and_audio_slots_vs_oracle():
    """OFA-large had only ever run token and image slots.  One update over two micro-batches -- a VIDEO slot (3
    frames of 64 x 64 through the default trunk resnet152, one all-zero padding frame, frame + patch rel-pos bias) and an AUDIO slot
    (fbank with ragged lengths and mask_emb rows) -- each with ragged text, fp32, against the oracle: loss, sample_size, every
    gradient norm (north-star 1e-3 tier outside the trunk; the 152-layer train-mode BatchNorm trunk at batch 2 keeps its
    documented 5e-2 on gradient norms, tests/test_model_gpu.py)."""
    from ofasys_amd.trainer import TrainStep
    from tests.golden_util import case_inputs
    video = dict(slots=[("VIDEO", True, ("vid", "large_av.video", (2, 3, 3, 64, 64), [(1, 1)]), None),
                        ("TEXT", True, ("tok", "large_av.vsrc", (2, 6), [6, 4]), None),
                        ("TEXT", False, ("tok", "prev", (2, 7), [7, 5]), None)])
    audio = dict(slots=[("AUDIO", True, ("fbank", "large_av.audio", (2, 50, 80), [50, 37], [(0, 2), (0, 3), (1, 5)]), ["use_mask"]),
                        ("TEXT", True, ("tok", "large_av.asrc", (2, 4), [4, 3]), None),
                        ("TEXT", False, ("tok", "prev", (2, 6), [6, 5]), None)])
    mbs = [case_inputs(video), case_inputs(audio)]
    model, d = build_model(LARGE_AV, DEV, torch.float32)
    cfg = OConfig(**ARCH["large"], resnet_layers=(3, 8, 36), training=True)
    loss, n, want = _oracle_step(oracle_state_for(model), cfg, mbs)
    tr = TrainStep(model, lr=0.0, clip_norm=0.0)
    samples = [{"slots": make_slots(v, DEV), "target": t.to(DEV), "task": name} for (v, t), name in zip(mbs, ("video", "audio"))]
    out = tr.train_step(samples)
    torch.cuda.synchronize()
    assert int(out["stats"][0]) == n
    _measured("OFA-large video + audio step fp32: loss deviation", abs(float(out["stats"][1]) - loss) / loss)
    assert abs(float(out["stats"][1]) - loss) <= 1e-3 * loss
    got = _arena_grads(tr, model)
    _check_grads(got, want, backbone_tol=5e-2)
    for k in ("encoder.adaptor.video_image_sequence.embed_frame_positions.weight", "encoder.adaptor.audio_fbank.mask_emb",
              "encoder.adaptor.audio_fbank.subsample.conv.0.weight"):
        assert float(got[k].abs().max()) > 0.0, k


def test_cfg3_two_task_step_resnet101():
    """tests/test_configs_gpu.py's cfg-3 step with the resnet101 trunk BASELINE.md names for cfg-2b / cfg-3 (that test runs resnet50)."""
    from ofasys_amd.trainer import TrainStep
    from tests.test_configs_gpu import CFG3, _cfg3_batches
    case = dict(CFG3, adaptor_overrides={"image_resnet": {"resnet_type": "resnet101"}})
    a, b = _cfg3_batches()
    model, d = build_model(case, DEV, torch.float32)
    cfg = OConfig(**ARCH["base"], resnet_layers=(3, 4, 23), training=True)
    loss, n, want = _oracle_step(oracle_state_for(model), cfg, [a, b])
    tr = TrainStep(model, lr=0.0, clip_norm=0.0)
    samples = [{"slots": make_slots(v, DEV), "target": t.to(DEV), "task": name} for (v, t), name in ((a, "caption"), (b, "text"))]
    out = tr.train_step(samples)
    torch.cuda.synchronize()
    assert int(out["stats"][0]) == n
    assert abs(float(out["stats"][1]) - loss) <= 1e-3 * loss
    _check_grads(_arena_grads(tr, model), want)


# ------------------------------------------------------------------------------------------------------------ cfg-2b
def _cfg2b_structures(d, B, want=4, tries=16):
    """Batches of bench.py's cfg-2b workload with `want` DISTINCT step structures (packed row buckets differ with the drawn lengths)."""
    import bench
    from ofasys_amd.trainer import sample_structure
    seen, out = set(), []
    for i in range(tries):
        sample, ntok, lens = bench.make_batch(d, B, 252, 64, 97 * i, torch.device(DEV), "cfg2b", pack=True)
        key = sample_structure([sample])
        if key not in seen:
            seen.add(key)
            out.append((sample, lens))
        if len(out) == want:
            break
    return out


def test_cfg2b_benchmarked_step_packed_graph_vs_oracle():
    """The reference's DEFAULT configuration as bench.py --workload cfg2b builds and steps it -- ResNet-101 trunk (train-mode
    BatchNorm over the whole batch), position-biased attention, ragged row packing, bf16, B = 32, several batch structures each
    with its own hipGraph in ONE memory pool -- against the oracle.  Every structure is captured, then all are replayed round-robin
    twice (the allocator cache flushed in between: what a captured graph addresses must not depend on it); the LAST replay runs the
    first batch, whose loss, sample size, clip norm and gradient norms are compared with the oracle on the same bf16 weights
    (fp32 arithmetic, the padded shapes computed in full as the reference does).  This is the step that died with a GPU memory
    access fault in round 5 (profiles/round6_graph_fault_root_cause.txt)."""
    import bench
    from ofasys_amd.trainer import TrainStep
    B = 32
    args = SimpleNamespace(arch="base", workload="cfg2b", dtype="bf16", dropout=0.0)      # dropout 0: the oracle has none
    bench._HALF_NOW[0] = torch.bfloat16
    model, d = bench.build(args, torch.device(DEV))
    batches = _cfg2b_structures(d, B)
    assert len(batches) >= 3, "the synthetic length distribution no longer produces several row buckets"
    sample, (slens, tlens) = batches[0]
    state = _state_from_model(model)
    cfg = OConfig(**ARCH["base"], resnet_layers=(3, 4, 23), training=True)
    img, src, prev = (s.value.detach().cpu() for s in sample["slots"])
    target = sample["target"].cpu()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    params = oracle_params(state)
    oslots = [OSlot("IMAGE", True, img.float(), None), OSlot("TEXT", True, src), OSlot("TEXT", False, prev)]
    lg, _ = restate.model_forward(state, cfg, oslots)       # ONE pass over the whole batch: BatchNorm's statistics are the batch's
    ref_loss, n = restate.cross_entropy(lg, target)
    ref_loss.backward()
    ref_loss, n = float(ref_loss), int(n)
    want = {k: (None if p.grad is None else p.grad.detach()) for k, p in params.items()}
    ref_logits = lg.detach()
    del lg

    tr = TrainStep(model, lr=0.0, clip_norm=0.0, use_graph=True, graph_warmup=1)
    for s, _ in batches:                                     # one eager step + the capture (+ first replay) per structure
        for _ in range(2):
            tr.train_step([s])
    torch.cuda.synchronize()
    assert tr.captured_graphs() == len(batches)
    foreign = [r for e in tr._graphs.values() if "graphs" in e for r in tr.audit_report(e)]
    print(f"MEASURED cfg-2b capture audit: {len(foreign)} pinned default-pool tensors outside the engine's own (index / plan caches)")
    for rnd in range(2):
        torch.cuda.empty_cache()
        order = batches[1:] + batches[:1] if rnd else batches
        for s, _ in order:
            out = tr.train_step([s])
        torch.cuda.synchronize()
    assert int(out["stats"][0]) == n == sum(tlens)
    dev_loss = abs(float(out["stats"][1]) - ref_loss) / ref_loss
    _measured("cfg-2b packed graph step (B = 32): loss deviation", dev_loss)
    assert dev_loss <= 5e-3
    got = _arena_grads(tr, model)
    _bf16_grad_check(got, want, 4 * CFG2_GRAD_TOL, "cfg-2b packed graph step (B = 32):")
    gn = np.sqrt(sum(float(g.double().pow(2).sum()) for g in want.values() if g is not None)) / n
    _measured("cfg-2b packed graph step (B = 32): clip-norm deviation", abs(float(out["gnorm"]) - gn) / gn)
    assert abs(float(out["gnorm"]) - gn) <= 2e-2 * gn
    model.train()
    with torch.no_grad():
        logits = model(sample["slots"], pack=sample["pack"])[0].float().cpu()
    idx = sample["pack"].dec_index.cpu()
    rows = torch.nonzero(idx >= 0).squeeze(1)
    ref_rows = ref_logits.reshape(-1, ref_logits.shape[-1])[idx[rows]]
    _measured("cfg-2b packed forward (B = 32): logits max |diff| / max |logit|", rel_err(logits[0, rows], ref_rows))
    assert rel_err(logits[0, rows], ref_rows) < 5e-2


def _oslots_of(sample):
    """bench.make_micro's device slots as oracle slots (CPU; floating-point values in fp32 -- the bf16 values themselves)."""
    def cpu(v):
        if isinstance(v, dict):
            return {k: cpu(x) for k, x in v.items()}
        v = v.detach().cpu()
        return v.float() if v.is_floating_point() else v
    return [OSlot(s.modality.name, s.is_src, cpu(s.value), s.attributes) for s in sample["slots"]]


# measured on MI355X (profiles/round6_parity_measured.txt: loss 1.46e-5, clip norm 4.01e-3, worst per-parameter gradient norm 6.62e-3 of 774
# parameters); the bounds below are 2x these (the loss: 2x a rounded-up 5e-5 -- a sum of 7 bf16-logit losses moves by 1e-5 between boxes)
CFG5_LOSS_DEV, CFG5_GNORM_DEV, CFG5_GRAD_DEV = 5.0e-5, 4.0e-3, 6.6e-3


def test_cfg5_benchmarked_step_large_bf16_packed_vs_oracle():
    """cfg-5 AT ITS OWN SIZE: OFA-large (12 + 12 layers, D = 1024, 16 heads) with every adaptor bench.py --workload
    cfg5 activates (text, image_resnet with the resnet152 trunk, video_image_sequence, audio_fbank), bf16, the seven micro-batches
    bench.make_micro builds -- text, image (ragged row packing), box, video, audio, struct, motion -- accumulated into ONE update by a
    replayed hipGraph, against the oracle on the model's own bf16 weights (fp32 arithmetic, one micro-batch after the other, as
    engine/trainer.py:747-840 sums the tasks' gradients).  Micro-batch 2 (video: 1) bounds the CPU leg.  Held to twice what it measured:
    loss, clip norm and per-parameter gradient norms -- not the 2 x 27.6 % "does not blow up" bound of the large image leg above."""
    import bench
    from ofasys_amd.trainer import TrainStep
    args = SimpleNamespace(arch="large", workload="cfg5", dtype="bf16", dropout=0.0)      # dropout 0: the oracle has none
    bench._HALF_NOW[0] = torch.bfloat16
    model, d = bench.build(args, torch.device(DEV))
    samples = []
    for i, kind in enumerate(bench.STEP_MICRO["cfg5"]):
        s, _, _ = bench.make_micro(d, kind, 1 if kind == "video" else 2, 7 + i, torch.device(DEV))
        s.setdefault("task", kind)
        samples.append(s)
    assert "pack" in samples[1]                                                             # the image micro-batch runs packed rows
    state = _state_from_model(model)
    cfg = OConfig(**ARCH["large"], resnet_layers=(3, 8, 36), training=True)
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    params = oracle_params(state)
    ref_loss, n = 0.0, 0
    for s in samples:
        lg, _ = restate.model_forward(state, cfg, _oslots_of(s))
        l, k = restate.cross_entropy(lg, s["target"].cpu())
        l.backward()
        ref_loss += float(l.detach())
        n += int(k)
        del lg, l
    want = {k: (None if p.grad is None else p.grad.detach()) for k, p in params.items()}

    tr = TrainStep(model, lr=0.0, clip_norm=0.0, use_graph=True, graph_warmup=1)
    for _ in range(3):                                       # eager, capture (+ first replay), replay
        out = tr.train_step(samples)
    torch.cuda.synchronize()
    assert tr.captured_graphs() == 1
    assert int(out["stats"][0]) == n
    dev_loss = abs(float(out["stats"][1]) - ref_loss) / ref_loss
    _measured("cfg-5 large bf16 seven-micro-batch graph step: loss deviation", dev_loss)
    got = _arena_grads(tr, model)
    gn = np.sqrt(sum(float(g.double().pow(2).sum()) for g in want.values() if g is not None)) / n
    dev_gn = abs(float(out["gnorm"]) - gn) / gn
    _measured("cfg-5 large bf16 seven-micro-batch graph step: clip-norm deviation", dev_gn)
    for k, g in got.items():
        assert bool(torch.isfinite(g.float()).all()), k
    # the 152-layer train-mode BatchNorm trunk at micro-batch 2 / 1 is the ill-conditioned part (tests/test_model_gpu.py): its
    # parameters are held to finiteness and the stem's norm; everything else to per-parameter gradient norms
    outside = {k: v for k, v in want.items() if ".embed_images." not in k and ".embed_video." not in k}
    _bf16_grad_check(got, outside, 2 * CFG5_GRAD_DEV / 2.5, "cfg-5 large bf16 seven-micro-batch graph step (outside the trunks):")
    assert dev_loss <= 2 * CFG5_LOSS_DEV
    assert dev_gn <= 2 * CFG5_GNORM_DEV


@pytest.mark.parametrize("pins", ["1", "0"])
def test_cfg2b_replayed_graphs_survive_dropped_caches(pins, monkeypatch):
    """The benchmarked cfg-2b step with dropout ON (the row-per-wave residual joins and their keep-bit tensors: the allocation pattern
    that exposed round 5's fault): every structure captured, then -- between replays -- every clearable cache of the package is
    dropped and the allocator cache flushed.  With pins (the product) the graphs cannot lose anything; without them (the negative
    control) the run documents that today's dangling candidates share their allocator blocks with live tensors -- it must still be
    finite, and a fault here would be the regression the pins exist for."""
    import bench
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    from tools.capture_audit import drop_caches
    monkeypatch.setenv("OFA_CAPTURE_PINS", pins)
    args = SimpleNamespace(arch="base", workload="cfg2b", dtype="bf16", dropout=None)
    bench._HALF_NOW[0] = torch.bfloat16
    model, d = bench.build(args, torch.device(DEV))
    ops.manual_seed(3)
    batches = _cfg2b_structures(d, 32)
    tr = TrainStep(model, lr=1e-4, clip_norm=1.0, use_graph=True)
    for s, _ in batches:
        for _ in range(tr.graph_warmup + 1):
            tr.train_step([s])
    torch.cuda.synchronize()
    assert tr.captured_graphs() == len(batches)
    losses = []
    for rnd in range(3):
        drop_caches(model)
        for s, _ in batches:
            out = tr.train_step([s])
            torch.cuda.synchronize()
            losses.append(float(out["stats"][1]) / max(float(out["stats"][0]), 1.0))
    tr.check()
    assert all(np.isfinite(v) and 5.0 < v < 12.0 for v in losses), losses
