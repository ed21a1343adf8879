"""GPU: the BASELINE.json configurations round 1 left untested, each against the CPU oracle (pinned to
the reference by tests/golden/*) or through size-independent properties at the full shape:

  cfg-3  base, two tasks in ONE step: caption through the default image_resnet adaptor + text -> text
  cfg-4  base, video clip 8 x 224 x 224 (1568 patch tokens + text) at the real shape
  cfg-5  OFA-large golden (tests/test_model_gpu.py picks `large_multislot` up) + a 7-micro-batch mixed-modality step
  f3     collate -> to_device("cuda") -> model
  b-i    scripts/trainer_api.py with only the import line changed, on synthetic in-memory datasets
"""
import numpy as np
import pytest
import torch

from oracle import recipe, restate
from oracle.cases import CASES, VOCAB_EXTRA, make_target
from oracle.restate import OConfig
from tests.golden_util import ARCH, case_inputs, oracle_params, oracle_slots, oracle_state_for, rel_err
from tests.model_util import build_model, make_slots

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")]
DEV = "cuda"
V = 4 + VOCAB_EXTRA


def _tok(key, shape, lengths=None, bos=False):
    return recipe.tokens("cfg." + key, shape, V, lengths, bos=0 if bos else None)


def _oracle_step(state, cfg, micro_batches):
    """Accumulated loss / sample_size / gradients of a list of (vals, target) micro-batches on the CPU oracle."""
    torch.set_num_threads(min(64, torch.get_num_threads() if torch.get_num_threads() > 8 else 64))
    params = oracle_params(state)
    loss_sum, n_sum = 0.0, 0
    for vals, target in micro_batches:
        logits, _ = restate.model_forward(state, cfg, oracle_slots(vals))
        loss, n = restate.cross_entropy(logits, target)
        loss.backward()
        loss_sum += float(loss.detach())
        n_sum += n
    grads = {k: (None if p.grad is None else p.grad.detach()) for k, p in params.items()}
    return loss_sum, n_sum, grads


def _arena_grads(tr, model):
    offs = {id(p): (o, p.numel()) for p, o in zip(tr.fp.params, tr.fp.offsets)}
    return {k: tr.fp.grad[offs[id(p)][0]:offs[id(p)][0] + offs[id(p)][1]].view(p.shape) for k, p in model.named_parameters()
            if id(p) in offs}


def _check_grads(got, want, backbone_tol=2e-2, tol=2e-3):
    """Per-parameter gradient norms: tol outside the ResNet backbone, backbone_tol inside (ill-conditioned BatchNorm chain,
    see tests/test_model_gpu.py)."""
    gmax = max(float(g.double().norm()) for g in want.values() if g is not None)
    checked = 0
    for k, w in want.items():
        if k not in got:
            continue
        g = float(got[k].double().norm())
        if w is None:
            assert g == 0.0, k
            continue
        wn = float(w.double().norm())
        t = backbone_tol if ".embed_images." in k else tol
        assert abs(g - wn) <= t * wn + 1e-5 * gmax, (k, g, wn)
        checked += 1
    assert checked > 50


# ------------------------------------------------------------------------------------------------------------ cfg-3
CFG3 = {"arch": "base", "active": {"text", "image_resnet"}, "overrides": {"dropout": 0.0},
        "adaptor_overrides": {"image_resnet": {"resnet_type": "resnet50"}}}


def _cfg3_batches():
    img = recipe.floats("cfg.cfg3.image", (2, 3, 224, 224))
    a_prev = _tok("cfg3.a.prev", (2, 12), [12, 7], bos=True)
    a = ([("IMAGE", True, img, None), ("TEXT", True, _tok("cfg3.a.src", (2, 9), [9, 5]), None), ("TEXT", False, a_prev, None)],
         make_target(a_prev))
    b_prev = _tok("cfg3.b.prev", (2, 128), [128, 77], bos=True)
    b = ([("TEXT", True, _tok("cfg3.b.src", (2, 128), [101, 128]), None), ("TEXT", False, b_prev, None)], make_target(b_prev))
    return a, b


def test_cfg3_two_task_step_base_heterogeneous_adaptors():
    """One TrainStep over [caption via image_resnet (196 patch tokens, 2-D rel-pos bias, BatchNorm in train mode), text -> text
    (Ts = Tt = 128)] on OFA-base, fp32, against the oracle: loss, sample_size, every gradient norm, grad-norm.  Then the
    text-only step: gradients of the inactive image adaptor are EXACTLY zero (arena zero-fill, find_unused_parameters)."""
    from ofasys_amd.trainer import TrainStep
    a, b = _cfg3_batches()
    model, d = build_model(CFG3, DEV, torch.float32)
    state = oracle_state_for(model)
    cfg = OConfig(**ARCH["base"], resnet_layers=(3, 4, 6), training=True)
    loss, n, want = _oracle_step(state, cfg, [a, b])
    tr = TrainStep(model, lr=0.0, clip_norm=0.0)                 # lr 0: the arena keeps this step's gradients, weights stay put
    samples = [{"slots": make_slots(v, DEV), "target": t.to(DEV), "task": name} for (v, t), name in ((a, "caption"), (b, "text"))]
    out = tr.train_step(samples)
    torch.cuda.synchronize()
    assert int(out["stats"][0]) == n
    assert abs(float(out["stats"][1]) - loss) <= 1e-3 * loss
    gn = np.sqrt(sum(float(g.double().pow(2).sum()) for g in want.values() if g is not None)) / n
    assert abs(float(out["gnorm"]) - gn) <= 5e-3 * gn
    _check_grads(_arena_grads(tr, model), want)
    # text-only step: the image adaptor is unused
    tr.train_step(samples[1:])
    torch.cuda.synchronize()
    got = _arena_grads(tr, model)
    unused = [k for k in got if ".image_resnet." in k]
    assert len(unused) > 100 and all(float(got[k].abs().max()) == 0.0 for k in unused)
    assert float(got["encoder.layers.0.fc1.weight"].abs().max()) > 0.0


def _shifted_slots(vals, shift):
    out = []
    for m, is_src, x, attrs in vals:
        if not x.is_floating_point():
            x = torch.where(x > 3, 4 + (x - 4 + shift) % (V - 4), x)        # specials (bos / pad / eos) stay
        out.append((m, is_src, x, attrs))
    return make_slots(out, DEV, torch.bfloat16)


def test_cfg3_graph_replay_of_two_step_structures():
    """Steps of two structures ([caption, text] and [text]) interleaved: each structure gets its own hipGraph, replays pick up
    new batches through the static inputs, and the trajectory equals the eager one (bf16, dropout on)."""
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    a, b = _cfg3_batches()
    runs = []
    for use_graph in (False, True):
        case = dict(CFG3, overrides={})
        model, d = build_model(case, DEV, torch.bfloat16)
        tr = TrainStep(model, lr=1e-3, clip_norm=1.0, use_graph=use_graph, graph_warmup=1)
        ops.manual_seed(11)
        losses = []
        for step in range(8):
            shift = step // 2                                        # new token values every other step: static inputs are refreshed
            mb_a = {"slots": _shifted_slots(a[0], shift), "target": a[1].to(DEV)}
            mb_b = {"slots": _shifted_slots(b[0], shift), "target": b[1].to(DEV)}
            losses.append(float(tr.train_step([mb_a, mb_b] if step % 2 == 0 else [mb_b])["stats"][1]))
        torch.cuda.synchronize()
        runs.append((losses, tr.master.clone(), tr))
    (l0, m0, _), (l1, m1, tr1) = runs
    assert sum(1 for e in tr1._graphs.values() if "graphs" in e) == 2
    assert l0 == l1 and torch.equal(m0, m1)


# ------------------------------------------------------------------------------------------------------------ cfg-5
CFG5 = {"arch": "tiny", "active": {"text", "image_resnet", "video_image_sequence", "audio_fbank"}, "overrides": {"dropout": 0.0},
        "adaptor_overrides": {"image_resnet": {"resnet_type": "resnet50"}}}


def _cfg5_batches():
    """7 micro-batches, one per modality family of BASELINE.json configs[4] (struct / motion enter as tokens through the text
    adaptor, adaptor/general.py:36-46)."""
    out = []
    for name in ("tiny_text", "tiny_resnet", "tiny_multislot", "tiny_video", "tiny_audio"):
        out.append(case_inputs(CASES[name]))
    for mod in ("STRUCT", "MOTION"):
        prev = _tok(f"cfg5.{mod}.prev", (2, 7), [7, 5], bos=True)
        out.append(([(mod, True, _tok(f"cfg5.{mod}.src", (2, 11), [11, 6]), None), ("TEXT", True, _tok(f"cfg5.{mod}.q", (2, 4)), None),
                     ("TEXT", False, prev, None)], make_target(prev)))
    return out


@pytest.mark.parametrize("use_graph", [False, True])
def test_cfg5_seven_modality_mixed_step(use_graph):
    """text, image, box(+struct), video, audio, struct, motion-as-tokens micro-batches accumulated in ONE step on a model with every
    adaptor active (ragged slot collation: every micro-batch has its own slot list and lengths), against the oracle."""
    from ofasys_amd.trainer import TrainStep
    mbs = _cfg5_batches()
    model, d = build_model(CFG5, DEV, torch.float32)
    state = oracle_state_for(model)
    cfg = OConfig(**ARCH["tiny"], resnet_layers=(3, 4, 6), training=True)
    loss, n, want = _oracle_step(state, cfg, mbs)
    tr = TrainStep(model, lr=0.0, clip_norm=0.0, use_graph=use_graph, graph_warmup=0)
    samples = [{"slots": make_slots(v, DEV), "target": t.to(DEV), "task": f"mb{i}"} for i, (v, t) in enumerate(mbs)]
    out = tr.train_step(samples)
    torch.cuda.synchronize()
    if use_graph:
        assert any("graphs" in e for e in tr._graphs.values())
    assert int(out["stats"][0]) == n
    assert abs(float(out["stats"][1]) - loss) <= 1e-3 * loss
    _check_grads(_arena_grads(tr, model), want, backbone_tol=5e-2)


# ------------------------------------------------------------------------------------------------------------ cfg-4
def test_cfg4_video_real_shape_properties():
    """cfg-4 at its real shape (B=2, 8 frames of 224 x 224 -> 1568 patch tokens + 32 text, OFA-base, bf16): finite loss and
    gradients, the zero frame of row 1 is masked, the frame-position table receives gradient only in the 8 rows used, and
    incremental decoding reproduces the teacher-forced logits."""
    from ofasys_amd import ModalityType, Slot, ops
    case = {"arch": "base", "active": {"text", "video_image_sequence"}, "overrides": {"dropout": 0.0},
            "adaptor_overrides": {"image_resnet": {"resnet_type": "resnet101"}}}
    model, d = build_model(case, DEV, torch.bfloat16)
    model.train()
    video = recipe.floats("cfg.cfg4.video", (2, 3, 8, 224, 224))
    video[1, :, 5] = 0.0                                               # an all-zero frame is padding (video_image_sequence.py:131-133)
    src = _tok("cfg4.src", (2, 32), [32, 20])
    prev = _tok("cfg4.prev", (2, 32), [32, 25], bos=True)
    target = make_target(prev)
    slots = make_slots([("VIDEO", True, video, None), ("TEXT", True, src, None), ("TEXT", False, prev, None)], DEV, torch.bfloat16)
    logits, extra, enc = model(slots, return_encoder_out=True)
    assert logits.shape == (2, 32, len(d))
    mask = enc["encoder_padding_mask"][0]
    assert mask.shape == (2, 1568 + 32)
    assert not bool(mask[0, :1568].any()) and bool(mask[1, 5 * 196:6 * 196].all()) and int(mask[1, :1568].sum()) == 196
    assert int(mask[1, 1568:].sum()) == 12
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and float(loss) > 0
    params = dict(model.named_parameters())
    for k, p in params.items():
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad.float()).all()), k
    gf = params["encoder.adaptor.video_image_sequence.embed_frame_positions.weight"].grad.float()
    used = gf.abs().sum(1) > 0                                          # frame f uses row f + 1 (video_image_sequence.py:143-147)
    assert bool(used[1:9].all()) and not bool(used[0]) and not bool(used[9:].any())
    assert float(params["encoder.adaptor.image_resnet.embed_images.conv1.weight"].grad.float().abs().sum()) > 0
    # incremental == teacher-forced (eval mode: BatchNorm on running statistics both ways)
    model.eval()
    with torch.no_grad():
        full, _, enc = model(slots, return_encoder_out=True)
        inc = {}
        steps = []
        for t in range(1, 9):
            out, _ = model.decoder([Slot(ModalityType.TEXT, False, prev[:, :t].to(DEV))], encoder_out=enc, incremental_state=inc)
            steps.append(out[:, -1])
    got = torch.stack(steps, 1).float()
    assert rel_err(got.cpu(), full[:, :8].float().cpu()) < 3e-2


# ------------------------------------------------------------------------------------------------------------ f3
def test_collate_to_cuda_to_model():
    """SURVEY.md 8f-3 on the device: samples -> GeneralPreprocess (map / group_map / collate) -> to_device("cuda") (ONE pinned
    staging buffer, one H2D copy for every integer field) -> GeneralistModel.  Integer fields arrive bit-exact, the slots are
    views of one device buffer, and the model's output equals feeding the same tensors directly."""
    from ofasys_amd import Instruction, ModalityType, Slot
    from ofasys_amd.preprocessor import DefaultTextPreprocess, GeneralPreprocess, to_device
    case = CASES["tiny_multislot"]
    model, d = build_model(case, DEV, torch.float32)
    gp = GeneralPreprocess(d, {"text": DefaultTextPreprocess(d)})
    g = np.random.Generator(np.random.Philox(key=3))
    samples = []
    for i in range(5):
        n_src, n_tgt = int(g.integers(3, 12)), int(g.integers(2, 9))
        slots = [Slot(ModalityType.TEXT, True, torch.from_numpy(g.integers(4, V, n_src)), global_position=0),
                 Slot(ModalityType.TEXT, True, torch.from_numpy(g.integers(4, V, 3)), global_position=1),
                 Slot(ModalityType.TEXT, False, torch.from_numpy(g.integers(4, V, n_tgt)), global_position=2)]
        samples.append(gp(Instruction(slots, "[TEXT] [TEXT] -> [TEXT]", {"uid": i})))
    host = gp.collate(samples)
    ref = {"src": host["net_input"]["slots"][0].value.clone(), "prev": host["net_input"]["slots"][1].value.clone(),
           "target": host["target"].clone()}
    dev = to_device(host, "cuda")
    s_src, s_prev = dev["net_input"]["slots"]
    assert s_src.value.is_cuda and s_src.value.dtype == torch.int64 and dev["target"].is_cuda
    assert torch.equal(s_src.value.cpu(), ref["src"]) and torch.equal(s_prev.value.cpu(), ref["prev"])
    assert torch.equal(dev["target"].cpu(), ref["target"])
    assert len({s_src.value.untyped_storage().data_ptr(), s_prev.value.untyped_storage().data_ptr(),
                dev["target"].untyped_storage().data_ptr()}) == 1                     # one staging buffer, one copy
    assert sum(n for _, _, n, _, _ in dev["segments"]) >= ref["src"].numel() + ref["prev"].numel() + ref["target"].numel()
    model.eval()
    with torch.no_grad():
        a = model(dev["net_input"]["slots"])[0]
        b = model([Slot(ModalityType.TEXT, True, ref["src"].to(DEV)), Slot(ModalityType.TEXT, False, ref["prev"].to(DEV))])[0]
    assert torch.equal(a, b)
    # and against the oracle
    state = oracle_state_for(model)
    cfg = OConfig(**ARCH["tiny"])
    want, _ = restate.model_forward(state, cfg, oracle_slots([("TEXT", True, ref["src"], None), ("TEXT", False, ref["prev"], None)]))
    assert rel_err(a.cpu(), want.detach()) < 1e-3


# ------------------------------------------------------------------------------------------------------------ b-i
def test_trainer_api_script_runs_with_only_the_import_changed():
    """scripts/trainer_api.py:1-27 verbatim except (1) the import line and (2) the two `load_dataset(...)` calls, replaced by
    synthetic in-memory rows with the same column names (there is no network)."""
    from ofasys_amd import Task, Trainer, GeneralistModel

    # 1. Define the multi-modal tasks
    task1 = Task(
        name='caption',
        instruction='[IMAGE:image_url] what does the image describe? -> [TEXT:caption]',
        micro_batch_size=4,
    )
    task2 = Task(
        name='text_infilling',
        instruction='what is the complete text of " [TEXT:sentence,mask_ratio=0.3] "? -> [TEXT:sentence]',
        micro_batch_size=2,
    )

    # 2. Bind the dataset
    g = torch.Generator().manual_seed(0)
    task1.add_dataset([{"image_url": torch.randn(3, 224, 224, generator=g), "caption": f"a drawing of creature number {i}"}
                       for i in range(16)], 'train')
    task2.add_dataset([{"sentence": f"the {i}th book was written by a very careful author .", "label": i % 2} for i in range(16)], 'train')

    # 3. Create an OFA-Sys Unify Model
    model = GeneralistModel()
    # model.cfg.arch = 'base'

    # 4. Train all tasks together
    trainer = Trainer(max_update=6, log_interval=1, lr=1e-3)
    history = trainer.fit(model=model, tasks=[task1, task2])

    assert len(history) == 6 and all(np.isfinite(h["loss"]) and h["sample_size"] > 0 for h in history)
    assert model.cfg.adaptor.image_resnet.is_active and next(model.parameters()).is_cuda
    assert trainer.step_engine.num_updates == 6
    assert min(h["loss"] for h in history[-2:]) < history[0]["loss"]    # 6 updates at lr 1e-3 on repeated rows: the loss moves down
    assert any("graphs" in e for e in trainer.step_engine._graphs.values())


# ------------------------------------------------------------------------------------------------------------ step engine guards
def test_graph_static_inputs_cover_dict_slots_and_constraint_masks():
    """audio slots carry a dict {fbank, fbank_lengths, mask_indices} and `constraint_masks` sits on the sample.
    A replay must train on the NEW batch's values of all of them: graph mode fed changing batches equals eager."""
    from ofasys_amd.trainer import TrainStep, sample_structure
    case = CASES["tiny_audio"]
    vals, target = case_inputs(case)
    Vd = 4 + VOCAB_EXTRA
    runs = []
    for use_graph in (False, True):
        model, d = build_model(case, DEV, torch.float32)
        tr = TrainStep(model, lr=1e-2, clip_norm=1.0, use_graph=use_graph, graph_warmup=1, label_smoothing=0.1)
        losses = []
        for step in range(5):
            g = torch.Generator().manual_seed(100 + step)
            v2 = []
            for m, s, x, at in vals:
                if isinstance(x, dict):
                    fb = torch.randn(x["fbank"].shape, generator=g)
                    lens = torch.tensor([50, 30 + step])
                    for r, n in enumerate(lens):
                        fb[r, int(n):] = 0
                    mi = torch.zeros_like(x["mask_indices"])
                    mi[0, step % mi.shape[1]] = True
                    x = {"fbank": fb, "fbank_lengths": lens, "mask_indices": mi}
                v2.append((m, s, x, at))
            cm = torch.rand(target.shape + (Vd,), generator=g) > 0.3
            cm.scatter_(2, target.unsqueeze(-1), True)
            sample = {"slots": make_slots(v2, DEV), "target": target.to(DEV), "constraint_masks": cm.to(DEV)}
            losses.append(float(tr.train_step([sample])["stats"][1]))
        torch.cuda.synchronize()
        runs.append((losses, tr.master.clone(), tr))
    (l0, m0, _), (l1, m1, tr1) = runs
    assert any("graphs" in e for e in tr1._graphs.values())
    assert len(set(l0)) == len(l0)
    assert l0 == l1 and torch.equal(m0, m1)
    # the structure key separates what used to collide: adaptor attribute, modality, dict-valued slot shapes
    from ofasys_amd import ModalityType, Slot
    img = torch.zeros(1, 3, 224, 224)
    tok = torch.zeros(1, 4, dtype=torch.long)
    mk = lambda attrs, mod=ModalityType.IMAGE: [{"slots": [Slot(mod, True, img, attributes=attrs), Slot(ModalityType.TEXT, False, tok)],  # noqa: E731
                                                 "target": tok}]
    assert sample_structure(mk(None)) != sample_structure(mk(["adaptor=image_patch_embed"]))
    assert sample_structure(mk(None)) != sample_structure(mk(None, ModalityType.VIDEO))


def test_nonfinite_or_empty_step_is_skipped_on_the_device():
    """ADVICE r1 (medium) / engine/trainer.py:866-876: a step with a non-finite gradient norm (or no target token) must not
    touch weights, moments or the update counter; the host poll raises FloatingPointError."""
    from ofasys_amd.trainer import TrainStep
    case = CASES["tiny_text"]
    vals, target = case_inputs(case)
    model, d = build_model(case, DEV, torch.float32)
    tr = TrainStep(model, lr=1e-2, clip_norm=1.0)
    good = {"slots": make_slots(vals, DEV), "target": target.to(DEV)}
    tr.train_step([good])
    tr.check()
    w, m, v, t = tr.master.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), float(tr._step_t)
    empty = {"slots": make_slots(vals, DEV), "target": torch.full_like(target, d.pad()).to(DEV)}
    out = tr.train_step([empty])                                   # sample_size == 0
    torch.cuda.synchronize()
    assert float(out["skipped"][0]) == 1.0
    assert torch.equal(tr.master, w) and torch.equal(tr.exp_avg, m) and torch.equal(tr.exp_avg_sq, v) and float(tr._step_t) == t
    with pytest.raises(FloatingPointError):
        tr.check()
    tr.check()                                                     # reported once
    # an Inf in the gradient arena: same guard (inject through a poisoned parameter)
    with torch.no_grad():
        p = dict(model.named_parameters())["decoder.layers.0.fc1.bias"]
        keep = p[0].clone()
        p[0] = float("inf")
    out = tr.train_step([good])
    torch.cuda.synchronize()
    assert float(out["skipped"][0]) == 1.0 and not bool(torch.isfinite(out["gnorm"]).all())
    with torch.no_grad():
        p[0] = keep
    assert torch.equal(tr.exp_avg, m) and float(tr._step_t) == t
    with pytest.raises(FloatingPointError):
        tr.check()
    out = tr.train_step([good])                                    # and training continues
    torch.cuda.synchronize()
    assert float(out["skipped"][0]) == 0.0 and float(tr._step_t) == t + 1 and not torch.equal(tr.master, w)


# ------------------------------------------------------------------------------------------------------------ data parallel
def test_dp_step_graph_with_captured_rccl_collectives():
    """The benchmarked multi-GPU mode: ONE hipGraph holding forward, backward, the bucketed all-reduces launched from inside
    backward, the scalar all-reduce, clip and Adam.  One GPU here, so the process group is a 1-rank RCCL group and the step
    engine is told world = 2 (the reducer launches every collective it would launch on 2 ranks; a 1-rank sum is the identity,
    so the captured trajectory must equal the eager data-parallel one).  What this proves: torch + RCCL capture the collectives on their own
    stream inside the step graph and the replayed graph is correct; what it cannot prove: xGMI traffic."""
    import os
    import torch.distributed as dist
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        case = CASES["tiny_multislot"]
        vals, target = case_inputs(case)
        runs = []
        for dp in (False, True):
            model, d = build_model(case, DEV, torch.bfloat16)
            tr = TrainStep(model, lr=1e-3, clip_norm=1.0, use_graph=dp, graph_warmup=1, bucket_bytes=1 << 20, dp_graph="full")
            tr.world = tr.reducer.world = 2          # both runs: eager overlapped reduce vs the captured one (same fold order)
            batch = {"slots": make_slots(vals, DEV, torch.bfloat16), "target": target.to(DEV)}
            ops.manual_seed(5)
            losses = [float(tr.train_step([batch])["stats"][1]) for _ in range(6)]
            torch.cuda.synchronize()
            runs.append((losses, tr.master.clone(), tr))
        (l0, m0, _), (l1, m1, tr1) = runs
        entries = [e for e in tr1._graphs.values() if "graphs" in e]
        assert len(entries) == 1 and entries[0]["mode"] == "full" and len(entries[0]["graphs"]) == 1
        assert len(tr1.reducer.buckets) > 3 and tr1.reducer.knows(next(iter(tr1._graphs)))
        assert l0 == l1 and torch.equal(m0, m1)
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_end_to_end_on_one_gpu():
    """The driver's multi-GPU invocation, `python bench.py --gpus 2 ...` run BARE: bench.py self-launches two ranks under
    torch.distributed.run (127.0.0.1 rendezvous), every rank builds the model, broadcasts the weights, runs the data-parallel train
    step and rank 0 prints ONE JSON line with the whole-job rate.  One GPU here, so both ranks sit on device 0
    (OFA_BENCH_DEVICE=0) and the collectives go through gloo: gloo's host-side all-reduce cannot be captured into a hipGraph, so
    the step engine picks the split mode by itself (two graphs around an eager all-reduce)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OFA_BENCH_DEVICE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "4", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline", "--profile-gemm", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and abs(out["value"] - 2 * out["tokens_per_sec_per_gpu"]) <= 1e-6 * out["value"]
    assert out["config"]["step_mode"] == "two hipGraphs + eager all-reduce", out["config"]["step_mode"]
    dp = out["dp"]                                                   # the exchange describes itself
    assert dp["world_size"] == 2 and dp["backend"] == "gloo" and dp["buckets"] == len(dp["bucket_bytes"]) == len(dp["bucket_allreduce_ms_alone"])
    assert dp["launch_order"] == list(range(dp["buckets"])) and dp["exposed_wait_ms_per_step"] is not None


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the one-GPU lease runs the gloo variant above")
def test_bench_two_ranks_rccl_full_graph():
    """The product path of `bench.py --gpus 2`: one rank per GPU, backend nccl (= RCCL), the bucketed all-reduces captured inside the
    step graph.  Runs wherever two devices are visible."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--profile-gemm", "0"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["step_mode"] == "one hipGraph (collectives captured)"
    dp = out["dp"]
    assert dp["world_size"] == 2 and dp["backend"] == "nccl" and dp["rccl_version"][0].isdigit()
    assert dp["launch_order"] == list(range(dp["buckets"])) and dp["buckets_launched_inside_backward"] >= 1
    assert all(t > 0 for t in dp["bucket_allreduce_ms_alone"])


def test_bench_captured_collective_selftest_child_and_fallback_decision():
    """bench.py asks a CHILD process per rank whether RCCL collectives captured in a hipGraph replay on this node before it captures the
    real step (`captured_collectives_ok`; a failed capture cannot be retried in the process it failed in), and falls back to
    dp_graph = split when any rank says no.  One GPU here: the child runs as a one-rank job (its RCCL communicator, its capture, two
    replays on fresh data), and the parent-side helper returns False -- not an exception, not a hang -- when the child cannot work."""
    import importlib.util
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dp-selftest-child"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    old = dict(os.environ)
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 1))
        assert bench.captured_collectives_ok(1, 0, 0, timeout=300) is True
        assert bench.captured_collectives_ok(1, 0, 99, timeout=300) is False          # a device that does not exist: the child fails, the parent survives
    finally:
        os.environ.clear()
        os.environ.update(old)
