"""GPU: Trainer.train_step (HIP: forward, criterion, backward into the flat arena, ofa_sumsq, ofa_step_schedule, ofa_adam_step)
against tests/golden/trainstep.npz -- the REFERENCE's own criterion + FairseqOptimizer.multiply_grads / clip_grad_norm +
Adam.step recorded on a 2-task x 2-micro-batch step for two consecutive updates (oracle/gen_trainstep_golden.py;
engine/trainer.py:747-884, module/utils.py:342-384, engine/optim/adam.py:144-218)."""
import numpy as np
import pytest
import torch

from oracle import trainstep_case as TC
from oracle.cases import VOCAB_EXTRA
from tests.golden_util import load_golden, rel_err
from tests.model_util import build_model, make_slots

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")]
DEV = "cuda"
CASE = {"arch": TC.ARCH, "active": TC.ACTIVE, "overrides": TC.OVERRIDES, "adaptor_overrides": {}}


def _samples(dtype=torch.float32):
    V = 4 + VOCAB_EXTRA
    out = []
    for t, task in enumerate(TC.TASKS):
        for specs in task:
            vals, target = TC.micro_batch(specs, V)
            out.append({"slots": make_slots(vals, DEV, dtype), "target": target.to(DEV), "task": f"task{t}"})
    return out


def _arena_views(tr, model):
    offs = {id(p): (o, p.numel()) for p, o in zip(tr.fp.params, tr.fp.offsets)}
    return {k: offs[id(p)] for k, p in model.named_parameters() if id(p) in offs}


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_matches_reference_recording(use_graph):
    from ofasys_amd.trainer import Trainer
    g = load_golden("trainstep")
    h = TC.HYPER
    keys = [str(k) for k in g["param_keys"]]
    model, d = build_model(CASE, DEV, torch.float32)
    tr = Trainer(model, lr=h["lr"], betas=h["betas"], eps=h["eps"], weight_decay=h["weight_decay"], clip_norm=h["clip_norm"],
                 use_graph=use_graph, graph_warmup=0)
    views = _arena_views(tr, model)
    samples = _samples()
    for step in range(TC.STEPS):
        p = f"s{step}."
        out = tr.train_step(samples)
        torch.cuda.synchronize()
        n = float(g[p + "task_sample_size"].sum())
        assert float(out["stats"][0]) == n
        assert abs(float(out["stats"][1]) - float(g[p + "task_loss"].sum())) <= 1e-3 * float(g[p + "task_loss"].sum())
        assert abs(float(out["gnorm"]) - float(g[p + "gnorm"][0])) <= 1e-3 * float(g[p + "gnorm"][0])
        # Adam moments of EVERY parameter: linear / quadratic in the multiplied, clipped gradient
        for i, k in enumerate(keys):
            o, cnt = views[k]
            want_m, want_v = float(g[p + "exp_avg_norms"][i]), float(g[p + "exp_avg_sq_norms"][i])
            got_m = float(tr.exp_avg[o:o + cnt].double().norm())
            got_v = float(tr.exp_avg_sq[o:o + cnt].double().norm())
            if want_m < 0:                     # unused parameter (find_unused_parameters): zero gradient, zero moments
                assert got_m == 0.0 and got_v == 0.0, k
                continue
            assert abs(got_m - want_m) <= 2e-3 * want_m + 1e-9, (k, got_m, want_m)
            assert abs(got_v - want_v) <= 4e-3 * want_v + 1e-14, (k, got_v, want_v)
        params = dict(model.named_parameters())
        # absolute floor: gradients that are zero in exact arithmetic (cross_pos_k_linear.bias: a per-row constant under the
        # softmax) are 1e-13-sized rounding noise on both sides
        m_floor = 1e-6 * max(float(np.abs(g[p + "exp_avg." + k]).max()) for k in TC.FULL)
        for k in TC.FULL:
            o, cnt = views[k]
            gc = g[p + "grad_clip." + k]
            gm = float(np.abs(gc).max())
            m_got = TC.sample(tr.exp_avg[o:o + cnt].view(params[k].shape)).cpu().numpy()
            v_got = TC.sample(tr.exp_avg_sq[o:o + cnt].view(params[k].shape)).cpu().numpy()
            m_want, v_want = g[p + "exp_avg." + k], g[p + "exp_avg_sq." + k]
            assert np.abs(m_got - m_want).max() <= 2e-3 * np.abs(m_want).max() + m_floor, k
            assert np.abs(v_got - v_want).max() <= 4e-3 * np.abs(v_want).max() + m_floor ** 2, k
            if gm < 1e3 * m_floor:
                continue
            # the parameter after the update, where m / (sqrt(v) + eps) is determined (gradient above rounding noise)
            want, mine = g[p + "param." + k], TC.sample(params[k].detach()).cpu().numpy()
            sel = np.abs(gc) > 1e-2 * gm
            assert sel.sum() > 0
            assert np.abs(want - mine)[sel].max() <= 2e-2 * h["lr"], k


def test_adam_kernel_vs_reference_recording():
    """ofa_adam_step on the recorded pre-multiply gradients with coef = clip_coef / sum(sample_size): moments and parameters of
    both recorded updates (adam.py:192-212, decoupled weight decay included)."""
    from ofasys_amd import kernels as K
    g = load_golden("trainstep")
    h = TC.HYPER
    from oracle import recipe
    for k in TC.FULL:
        shape = None
        p0 = None
        for item in load_golden("tiny_text")["state_keys"]:
            key, shp, _ = str(item).split("|")
            if key == k:
                shape = tuple(int(x) for x in shp.strip("()").split(",") if x.strip())
        p0 = TC.sample(recipe.value_for(k, shape)).contiguous().reshape(-1).to(DEV)
        master, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        model = master.clone()
        for step in range(TC.STEPS):
            p = f"s{step}."
            grad = torch.from_numpy(g[p + "grad_pre." + k]).reshape(-1).to(DEV)
            coef = torch.tensor([float(g[p + "clip_coef"][0]) / float(g[p + "task_sample_size"].sum())], device=DEV)
            K.adam_step(master, m, v, grad, model, coef, h["lr"], h["betas"][0], h["betas"][1], h["eps"], h["weight_decay"], step + 1)
            torch.cuda.synchronize()
            assert rel_err(m.cpu(), g[p + "exp_avg." + k].reshape(-1)) < 1e-5, k
            assert rel_err(v.cpu(), g[p + "exp_avg_sq." + k].reshape(-1)) < 1e-5, k
            gc = np.abs(g[p + "grad_clip." + k].reshape(-1))
            sel = gc > 1e-3 * gc.max()
            assert np.abs(master.cpu().numpy() - g[p + "param." + k].reshape(-1))[sel].max() <= 2e-3 * h["lr"], k
    # sum of squares of a recorded gradient == its recorded norm^2
    grad = torch.from_numpy(g["s0.grad_pre.decoder.layers.3.ffn_layernorm.weight"]).to(DEV)
    out = torch.zeros(1, device=DEV)
    K.sumsq(grad, out)
    i = [str(x) for x in g["param_keys"]].index("decoder.layers.3.ffn_layernorm.weight")
    assert abs(float(out) ** 0.5 - float(g["s0.grad_pre_norms"][i])) <= 1e-5 * float(g["s0.grad_pre_norms"][i])


def test_dynamic_loss_scaler_kernel_vs_reference_recording():
    """ofa_step_schedule_scaled (scale, overflow counters, skip flag, multiply factor, all on the device) against the reference's
    DynamicLossScaler + fp16-optimizer arithmetic recorded in tests/golden/loss_scaler.json."""
    import json
    import os
    from ofasys_amd import kernels as K
    from oracle import scaler_cases as SC
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "loss_scaler.json")))
    for name, c in SC.CASES.items():
        ls = torch.tensor([c["init_scale"], 0, -1, -1, 0, 0, 0, 0], dtype=torch.float64, device=DEV)
        step = torch.zeros(1, dtype=torch.float64, device=DEV)
        lr = torch.tensor([1e-3], dtype=torch.float64, device=DEV)
        sched, gnorm = torch.zeros(5, device=DEV), torch.zeros(1, device=DEV)
        n_ok = 0
        for (raw, n), want in zip(c["seq"], g[name]):
            raw = SC.raw_value(raw)
            gsq = torch.tensor([raw * raw], dtype=torch.float32, device=DEV)
            stats = torch.tensor([float(n), 0.0, float(n)], dtype=torch.float64, device=DEV)
            K.step_schedule_scaled(gsq, stats, step, lr, sched, gnorm, ls, c["clip"], 0.9, 0.999, c["scale_factor"], c["scale_window"],
                                   c["tolerance"], c["threshold"], c["min_loss_scale"])
            h = ls.tolist()
            assert h[0] == want["loss_scale"] and h[1] == want["iter"], (name, want, h)
            assert (h[5] == 1.0) == (want["status"] == "fatal") or any(r["status"] == "fatal" for r in g[name])
            if want["status"] == "ok":
                n_ok += 1
                assert float(sched[3]) == 0.0 and abs(float(sched[0]) - want["multiply_factor"]) <= 2e-6 * want["multiply_factor"]
                assert abs(float(gnorm) - want["grad_norm"]) <= 2e-6 * want["grad_norm"]
            else:
                assert float(sched[3]) == 1.0 and float(sched[0]) == 0.0 and float(sched[1]) == 0.0
            assert float(step) == n_ok                              # Adam's update counter advances on real updates only


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_with_dynamic_loss_scale(use_graph):
    """TrainStep(loss_scale=...): the scaled run walks the unscaled trajectory (powers of two scale gradients exactly), the scale
    doubles every scale_window clean updates, a poisoned step is skipped on the device with the scale halved, training goes on."""
    from ofasys_amd.trainer import TrainStep
    from oracle.cases import CASES
    from tests.golden_util import case_inputs
    case = dict(CASES["tiny_text"], overrides={"dropout": 0.0})
    vals, target = case_inputs(case)
    runs = []
    for scaled in (False, True):
        model, d = build_model(case, DEV, torch.bfloat16)
        tr = TrainStep(model, lr=1e-3, clip_norm=1.0, use_graph=use_graph, graph_warmup=1,
                       loss_scale={"init_scale": 64.0, "scale_window": 3} if scaled else None)
        batch = {"slots": make_slots(vals, DEV, torch.bfloat16), "target": target.to(DEV)}
        losses = [float(tr.train_step([batch])["stats"][1]) for _ in range(7)]
        runs.append((losses, tr, model))
    (l0, _, _), (l1, tr, model) = runs
    # the first update sees identical weights (2^k scaling is exact); later ones drift apart like any two bf16 runs whose clip
    # coefficient differs in the 6th digit (clip/(gn + 1e-6) clamped vs the fp16 optimizer's clip/gn)
    assert abs(l0[0] - l1[0]) <= 1e-6 * abs(l0[0]) and abs(l0[1] - l1[1]) <= 2e-3 * abs(l0[1])
    for a, b in zip(l0, l1):                      # (seven updates at lr 1e-3 amplify that 6th-digit difference: the same ballpark, not more)
        assert abs(a - b) <= 1e-1 * abs(a)
    # iters 0..6 clean, last_overflow_iter = -1: the scale doubles at iter 2 and 5 ((iter + 1) % 3 == 0)
    assert float(tr.last["loss_scale"]) == 64.0 * 4 and float(tr._ls[1]) == 7.0
    with torch.no_grad():
        p = dict(model.named_parameters())["decoder.layers.0.fc1.bias"]
        keep = p[0].clone()
        p[0] = float("inf")
    w, t = tr.master.clone(), float(tr._step_t)
    out = tr.train_step([batch])
    torch.cuda.synchronize()
    with torch.no_grad():
        p[0] = keep
        tr.master[:] = torch.where(torch.isfinite(tr.master), tr.master, w)      # (the poisoned element itself)
    assert float(out["skipped"][0]) == 1.0 and float(out["loss_scale"]) == 128.0 and float(tr._step_t) == t
    assert torch.equal(tr.master, w)
    tr.check()                                                       # an overflow is not an error for the scaler
    out = tr.train_step([batch])
    torch.cuda.synchronize()
    assert float(out["skipped"][0]) == 0.0 and float(tr._step_t) == t + 1
