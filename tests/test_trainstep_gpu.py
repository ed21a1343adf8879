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
