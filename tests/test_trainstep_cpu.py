"""CPU: the oracle's restatement of the train-step UPDATE arithmetic (oracle/restate.py: cross_entropy, multiply_clip,
adam_update) against tests/golden/trainstep.npz, which oracle/gen_trainstep_golden.py recorded from the reference's own
criterion, FairseqOptimizer.multiply_grads / clip_grad_norm and Adam.step on a 2-task x 2-micro-batch step."""
import numpy as np
import torch

from oracle import restate
from oracle import trainstep_case as TC
from oracle.cases import VOCAB_EXTRA
from tests.golden_util import ARCH, load_golden, oracle_slots, rel_err
from oracle.restate import OConfig


def build_oracle_state():
    """Every state entry of tiny + text adaptor from the recipe (schema = the tiny_text golden's state_keys)."""
    from tests.golden_util import state_from_golden
    st = state_from_golden(load_golden("tiny_text"))
    return st


def run_oracle_steps(g):
    torch.set_num_threads(8)
    V = 4 + VOCAB_EXTRA
    h = TC.HYPER
    state = build_oracle_state()
    keys = [str(k) for k in g["param_keys"]]
    params = {k: state[k].requires_grad_(True) for k in keys}
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    cfg = OConfig(**ARCH[TC.ARCH], training=True)
    m = {k: torch.zeros_like(p) for k, p in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    rec = []
    for step in range(TC.STEPS):
        for p in params.values():
            p.grad = None
        sizes, losses = [], []
        for task in TC.TASKS:
            n_t, l_t = 0, 0.0
            for specs in task:
                vals, target = TC.micro_batch(specs, V)
                logits, _ = restate.model_forward(state, cfg, oracle_slots(vals))
                loss, n = restate.cross_entropy(logits, target)
                loss.backward()
                n_t += n
                l_t += float(loss.detach())
            sizes.append(n_t)
            losses.append(l_t)
        gpre = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in params.items()}
        gmul, _, _ = restate.multiply_clip(gpre, float(sum(sizes)), 0.0)
        gclip, gnorm, coef = restate.multiply_clip(gpre, float(sum(sizes)), h["clip_norm"])
        before = {k: p.detach().clone() for k, p in params.items()}
        with torch.no_grad():
            for k, p in params.items():
                if gclip[k] is None:
                    continue
                restate.adam_update(p, gclip[k], m[k], v[k], step + 1, h["lr"], h["betas"], h["eps"], h["weight_decay"])
        rec.append(dict(sizes=sizes, losses=losses, gpre=gpre, gmul=gmul, gclip=gclip, gnorm=gnorm, coef=coef,
                        params={k: p.detach().clone() for k, p in params.items()}, before=before,
                        m={k: t.clone() for k, t in m.items()}, v={k: t.clone() for k, t in v.items()}))
    return keys, rec


def test_update_arithmetic_matches_reference_recording():
    g = load_golden("trainstep")
    keys, rec = run_oracle_steps(g)
    assert list(g["hyper"]) == [TC.HYPER["lr"], *TC.HYPER["betas"], TC.HYPER["eps"], TC.HYPER["weight_decay"], TC.HYPER["clip_norm"]]
    for step, r in enumerate(rec):
        p = f"s{step}."
        assert [float(x) for x in r["sizes"]] == list(g[p + "task_sample_size"])
        np.testing.assert_allclose(r["losses"], g[p + "task_loss"], rtol=2e-5)
        assert abs(r["gnorm"] - float(g[p + "gnorm"][0])) <= 2e-5 * r["gnorm"]
        assert abs(r["coef"] - float(g[p + "clip_coef"][0])) <= 2e-5 * r["coef"]
        assert r["coef"] < 1.0                                   # the clip is active in this fixture
        for i, k in enumerate(keys):
            want = float(g[p + "grad_pre_norms"][i])
            if want < 0:                                         # unused parameter: no gradient, no update, no Adam state
                assert r["gpre"][k] is None
                assert float(g[p + "update_norms"][i]) == 0.0 and torch.equal(r["params"][k], r["before"][k])
                continue
            got = float(r["gpre"][k].double().norm())
            assert abs(got - want) <= 1e-4 * want + 1e-5, (k, got, want)
            for name, mine in (("exp_avg_norms", r["m"][k]), ("exp_avg_sq_norms", r["v"][k])):
                want = float(g[p + name][i])
                assert abs(float(mine.double().norm()) - want) <= 2e-4 * want + 1e-12, (name, k)
        for k in TC.FULL:
            gm = float(np.abs(g[p + "grad_clip." + k]).max())
            for tag, mine, tol in (("grad_pre.", r["gpre"][k], 1e-4), ("grad_mul.", r["gmul"][k], 1e-4),
                                   ("grad_clip.", r["gclip"][k], 1e-4), ("exp_avg.", r["m"][k], 1e-4),
                                   ("exp_avg_sq.", r["v"][k], 2e-4)):
                assert rel_err(TC.sample(mine), g[p + tag + k]) < tol, (tag, k)
            # the update m/(sqrt(v)+eps) is ill-conditioned where the gradient is rounding noise: compare where it is determined
            want, mine, gc = g[p + "param." + k], TC.sample(r["params"][k]).numpy(), g[p + "grad_clip." + k]
            sel = np.abs(gc) > 1e-3 * gm
            assert sel.sum() > 0
            assert np.abs(want - mine)[sel].max() <= 2e-3 * TC.HYPER["lr"], k


def test_dynamic_loss_scaler_restatement_matches_reference_recording():
    """oracle/restate.loss_scaler_step against tests/golden/loss_scaler.json (the reference's DynamicLossScaler run inside the
    fp16 optimizer's clip / step arithmetic, oracle/gen_scaler_golden.py)."""
    import json
    import os
    from oracle import scaler_cases as SC
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "loss_scaler.json")))
    for name, c in SC.CASES.items():
        st = restate.new_loss_scaler(c["init_scale"])
        for (raw, n), want in zip(c["seq"], g[name]):
            status, mf = restate.loss_scaler_step(st, SC.raw_value(raw), n, c["clip"], c["scale_factor"], c["scale_window"],
                                                  c["tolerance"], c["threshold"], c["min_loss_scale"])
            assert status == want["status"] and st["loss_scale"] == want["loss_scale"] and st["iter"] == want["iter"], (name, want)
            assert abs(mf - want["multiply_factor"]) <= 1e-12 * abs(want["multiply_factor"])
