"""Generate tests/golden/trainstep.npz by RUNNING THE REFERENCE's own update arithmetic (build container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_trainstep_golden.py
One optimisation step of the reference trainer (engine/trainer.py:747-884) on tiny dims, 2 tasks x 2 micro-batches:
    for task: for micro-batch: loss = nll_loss(log_softmax(model(slots)), target, sum) ; loss.backward()   (grads accumulate;
        engine/criterion/cross_entropy.py:27-67, sample_size = non-pad targets)
    optimizer.multiply_grads(world / sum(task_sample_size))          (trainer.py:849-860; world = 1: no DDP pre-division)
    grad_norm = optimizer.clip_grad_norm(clip_norm)                  (FairseqOptimizer -> module/utils.py:342-384)
    optimizer.step()                                                 (engine/optim/adam.py:144-218, the class `Adam`)
run for STEPS consecutive updates with the reference's OWN FairseqOptimizer methods and Adam class.  Stored: per-task
sample_size, loss, gradient norms before the multiply, grad-norm, clip coefficient, per-parameter norms of the parameter
UPDATE and both Adam moments after each step, and the full tensors of oracle/trainstep_case.FULL.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe  # noqa: E402
from oracle import trainstep_case as TC  # noqa: E402
from oracle.cases import VOCAB_EXTRA  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "trainstep.npz")


def main():
    install()
    import ofasys  # noqa: F401
    from ofasys import ModalityType
    from ofasys.preprocessor import Slot
    from ofasys.engine.criterion.cross_entropy import nll_loss
    from ofasys.engine.optim.adam import Adam
    from ofasys.engine.optim.fairseq_optimizer import FairseqOptimizer

    torch.set_num_threads(8)
    model, d = build_reference_model(TC.ARCH, VOCAB_EXTRA, TC.ACTIVE, TC.OVERRIDES, {})
    recipe.fill_state(model.state_dict())
    model.train()
    V = len(d)
    named = dict(model.named_parameters())           # shared embedding appears once (encoder.adaptor.embed_tokens.weight)
    keys = sorted(named.keys())
    h = TC.HYPER
    opt = FairseqOptimizer(None)                      # the reference's multiply_grads / clip_grad_norm live on this class
    opt._optimizer = Adam([named[k] for k in keys], lr=h["lr"], betas=h["betas"], eps=h["eps"], weight_decay=h["weight_decay"])

    out = {"param_keys": np.array(keys), "hyper": np.array([h["lr"], h["betas"][0], h["betas"][1], h["eps"], h["weight_decay"],
                                                            h["clip_norm"]], dtype=np.float64)}
    world = 1
    for step in range(TC.STEPS):
        before = {k: named[k].detach().clone() for k in keys}
        opt.zero_grad()
        task_sample_size, task_loss = [], []
        for task in TC.TASKS:
            sample_size, loss_sum = 0, 0.0
            for specs in task:
                vals, target = TC.micro_batch(specs, V)
                slots = [Slot(ModalityType[m], s, v, attributes=a) for m, s, v, a in vals]
                logits, extra = model(slots)
                lprobs = model.get_normalized_probs((logits, extra), log_probs=True)
                loss = nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=d.pad(), reduce=True)
                opt.backward(loss)
                sample_size += int(target.ne(d.pad()).sum())
                loss_sum += float(loss)
            task_sample_size.append(float(sample_size))
            task_loss.append(loss_sum)
        p = f"s{step}."
        out[p + "task_sample_size"] = np.array(task_sample_size)
        out[p + "task_loss"] = np.array(task_loss)
        gpre = {k: (named[k].grad.detach().clone() if named[k].grad is not None else None) for k in keys}
        out[p + "grad_pre_norms"] = np.array([-1.0 if gpre[k] is None else float(gpre[k].double().norm()) for k in keys])
        opt.multiply_grads(world / (sum(task_sample_size) or 1.0))
        gmul = {k: (named[k].grad.detach().clone() if named[k].grad is not None else None) for k in keys}
        gnorm = opt.clip_grad_norm(h["clip_norm"])
        out[p + "gnorm"] = np.array([float(gnorm)])
        out[p + "clip_coef"] = np.array([min(1.0, h["clip_norm"] / (float(gnorm) + 1e-6))])
        gclip = {k: (named[k].grad.detach().clone() if named[k].grad is not None else None) for k in keys}
        opt.step()
        st = opt.optimizer.state
        out[p + "update_norms"] = np.array([float((named[k].detach() - before[k]).double().norm()) for k in keys])
        out[p + "exp_avg_norms"] = np.array([float(st[named[k]]["exp_avg"].double().norm()) if named[k] in st else -1.0 for k in keys])
        out[p + "exp_avg_sq_norms"] = np.array([float(st[named[k]]["exp_avg_sq"].double().norm()) if named[k] in st else -1.0 for k in keys])
        for k in TC.FULL:
            for tag, t in (("grad_pre.", gpre[k]), ("grad_mul.", gmul[k]), ("grad_clip.", gclip[k]), ("param.", named[k].detach()),
                           ("exp_avg.", st[named[k]]["exp_avg"]), ("exp_avg_sq.", st[named[k]]["exp_avg_sq"])):
                out[p + tag + k] = TC.sample(t).numpy().copy()
        print(f"step {step}: sample_size {task_sample_size} loss {task_loss} gnorm {float(gnorm):.6f} "
              f"coef {float(out[p + 'clip_coef'][0]):.6f} unused {sum(1 for k in keys if gpre[k] is None)}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
