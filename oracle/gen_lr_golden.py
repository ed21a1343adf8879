"""Generate tests/golden/lr_schedule.json by RUNNING THE REFERENCE's ofa_polynomial_decay scheduler
(engine/lr/polynomial_decay_schedule.py) the way its trainer drives it (engine/trainer.py:457 build -> reinit(total, 0) ->
step_update(num_updates) after every completed update, :932, :1091-1094).  Build container only.  TEST INFRASTRUCTURE: only
data is stored (the learning rate each update runs with)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_import import install  # noqa: E402

CASES = {
    "default_100": dict(lr=1e-3, total=100, warmup_ratio=0.06, end=0.0, power=1.0),
    "tiny_total": dict(lr=5e-4, total=7, warmup_ratio=0.06, end=0.0, power=1.0),          # int(7 * .06) = 0 -> max(.., 1)
    "power2_end": dict(lr=2e-4, total=50, warmup_ratio=0.1, end=1e-5, power=2.0),
}


def _opt_class():
    from ofasys.engine.optim.fairseq_optimizer import FairseqOptimizer

    class _Opt(FairseqOptimizer):          # the scheduler insists on a FairseqOptimizer; only set_lr / get_lr are used
        def __init__(self):
            self.lr = None

        def set_lr(self, lr):
            self.lr = lr

        def get_lr(self):
            return self.lr
    return _Opt


def main():
    install()
    from ofasys.engine.lr.polynomial_decay_schedule import PolynomialDecayLRSchedule, PolynomialDecayLRScheduleConfig
    out = {}
    for name, c in CASES.items():
        cfg = PolynomialDecayLRScheduleConfig()
        cfg.lr, cfg.warmup_ratio, cfg.end_learning_rate, cfg.power = [c["lr"]], c["warmup_ratio"], c["end"], c["power"]
        cfg.total_num_update = c["total"]
        opt = _opt_class()()
        sched = PolynomialDecayLRSchedule(cfg, opt)
        sched.step_update(0)                       # trainer.py:457
        sched.reinit(c["total"], 0)                # trainer: total number of updates known once the data is
        lrs = []
        for done in range(c["total"] + 3):         # lr in force while update `done + 1` runs
            lrs.append(float(opt.get_lr()))
            sched.step_update(done + 1)            # set_num_updates(done + 1) after the update
        out[name] = dict(c, lrs=lrs)
        print(name, lrs[:4], "...", lrs[-4:])
    path = os.path.join(ROOT, "tests", "golden", "lr_schedule.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
