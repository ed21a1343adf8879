"""Generate tests/golden/loss_scaler.json by RUNNING THE REFERENCE's DynamicLossScaler (engine/optim/dynamic_loss_scaler.py)
inside the fp16 optimizer's own update arithmetic (engine/optim/fp16_optimizer.py:170-204, 228-229) on the scripted sequences of
oracle/scaler_cases.py (build container only).  TEST INFRASTRUCTURE: only data is stored."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scaler_cases as SC  # noqa: E402
from oracle.ref_import import install  # noqa: E402


def main():
    install()
    from ofasys.engine.optim.dynamic_loss_scaler import DynamicLossScaler
    out = {}
    for name, c in SC.CASES.items():
        sc = DynamicLossScaler(init_scale=c["init_scale"], scale_factor=c["scale_factor"], scale_window=c["scale_window"],
                               tolerance=c["tolerance"], threshold=c["threshold"], min_loss_scale=c["min_loss_scale"])
        rows = []
        for raw, n in c["seq"]:
            raw = SC.raw_value(raw)
            mf = 1.0 / float(sc.loss_scale)                  # zero_grad(): _multiply_factor = 1 / loss_scale   (:228-229)
            mf *= 1.0 / n                                     # trainer.py:857-860 multiply_grads(world / sample_size), world = 1
            grad_norm = mf * raw                              # clip_grad_norm (:178)
            status = "ok"
            if grad_norm > c["clip"] > 0.0:                   # (:181-182)
                mf *= c["clip"] / grad_norm
            try:
                sc.check_overflow(grad_norm)                  # (:184)
            except OverflowError:
                status = "overflow"
            except FloatingPointError:
                status = "fatal"
            if status == "ok":
                sc.update()                                   # step() (:201-202)
            rows.append({"status": status, "loss_scale": float(sc.loss_scale), "multiply_factor": mf if status == "ok" else 0.0,
                         "grad_norm": grad_norm if grad_norm == grad_norm and abs(grad_norm) != float("inf") else None,
                         "iter": sc._iter})
        out[name] = rows
        print(name, [(r["status"][0], r["loss_scale"]) for r in rows])
    path = os.path.join(ROOT, "tests", "golden", "loss_scaler.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
