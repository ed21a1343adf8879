#!/usr/bin/env python
"""Which lines of ofasys_amd/ launch the torch-native kernels of a train step (copies, adds, fills, index kernels)?

    python tools/native_glue_trace.py [cfg2|cfg2b|cfg4]

Runs two eager steps of the bench workload to settle caches, then one under torch.profiler with Python stacks, and prints, per
(aten op, innermost ofasys_amd frame), the number of device kernels and their summed device time.  Kernels of the package's own
library (ctypes launches) do not pass through aten and are not listed: this is the glue only."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2b"
    args = argparse.Namespace(arch="base", workload=workload, dtype="bf16", dropout=None)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    model, d = bench.build(args, device)
    ops.manual_seed(1)
    trainer = TrainStep(model, lr=1e-4, clip_norm=1.0, use_graph=False)
    Ts_text, Tt, nvis, desc = bench.WORKLOADS[workload]
    B = 4 if workload == "cfg4" else 32
    batch = bench.make_batch(d, B, Ts_text, Tt, 0, device, workload, pack=workload in ("cfg2", "cfg2b"))[0]
    for _ in range(2):
        trainer.train_step([batch])
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        trainer.train_step([batch])
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, ""])
    for e in prof.events():
        ks = getattr(e, "kernels", None)
        if not ks or not e.name.startswith("aten::"):
            continue
        frame = "?"
        for f in (e.stack or []):
            if "ofasys_amd/" in f or "bench.py" in f:
                frame = f[f.find("ofasys_amd/"):] if "ofasys_amd/" in f else f
                break
        a = agg[(e.name, frame)]
        a[0] += len(ks)
        a[1] += sum(k.duration for k in ks)
        a[2] = ks[0].name[:70]
    tot_n = sum(a[0] for a in agg.values())
    tot_t = sum(a[1] for a in agg.values())
    print(f"# {workload}: {tot_n} torch-native kernels, {tot_t / 1e3:.3f} ms of device time in one eager step")
    for (name, frame), (n, t, kn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:9.1f} us {n:5d}  {name:28s} {frame:70s} {kn}")


if __name__ == "__main__":
    main()
