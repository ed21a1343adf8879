#!/usr/bin/env python
"""Which lines of ofasys_amd/ launch the torch-native kernels of a train step (copies, adds, fills, index kernels)?

    python tools/native_glue_trace.py [cfg2|cfg2b|cfg4]

Runs two eager steps of the bench workload to settle caches, then one under a TorchDispatchMode and prints, per (aten op, innermost
ofasys_amd frame), how many times it ran (each is one ~3-7 us launch inside the step graph) with its most frequent operand shapes.  Kernels of the package's own
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
    # attribute every aten call that launches a device kernel to the innermost ofasys_amd frame: a TorchDispatchMode sees the
    # forward and (through the autograd engine's thread-local state) the custom Functions' backward bodies
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    skip = ("aten.view", "aten._unsafe_view", "aten.as_strided", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute",
            "aten.expand", "aten.select", "aten.slice", "aten.unsqueeze", "aten.squeeze", "aten.empty", "aten.reshape", "aten.unbind",
            "aten.split", "aten.stride", "aten.sym_", "aten.size", "aten.is_", "aten._local_scalar", "aten.lift", "aten.unflatten",
            "aten.new_empty", "aten.empty_like", "aten.narrow", "aten.chunk", "aten.view_as", "aten.record_stream", "aten.resize_")
    agg = collections.Counter()

    class Tap(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not name.startswith(skip):
                frame = "?"
                for f in reversed(traceback.extract_stack(limit=40)):
                    if "ofasys_amd/" in f.filename and "native_glue" not in f.filename:
                        frame = f"{f.filename[f.filename.find('ofasys_amd/'):]}:{f.lineno} {f.name}"
                        break
                shape = ""
                for x in list(args) + list((kwargs or {}).values()):
                    if torch.is_tensor(x):
                        shape = f"{tuple(x.shape)} {str(x.dtype).replace('torch.', '')}"
                        break
                agg[(name, frame, shape)] += 1
            return func(*args, **(kwargs or {}))

    with Tap():
        trainer.train_step([batch])
    torch.cuda.synchronize()
    per_site = collections.Counter()
    for (name, frame, shape), n in agg.items():
        per_site[(name, frame)] += n
    print(f"# {workload}: {sum(agg.values())} aten calls with a device kernel in one eager train step (views / allocations excluded)")
    for (name, frame), n in per_site.most_common(70):
        shapes = sorted(((s, c) for (nm, fr, s), c in agg.items() if nm == name and fr == frame), key=lambda t: -t[1])[:3]
        print(f"{n:5d}  {name:32s} {frame:64s} " + " | ".join(f"{c}x{s}" for s, c in shapes))


if __name__ == "__main__":
    main()
