#!/usr/bin/env python
"""Pointer audit of the replayed step graphs: which device memory does a captured step address that neither
the graphs' private pool nor the engine owns -- and does the step survive losing it?

    python tools/capture_audit.py --workload cfg2b [--batch 32] [--rounds 2] [--stress] [--frames]

Flow = bench.py's set-up: four batches of the workload, each primed by `graph_warmup` eager steps and then captured (so the eager
steps of structure k + 1 run AFTER structure k's capture, and every capture begins with the allocator flush that unmaps whatever
the default pool holds free).  After every capture the tool prints
  * FOREIGN PINS: default-pool tensors the capture's kernels address that are not parameters / buffers / arenas / optimizer state /
    static inputs (lib.PointerAudit; each is kept alive by the graph entry's `pins` -- with OFA_CAPTURE_PINS=0 it is not, the
    negative control), with the C-ABI call or aten op that first used it;
  * SURVIVORS: blocks of the graphs' private pool still allocated after the capture (memory a Python object keeps from inside a
    capture: legal only for scratch that every graph rewrites before reading), with the allocating frames under --frames.
Then the graphs are replayed round-robin.  --stress: between the rounds every clearable cache of the package is dropped
(ops.cached_index, SegmentPlan, scratch buffers, the decoder's future mask), the garbage collector and torch.cuda.empty_cache() run --
a dangling default-pool pointer then faults on the next replay deterministically (the process dies with a GPU memory access fault:
run under `timeout`).
"""
import argparse
import gc
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def survivors(pool_id, frames):
    out = []
    for seg in torch.cuda.memory_snapshot():
        if tuple(seg.get("segment_pool_id", (0, 0))) != tuple(pool_id):
            continue
        for b in seg["blocks"]:
            if b["state"] == "active_allocated":
                where = ""
                if frames and b.get("frames"):
                    fr = [f for f in b["frames"] if "ofasys_amd" in f.get("filename", "") or "bench.py" in f.get("filename", "")]
                    where = " <- " + " <- ".join(f"{os.path.basename(f['filename'])}:{f['line']}" for f in fr[:4])
                out.append((b["size"], where))
    return out


def drop_caches(model):
    from ofasys_amd import kernels as K
    from ofasys_amd import ops
    n = 0
    for m in model.modules():
        c = m.__dict__.get("_ofa_index_cache")
        if c:
            n += len(c)
            c.clear()
        if hasattr(m, "_future_mask") and torch.is_tensor(m._future_mask) and m._future_mask.numel():
            m._future_mask = torch.empty(0)
            n += 1
    n += len(ops.SegmentPlan._cache)
    ops.SegmentPlan._cache.clear()
    n += len(K._ws_cache)
    K._ws_cache.clear()
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2b")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--arch", default=None)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--stress", action="store_true")
    ap.add_argument("--frames", action="store_true", help="record allocation stacks (slower) so that survivors name their call site")
    ap.add_argument("--no-pack", action="store_true")
    ap.add_argument("--dump-before-replay", default=None,
                    help="pickle torch.cuda.memory._snapshot() (segments + the alloc / free / segment trace with Python stacks) to this path in "
                         "front of the FIRST replay of every captured structure: the post-mortem of a replay that faults "
                         "(tools/r6/postmortem.py <pickle> <fault address>) then names the allocations next to the faulting address")
    a = ap.parse_args()
    import bench
    args = types.SimpleNamespace(workload=a.workload, arch=a.arch, batch=a.batch, dtype="bf16", dropout=None)
    if args.batch is None:
        args.batch = {"cfg4": 4, "cfg5": 4, "cfg3": 8}.get(a.workload, 32)
    if args.arch is None:
        args.arch = "large" if a.workload == "cfg5" else "base"
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if a.frames or a.dump_before_replay:
        torch.cuda.memory._record_memory_history(max_entries=3000000)
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    model, d = bench.build(args, device)
    ops.manual_seed(1)
    tr = TrainStep(model, lr=1e-4, clip_norm=1.0, use_graph=True)
    if a.dump_before_replay:
        import pickle
        inner = tr._replay

        def replay(entry, samples):
            if not entry.get("_dumped"):
                entry["_dumped"] = True
                torch.cuda.synchronize()
                snap = torch.cuda.memory._snapshot()
                with open(a.dump_before_replay, "wb") as f:
                    pickle.dump({"segments": snap["segments"], "device_traces": snap["device_traces"], "pool": tuple(tr._pool)}, f)
                print(f"   (memory trace dumped in front of the first replay of structure {tr.captured_graphs()}: "
                      f"{sum(len(t) for t in snap['device_traces'])} events)", flush=True)
            return inner(entry, samples)
        tr._replay = replay
    packed = a.workload in ("cfg2", "cfg2b", "cfg3", "cfg5") and not a.no_pack
    batches = [bench.make_step(d, args, args.batch, 97 * i, device, packed) for i in range(4)]
    total_foreign = 0
    for i, b in enumerate(batches):
        for _ in range(tr.graph_warmup + 1):
            tr.train_step(b[0])
        torch.cuda.synchronize()
        ent = [e for e in tr._graphs.values() if "graphs" in e]
        print(f"== batch {i}: {len(ent)} structure(s) captured so far", flush=True)
        for j, e in enumerate(ent):
            if e.get("_reported"):
                continue
            e["_reported"] = True
            rep = tr.audit_report(e)
            total_foreign += len(rep)
            print(f"   structure {j}: {len(e['audit'].pins)} default-pool tensors addressed, {len(rep)} FOREIGN (pinned: {'pins' in e})")
            for name, shape, dt, nb in sorted(rep, key=lambda r: -r[3]):
                print(f"      foreign  {name:28s} {str(shape):22s} {dt:10s} storage {nb} B")
            sv = survivors(tr._pool, a.frames)
            print(f"   private-pool survivors after this capture: {len(sv)} blocks, {sum(s for s, _ in sv)} B")
            for size, where in sorted(sv, key=lambda r: -r[0])[:24]:
                print(f"      survivor {size:>12d} B{where}")
    print(f"TOTAL foreign pins: {total_foreign}", flush=True)
    for r in range(a.rounds):
        if a.stress and r > 0:
            n = drop_caches(model)
            print(f"stress: dropped {n} cache entries, gc, empty_cache()", flush=True)
        for i, b in enumerate(batches):
            tr.train_step(b[0])
            torch.cuda.synchronize()
        st = tr.last["stats"].tolist()
        print(f"round {r}: replayed {len(batches)} batches ok; loss/token {st[1] / max(st[0], 1):.4f} gnorm {float(tr.last['gnorm']):.4f}", flush=True)
    tr.check()
    print("AUDIT RUN COMPLETE")


if __name__ == "__main__":
    main()
