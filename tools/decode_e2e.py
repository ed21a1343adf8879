"""End-to-end incremental decoding latency at the cfg-2 model size (OFA-base, bf16): encoder once, then greedy steps with the
KV cache for `rows` = batch x beam rows -- eagerly launched vs replayed per-length hipGraphs (ofasys_amd.generator)."""
import sys, time, argparse, torch
sys.path.insert(0, '.')
import bench
from ofasys_amd.generator import StepDecoder
dev = torch.device("cuda")
args = argparse.Namespace(arch="base", workload="cfg2", batch=32)
model, d = bench.build(args, dev)
model.eval()
steps = 32
for rows in (32, 160):
    batch, _, _ = bench.make_batch(d, rows, 191, 8, 0, dev, "cfg2")
    src = [s for s in batch["slots"] if s.is_src]
    for use_graph in (False, True):
        dec = StepDecoder(model, steps, use_graph=use_graph)
        for rep in range(3):                      # graph mode: eager warm-up, capture, replay
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.begin(src)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nxt = torch.full((rows,), d.bos(), dtype=torch.long, device=dev)
            for t in range(steps):
                nxt = dec.step(nxt).argmax(-1)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"rows={rows:4d} {'hipGraph' if use_graph else 'eager   '}: encoder {1e3*(t1-t0):6.2f} ms, decode {(t2-t1)/steps*1e3:6.3f} ms/step, "
              f"{rows*steps/(t2-t1):9.0f} tokens/s")
