"""Kernel mix of replayed decode steps (rocprofv3 --kernel-trace --stats -- python tools/decode_profile.py)."""
import sys, argparse, torch
sys.path.insert(0, '.')
import bench
from ofasys_amd.generator import StepDecoder
dev = torch.device("cuda")
model, d = bench.build(argparse.Namespace(arch="base", workload="cfg2", batch=32), dev)
model.eval()
rows, steps = 160, 16
batch, _, _ = bench.make_batch(d, rows, 191, 8, 0, dev, "cfg2")
src = [s for s in batch["slots"] if s.is_src]
dec = StepDecoder(model, steps, use_graph=True)
for rep in range(6):                                  # 1 eager + 1 capture + 4 replayed sequences
    dec.begin(src)
    nxt = torch.full((rows,), d.bos(), dtype=torch.long, device=dev)
    for t in range(steps):
        nxt = dec.step(nxt).argmax(-1)
torch.cuda.synchronize()
