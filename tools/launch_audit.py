"""Which Python call sites launch the small torch-native kernels of one eager cfg-2 train step?  torch.profiler with
stacks; prints aten ops that launched device kernels, grouped by the innermost frame inside the repo."""
import os, sys, collections, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ofasys_amd.trainer import Trainer
args = argparse.Namespace(arch="base", workload="cfg2", batch=32)
dev = torch.device("cuda", 0)
model, d = bench.build(args, dev)
tr = Trainer(model, lr=1e-4, clip_norm=1.0, use_graph=False)
batch, ntok, _ = bench.make_batch(d, 32, 191, 64, 0, dev, "cfg2")
for _ in range(3):
    tr.train_step([batch])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
import traceback
sites = collections.defaultdict(list)
class Spy(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        big = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or (isinstance(out, torch.Tensor) and out.is_cuda)
        if big and not any(s in name for s in ("view", "reshape", "transpose", "slice", "select", "expand", "unsqueeze", "squeeze", "permute",
                                               "detach", "alias", "as_strided", "empty", "t.default", "_unsafe_view", "unbind", "split", "size", "stride")):
            fr = [f for f in traceback.extract_stack() if "/ofasys_amd/" in f.filename or f.filename.endswith("trainer.py")]
            site = f"{fr[-1].filename.split('ofasys_amd/')[-1]}:{fr[-1].lineno}" if fr else "?"
            sites[(str(func), site)].append(1)
        return out
import threading
with Spy():
    tr.train_step([batch])
    torch.cuda.synchronize()
print("---- aten calls on CUDA tensors seen from Python (forward + python-side backward on this thread) ----")
for (name, site), v in sorted(sites.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(v):4d}  {name:40s} {site}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step([batch])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    kern = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not kern:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "/ofasys_amd/" in fr or "bench.py" in fr or "trainer.py" in fr:
            site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            site = site.split("repo/")[-1]
            break
    a = agg[(ev.name, site[-90:])]
    a[0] += len(kern)
    a[1] += sum(k.duration for k in kern)
tot = sum(v[0] for v in agg.values())
print("torch-native kernel launches in one step:", tot, " device us:", sum(v[1] for v in agg.values()))
for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:4d} launches {us:8.1f} us  {name:28s} {site}")
