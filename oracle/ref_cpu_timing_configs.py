"""TEST INFRASTRUCTURE (build container only): the REFERENCE ITSELF (stub-imported from /root/reference) timed per BASELINE.json
configuration on this host's CPU cores -- BASELINE.md section 3, step 1 -- with the inputs bench.py draws for the GPU line
(bench.make_step at a bounded micro-batch), forward + CE + backward in fp32, train mode, dropout 0.

Usage: python oracle/ref_cpu_timing_configs.py cfg1|cfg2|cfg2b|cfg3|cfg4|cfg5 [micro_batch] [timed_steps]
Prints one line: config, batch, threads, seconds per step, non-pad tokens per second."""
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402

wl = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"cfg1": 2, "cfg2": 4, "cfg2b": 4, "cfg3": 2, "cfg4": 1, "cfg5": 1}[wl]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
threads = min(os.cpu_count() or 8, 64)
torch.set_num_threads(threads)
install()
import ofasys  # noqa: E402,F401
from ofasys import ModalityType  # noqa: E402
from ofasys.preprocessor import Slot  # noqa: E402
from ofasys.engine.criterion.cross_entropy import nll_loss  # noqa: E402

arch = {"cfg1": "tiny", "cfg5": "large"}.get(wl, "base")
active = {"cfg1": {"text"}, "cfg2": {"text", "image_patch_embed"}, "cfg2b": {"text", "image_resnet"}, "cfg3": {"text", "image_resnet"},
          "cfg4": {"text", "image_resnet", "video_image_sequence"},
          "cfg5": {"text", "image_resnet", "video_image_sequence", "audio_fbank"}}[wl]
overrides = {"dropout": 0.0}
adaptor_overrides = {}
if wl == "cfg2":
    overrides.update(use_self_attn_bias=False, entangle_position_embedding=True)
    adaptor_overrides = {"text": {"entangle_position_embedding": True},
                         "image_patch_embed": {"entangle_position_embedding": True, "embed_dim": 768}}
# the bench's vocabulary size: 4 specials + 50260 <text>_i + <mask> + 1000 <bin>_i = 51265 symbols (the names do not matter here)
model, d = build_reference_model(arch, bench.V_TEXT + 1 + 1000, active, overrides, adaptor_overrides)
model.train()
from ofasys_amd import Dictionary  # noqa: E402  (only to draw the SAME synthetic batches bench.py draws: token ids, <bin> range)
d2 = Dictionary()
for i in range(bench.V_TEXT):
    d2.add_symbol(f"<text>_{i}")
d2.add_symbol("<mask>")
d2.add_bins(1000)
assert len(d2) == len(d)
bench._HALF_NOW[0] = torch.float32
args = SimpleNamespace(workload=wl, batch=B)
cpu = torch.device("cpu")
if wl == "cfg1":
    parts = [bench.make_micro(d2, "text", B, 0, cpu)]
    samples, ntok = [parts[0][0]], parts[0][1]
else:
    samples, ntok, _ = bench.make_step(d2, args, B, 0, cpu, packed=False)


def ref_slots(sample):
    out = []
    for sl in sample["slots"]:
        v = sl.value
        if isinstance(v, dict):
            v = {k: (t.float() if t.is_floating_point() else t) for k, t in v.items()}
        elif v.is_floating_point():
            v = v.float()
        out.append(Slot(ModalityType[sl.modality.name], sl.is_src, v, attributes=sl.attributes))
    return out


work = [(ref_slots(sm), sm["target"]) for sm in samples]


def step():
    model.zero_grad()
    total = 0.0
    for slots, target in work:                      # gradient accumulation over the step's micro-batches (engine/trainer.py:747-884)
        out = model(slots)
        lprobs = model.get_normalized_probs(out, log_probs=True)
        loss = nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=d.pad(), reduce=True)
        loss.backward()
        total += float(loss)
    return total


step()
t0 = time.time()
for _ in range(steps):
    loss = step()
dt = (time.time() - t0) / steps
print(f"{wl}: reference itself, arch {arch}, micro-batch {B} x {len(work)} micro-batch(es), fp32, {threads} threads: "
      f"{dt:.2f} s/step, {ntok / dt:.0f} non-pad tokens/s (loss {loss:.1f}; {steps} timed step(s) after 1 warm-up)")
