#!/usr/bin/env python
"""Headline benchmark: multimodal tokens/s for one encoder-decoder TRAINING step (fwd + CE loss + bwd + grad reduce +
clip + Adam) of the OFASys GeneralistModel on MI355X, BASELINE.json configs[1] ("cfg-2"):

    image_caption: image 224x224 through the image_patch_embed adaptor (257 tokens) + <=191 text tokens (Ts = 448),
    target <=64 tokens, OFA-base (D=768, 12 heads, 6+6 layers, V=51265), bf16, batch 32 per GPU, synthetic data.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run, one rank per GPU).
Prints ONE JSON line on rank 0.  `value` is whole-job tokens/s (non-pad encoder + decoder positions, SURVEY.md
section 8d); `roofline` prices the dominant kernel (the bf16 MFMA GEMM) against the 2.5 PFLOP/s dense bf16 peak;
`cpu_baseline` times the CPU oracle restatement (oracle/restate.py) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HALF = {"bf16": torch.bfloat16, "fp16": torch.float16}
_HALF_NOW = [torch.bfloat16]       # the 16-bit dtype of this run (--dtype)
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (2:1 sparsity excluded)
PEAK_HBM_GBPS = 8000.0        # HBM3E, same guide
V_TEXT = 50260                # 4 specials + 50260 <text>_i + <mask> + 1000 <bin>_i = 51265


def fwd_flops_per_sample(D, A, F, Le, Ld, Ts, Tt, V, patch_tokens=256, patch_k=588, bias=False):
    """Algorithmic forward FLOPs per sample, SURVEY.md section 8d formulas (multiply-add = 2)."""
    enc = 8 * Ts * D * D + 4 * Ts * Ts * D + 4 * Ts * D * F
    dec = 8 * Tt * D * D + 4 * Tt * Tt * D + 4 * Tt * D * D + 4 * Ts * D * D + 4 * Tt * Ts * D + 4 * Tt * D * F
    out = 2 * Tt * D * V
    adaptor = 2 * patch_tokens * patch_k * D
    pos = 0
    if bias:
        pos = (4 * Ts * D * D + 2 * Ts * Ts * D) + (4 * Tt * D * D + 2 * Tt * Tt * D + 2 * Tt * D * D + 2 * Ts * D * D + 2 * Tt * Ts * D)
    return Le * enc + Ld * dec + out + adaptor + pos


def build(args, device):
    from ofasys_amd import Dictionary, GeneralistModel
    d = Dictionary()
    for i in range(V_TEXT):
        d.add_symbol(f"<text>_{i}")
    d.add_symbol("<mask>")
    d.add_bins(1000)
    torch.manual_seed(1)
    m = GeneralistModel()
    m.cfg.arch = args.arch
    m.__init__(m.cfg)
    wl = getattr(args, "workload", "cfg2")
    if wl in ("cfg2b", "cfg3", "cfg4", "cfg5"):
        # cfg-2b / cfg-3 / cfg-4 / cfg-5: the default IMAGE / VIDEO / AUDIO adaptors (ResNet backbone) with the default
        # biased attention; the trunk is the arch's own (model/ofa.py:557-610: resnet101 for base, resnet152 for large)
        for name in ("text", "image_resnet") + (("video_image_sequence",) if wl in ("cfg4", "cfg5") else ()) + \
                (("audio_fbank",) if wl == "cfg5" else ()):
            getattr(m.cfg.adaptor, name).is_active = True
        if args.arch == "base":
            m.cfg.adaptor.image_resnet.resnet_type = "resnet101"
    else:
        m.cfg.use_self_attn_bias = False                  # the image_patch_embed adaptor's only working corner,
        m.cfg.entangle_position_embedding = True
        for name in ("text", "image_patch_embed"):
            a = getattr(m.cfg.adaptor, name)
            a.is_active = True
            a.entangle_position_embedding = True
        m.cfg.adaptor.image_patch_embed.embed_dim = m.cfg.encoder_embed_dim      # the adaptor's own default is 768 (base)
    if getattr(args, "dropout", None) is not None:        # (parity tests: the CPU oracle has no dropout; the bench keeps the default 0.1)
        m.cfg.dropout = float(args.dropout)
    m.initialize(d)
    m = m.to(device).to(HALF[getattr(args, 'dtype', 'bf16')])
    return m, d


def make_batch(d, B, Ts_text, Tt, rank, device, workload="cfg2", pack=False):
    """Synthetic instruction batch [IMAGE,adaptor=image_patch_embed][TEXT] -> [TEXT], seed 1234 + rank."""
    from ofasys_amd import ModalityType, Slot
    g = torch.Generator().manual_seed(1234 + rank)
    V = len(d)
    if workload == "cfg4":                                    # [VIDEO][TEXT] -> [TEXT]: 8 frames of 224 x 224, one zero frame in ~10% of rows
        img = torch.randn(B, 3, 8, 224, 224, generator=g)
        zero = torch.rand(B, generator=g) < 0.1
        zero[min(1, B - 1)] = True
        for b in range(B):
            if zero[b]:
                img[b, :, int(torch.randint(0, 8, (1,), generator=g))] = 0.0
        img = img.to(_HALF_NOW[0])
        nvis = B * 8 * 196 - int(zero.sum()) * 196
    else:
        img = torch.randn(B, 3, 224, 224, generator=g).to(_HALF_NOW[0])
    src = torch.randint(4, V, (B, Ts_text), generator=g)
    slen = torch.randint(Ts_text // 2, Ts_text + 1, (B,), generator=g)
    slen[0] = Ts_text
    prev = torch.randint(4, V, (B, Tt), generator=g)
    tlen = torch.randint(Tt // 4, Tt + 1, (B,), generator=g)
    tlen[0] = Tt
    prev[:, 0] = d.bos()
    target = torch.full_like(prev, d.pad())
    for b in range(B):
        src[b, slen[b]:] = d.pad()
        n = int(tlen[b])
        prev[b, n:] = d.pad()
        target[b, :n - 1] = prev[b, 1:n]
        target[b, n - 1] = d.eos()
    patch = workload == "cfg2"
    first = Slot(ModalityType.VIDEO, True, img.to(device)) if workload == "cfg4" else \
        Slot(ModalityType.IMAGE, True, img.to(device), attributes=["adaptor=image_patch_embed"] if patch else None)
    slots = [first, Slot(ModalityType.TEXT, True, src.to(device)), Slot(ModalityType.TEXT, False, prev.to(device))]
    ntok = (nvis if workload == "cfg4" else B * (257 if patch else 196)) + int(slen.sum()) + int(tlen.sum())
    sample = {"slots": slots, "target": target.to(device)}
    if pack:        # ragged row packing (ofasys_amd/packing.py): only the non-pad positions go through the stack
        from ofasys_amd.packing import build_pack_plan
        enc_mask = torch.cat([torch.zeros(B, 257 if patch else 196, dtype=torch.bool), src.eq(d.pad())], 1)
        plan = build_pack_plan(enc_mask, prev.eq(d.pad()), bucket=512, dec_bucket=256)
        # the targets by packed decoder row: collation work, done once per batch on the host (TrainStep gathers them on the device
        # every step when the sample does not bring them)
        idx = plan.dec_index
        sample["target_packed"] = target.reshape(-1)[idx.clamp_min(0)].masked_fill(idx < 0, d.pad()).view(1, -1).to(device)
        sample["pack"] = plan.to(device)
    return sample, ntok, (slen.tolist(), tlen.tolist())


def _ragged_tokens(d, B, T, g, lo_frac, bos=False, vocab_lo=4, vocab_hi=None):
    V = vocab_hi or len(d)
    t = torch.randint(vocab_lo, V, (B, T), generator=g)
    n = torch.randint(max(int(T * lo_frac), 1), T + 1, (B,), generator=g)
    n[0] = T
    if bos:
        t[:, 0] = d.bos()
    for b in range(B):
        t[b, n[b]:] = d.pad()
    return t, n


def _target_of(d, prev, tlen):
    target = torch.full_like(prev, d.pad())
    for b in range(prev.shape[0]):
        n = int(tlen[b])
        target[b, :n - 1] = prev[b, 1:n]
        target[b, n - 1] = d.eos()
    return target


def make_micro(d, kind, B, seed, device):
    """One micro-batch of a multi-task step (cfg-3 / cfg-5): (sample, non-pad tokens, forward-FLOP descriptor).  Kinds:
      text    [TEXT] -> [TEXT], 128 / 128 (cfg-1's text_infilling shapes)
      image   = the cfg-2b batch ([IMAGE via image_resnet][TEXT <= 252] -> [TEXT <= 64]), packed rows
      box     [IMAGE][BOX: 4 <bin> tokens][TEXT <= 32] -> [TEXT <= 32]   (boxes are vocabulary tokens, preprocessor/default/box.py)
      video   = the cfg-4 batch ([VIDEO 8 x 224 x 224][TEXT <= 32] -> [TEXT <= 32])
      audio   [AUDIO fbank [B, 400, 80], ragged lengths][TEXT <= 32] -> [TEXT <= 32]
      struct / motion   [STRUCT | MOTION as tokens, <= 64][TEXT <= 32] -> [TEXT <= 64]   (adaptor/general.py:36-46: the text adaptor)
    The descriptor is (visual tokens, trunk frames, audio frames, per-sample (encoder text+token length, target length) list)."""
    from ofasys_amd import ModalityType, Slot
    if kind == "image":
        s, ntok, (sls, tls) = make_batch(d, B, 252, 64, seed, device, "cfg2b", pack=True)
        return s, ntok, dict(nvis=196, frames=1, audio=0, lens=list(zip(sls, tls)), Ts=252, Tt=64)
    if kind == "video":
        s, ntok, (sls, tls) = make_batch(d, B, 32, 32, seed, device, "cfg4")
        return s, ntok, dict(nvis=1568, frames=8, audio=0, lens=[(32, 32)] * B, Ts=32, Tt=32)       # runs padded
    g = torch.Generator().manual_seed(4321 + seed)
    half = _HALF_NOW[0]
    if kind == "text":
        src, slen = _ragged_tokens(d, B, 128, g, 0.875)
        prev, tlen = _ragged_tokens(d, B, 128, g, 0.875, bos=True)
        slots = [Slot(ModalityType.TEXT, True, src.to(device)), Slot(ModalityType.TEXT, False, prev.to(device))]
        desc = dict(nvis=0, frames=0, audio=0, lens=[(128, 128)] * B, Ts=128, Tt=128)
        ntok = int(slen.sum()) + int(tlen.sum())
    elif kind == "box":
        img = torch.randn(B, 3, 224, 224, generator=g).to(half)
        first_bin = d.index("<bin>_0")
        box = torch.randint(first_bin, first_bin + 1000, (B, 4), generator=g)
        src, slen = _ragged_tokens(d, B, 32, g, 0.5)
        prev, tlen = _ragged_tokens(d, B, 32, g, 0.25, bos=True)
        slots = [Slot(ModalityType.IMAGE, True, img.to(device)), Slot(ModalityType.BOX, True, box.to(device)),
                 Slot(ModalityType.TEXT, True, src.to(device)), Slot(ModalityType.TEXT, False, prev.to(device))]
        desc = dict(nvis=196, frames=1, audio=0, lens=[(4 + 32, 32)] * B, Ts=36, Tt=32)
        ntok = B * (196 + 4) + int(slen.sum()) + int(tlen.sum())
    elif kind == "audio":
        T = 400
        flen = torch.randint(T // 2, T + 1, (B,), generator=g)
        flen[0] = T
        fb = torch.randn(B, T, 80, generator=g)
        for b in range(B):
            fb[b, flen[b]:] = 0.0
        t2 = ((T - 3) // 2 + 1 - 3) // 2 + 1
        value = {"fbank": fb.to(half).to(device), "fbank_lengths": flen.to(device),
                 "mask_indices": torch.zeros(B, t2, dtype=torch.bool, device=device)}
        src, slen = _ragged_tokens(d, B, 32, g, 0.5)
        prev, tlen = _ragged_tokens(d, B, 32, g, 0.25, bos=True)
        slots = [Slot(ModalityType.AUDIO, True, value), Slot(ModalityType.TEXT, True, src.to(device)),
                 Slot(ModalityType.TEXT, False, prev.to(device))]
        desc = dict(nvis=t2, frames=0, audio=1, lens=[(32, 32)] * B, Ts=32, Tt=32)
        alen = (((flen - 3) // 2 + 1) - 3) // 2 + 1
        ntok = int(alen.sum()) + int(slen.sum()) + int(tlen.sum())
    elif kind in ("struct", "motion"):
        mod = ModalityType.STRUCT if kind == "struct" else ModalityType.MOTION
        tok, mlen = _ragged_tokens(d, B, 64, g, 0.5)
        src, slen = _ragged_tokens(d, B, 32, g, 0.5)
        prev, tlen = _ragged_tokens(d, B, 64, g, 0.25, bos=True)
        slots = [Slot(mod, True, tok.to(device)), Slot(ModalityType.TEXT, True, src.to(device)),
                 Slot(ModalityType.TEXT, False, prev.to(device))]
        desc = dict(nvis=0, frames=0, audio=0, lens=[(64 + 32, 64)] * B, Ts=96, Tt=64)
        ntok = int(mlen.sum()) + int(slen.sum()) + int(tlen.sum())
    else:
        raise ValueError(kind)
    return {"slots": slots, "target": _target_of(d, prev, tlen).to(device), "task": kind}, ntok, desc


STEP_MICRO = {      # multi-task workloads: the micro-batches of ONE update (engine/trainer.py:747-884: every task's micro-batch, every step)
    "cfg3": ("image", "text"),
    "cfg5": ("text", "image", "box", "video", "audio", "struct", "motion"),
}
TRUNK_GMAC = {"resnet50": 4.1 * 0.75, "resnet101": 6.9, "resnet152": 10.7}     # stride-16 trunk (3 stages) per 224 x 224 frame, GMAC


def micro_fwd_flops(dims, V, desc, trunk_gmac, executed):
    """Forward FLOPs of one micro-batch by SURVEY 8d's formulas (+ the ResNet trunk, + the audio subsampling convolutions);
    executed: at every sample's own lengths where rows are packed, and the position-bias products once per batch (ops.SharedBias)."""
    D = dims[0]
    vis = desc["nvis"]
    per = 0.0
    if desc["frames"]:
        per += desc["frames"] * (2 * trunk_gmac * 1e9 + 2 * 196 * 1024 * D)
    if desc["audio"]:
        # Conv2d(1, D, 3, 2) on [400, 80] -> [199, 39]; Conv2d(D, D, 3, 2) -> [99, 19]; Linear(D * 19, D)   (module/subsample.py:11-63)
        per += 2 * 199 * 39 * 9 * D + 2 * 99 * 19 * 9 * D * D + 2 * 99 * 19 * D * D
    lens = desc["lens"] if executed else [(desc["Ts"], desc["Tt"])] * len(desc["lens"])
    tot = sum(fwd_flops_per_sample(*dims, vis + sl, tl, V, patch_tokens=0, bias=not executed) + per for sl, tl in lens)
    if executed:
        tot += fwd_flops_per_sample(*dims, vis + desc["Ts"], desc["Tt"], V, patch_tokens=0, bias=True) - \
            fwd_flops_per_sample(*dims, vis + desc["Ts"], desc["Tt"], V, patch_tokens=0, bias=False)
    return tot


CPU_SAMPLE = {"cfg2": (8, 6), "cfg2b": (4, 4), "cfg4": (1, 4), "cfg3": (2, 2), "cfg5": (1, 1)}      # workload -> (batch, timed steps) of the CPU leg: 10-30 s of host work


def cpu_baseline(args, model, d):
    """The CPU oracle (oracle/restate.py: plain torch fp32 restatement, verified against the reference's golden vectors) timed on
    this host for the SAME workload: the GPU model's own weights (held in fp32), a batch drawn by the same make_batch at a
    bounded size, forward + CE + backward."""
    from oracle import restate
    from oracle.restate import OConfig, OSlot
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    cfgm = model.cfg
    dims = dict(embed_dim=cfgm.encoder.embed_dim, ffn_dim=cfgm.encoder.ffn_embed_dim, heads=cfgm.encoder.attention_heads,
                enc_layers=cfgm.encoder.layers, dec_layers=cfgm.decoder.layers)
    if args.workload == "cfg2":
        cfg = OConfig(**dims, use_self_attn_bias=False, entangle_position_embedding=True,
                      adaptor_entangle={"text": True, "image_patch_embed": True})
    else:                                                                   # BatchNorm on batch statistics (train mode)
        rl = {"resnet50": (3, 4, 6), "resnet101": (3, 4, 23), "resnet152": (3, 8, 36)}[cfgm.adaptor.image_resnet.resnet_type]
        cfg = OConfig(**dims, resnet_layers=rl, training=True)
    st = {}
    for k, v in model.state_dict().items():
        v = v.detach().cpu()
        st[k] = v.float().requires_grad_(True) if (v.is_floating_point() and not k.endswith(("version", "running_mean", "running_var"))) \
            else (v.float() if v.is_floating_point() else v.clone())
    st["decoder.adaptor.embed_tokens.weight"] = st["encoder.adaptor.embed_tokens.weight"]
    B, steps = CPU_SAMPLE[args.workload]
    B = args.cpu_batch or B
    steps = args.cpu_steps or steps
    samples, ntok, _ = make_step(d, args, B, 0, torch.device("cpu"), packed=False)

    def oslots(sample):
        out = []
        for sl in sample["slots"]:
            v = sl.value
            if isinstance(v, dict):
                v = {k: (t.float() if t.is_floating_point() else t) for k, t in v.items()}
            elif v.is_floating_point():
                v = v.float()
            out.append(OSlot(sl.modality.name, sl.is_src, v, sl.attributes))
        return out
    work = [(oslots(sm), sm["target"]) for sm in samples]
    params = [v for v in st.values() if v.requires_grad]

    def step():
        for p in params:
            p.grad = None
        for slots, target in work:                      # gradient accumulation over the step's micro-batches
            logits, _ = restate.model_forward(st, cfg, slots)
            loss, _ = restate.cross_entropy(logits, target)
            loss.backward()
    step()
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = (time.time() - t0) / steps
    shape = WORKLOADS[args.workload][3]
    return {"value": ntok / dt, "unit": "tokens/s", "cores": cores, "kind": "port", "batch": B, "gpu_batch": args.batch,
            "s_per_sample": dt / (B * len(work)),
            "sample": f"oracle/restate.py fp32, {args.workload}, micro-batch {B} x {len(work)} micro-batch(es) (the GPU line runs "
                      f"micro-batch {args.batch}), the padded shapes computed in full as the reference does ({shape}), non-pad tokens "
                      f"counted like `value`; fwd+CE+bwd, {steps} timed step(s) after 1 warm-up, torch.set_num_threads({cores})",
            "s_per_step": dt}


def make_step(d, args, B, seed, device, packed):
    """The micro-batches of ONE update of the workload: (samples, non-pad tokens, descriptors)."""
    if args.workload in STEP_MICRO:
        parts = [make_micro(d, kind, B, seed + 1009 * i, device) for i, kind in enumerate(STEP_MICRO[args.workload])]
        if not packed:
            for p in parts:
                p[0].pop("pack", None)
        return [p[0] for p in parts], sum(p[1] for p in parts), [p[2] for p in parts]
    Ts_text, Tt, _, _ = WORKLOADS[args.workload]
    s1, ntok, lens = make_batch(d, B, Ts_text, Tt, seed, device, args.workload, pack=packed)
    return [s1], ntok, lens


def source_id():
    """Identity of the product sources this process runs (tools/build_id.py): stamped into the line, and what a committed profile
    summary must carry to be quoted by it."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "tools"))
    try:
        from build_id import source_id as sid
        return sid(here)
    finally:
        sys.path.pop(0)


PROFILE_ROUNDS = ("round6", "round5", "round4", "round3", "round2", "round1")


def pmc_traffic(workload="cfg2"):
    """(HBM bytes per launch of the MFMA GEMM kernels, HBM bytes of one whole step, source) from the committed rocprofv3 PMC
    passes of THIS workload's eager step (profiles/, produced by tools/collect_profiles.sh: separate FETCH_SIZE and WRITE_SIZE passes,
    FETCH_SIZE doubled on gfx950, tools/pmc_traffic.py); (None, None, reason) when no pass of the workload is committed."""
    here = os.path.dirname(os.path.abspath(__file__))
    tag = "" if workload == "cfg2" else "_" + workload
    me = source_id()
    for name in [f"{r}_pmc_traffic{tag}.json" for r in PROFILE_ROUNDS]:
        try:
            rows = json.load(open(os.path.join(here, "profiles", name)))
        except (OSError, ValueError):
            continue
        meta = rows.pop("__meta__", {})
        if meta.get("source_id") != me:
            # the NEWEST committed pass describes other sources (round 5's line quoted a mid-round pass that still counted buffers the final
            # build had removed: VERDICT r5 weak 7): no number is better than that one
            return None, None, (f"refused: profiles/{name} was taken on sources {meta.get('source_id', '(unstamped)')}, this run is {me} "
                                "(tools/build_id.py) -- re-run tools/collect_profiles.sh on this build")
        n = tot = 0.0
        for kname, r in rows.items():
            if "gemm_" in kname and "_kernel" in kname and "simple" not in kname:
                n += r["launches"]
                tot += r["launches"] * (r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"])
        if n:
            return tot / n, meta.get("bytes_per_step"), (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per launch, all MFMA "
                                                         f"GEMM instantiations of the eager {workload} step)")
    return None, None, f"none (no PMC pass of the {workload} step is committed under profiles/)"


def rocprof_gemm_ms(workload):
    """(ms per step of the MFMA GEMM kernel family incl. split-K reduces and slab folds, ms per step of ALL kernels, file) from the
    committed rocprofv3 kernel trace of this very command -- the REPLAYED hipGraph's kernels, timed by the profiler, not by HIP
    events around relaunches (tools/collect_profiles.sh -> tools/prof_summary.py --json)."""
    here = os.path.dirname(os.path.abspath(__file__))
    stem = {"cfg2": "rocprof_kernel_stats.json", "cfg2b": "rocprof_cfg2b_kernel_stats.json", "cfg4": "rocprof_cfg4_kernel_stats.json",
            "cfg3": "rocprof_cfg3_kernel_stats.json", "cfg5": "rocprof_cfg5_kernel_stats.json"}[workload]
    rows = name = None
    for rnd in PROFILE_ROUNDS:                                       # the newest committed trace of this command
        try:
            rows = json.load(open(os.path.join(here, "profiles", rnd + "_" + stem)))
            name = rnd + "_" + stem
            break
        except (OSError, ValueError):
            continue
    if rows is None:
        return None, None, None
    meta = rows.pop("__meta__", {})
    if meta.get("source_id") != source_id():                         # a trace of other sources is not this command's trace
        return None, None, f"refused: profiles/{name} was taken on sources {meta.get('source_id', '(unstamped)')}, this run is {source_id()}"
    fam = sum(r["ms_per_step"] for k, r in rows.items()
              if ("gemm_" in k and "simple" not in k) or "splitk_reduce" in k or "fold_batched" in k)
    return fam, meta.get("total_ms_per_step"), f"profiles/{name}"


WORKLOADS = {   # name -> (source text length, target length, visual tokens per sample, description)
    "cfg2": (191, 64, 257, "cfg-2 image_caption: image_patch_embed 224x224 (257 tok) + text<=191 -> text<=64"),
    "cfg2b": (252, 64, 196, "cfg-2b image_caption: image_resnet101 224x224 (196 tok, rel-pos biased attention) + text<=252 -> text<=64"),
    "cfg4": (32, 32, 1568, "cfg-4 video_caption: 8 frames 224x224 through image_resnet101 (1568 tok, frame+image rel-pos bias) + "
                           "text<=32 -> text<=32"),
    # multi-task steps (STEP_MICRO): every micro-batch has its own shapes; the three numbers are the largest micro-batch's
    "cfg3": (252, 128, 196, "cfg-3 two-task step (scripts/trainer_api.py:22-27): caption micro-batch = cfg-2b (image_resnet101 196 tok + "
                            "text<=252 -> text<=64, packed rows) + text_infilling micro-batch 128 -> 128, both every update"),
    "cfg5": (96, 128, 1568, "cfg-5 seven-modality step on OFA-large (resnet152 trunk): 7 micro-batches per update -- text 128->128, image "
                            "(196+<=252 -> <=64), image+box tokens, video 8x224x224 (1568 tok), audio fbank [400,80] (99 tok), "
                            "struct-as-tokens, motion-as-tokens -- gradient accumulation over all of them, ragged slot collation"),
}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU, 127.0.0.1 rendezvous."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: what RCCL needs on this host driver
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def dp_selftest_child():
    """Child process of `captured_collectives_ok` (one per rank, its own rendezvous): capture an async RCCL all-reduce the way the step
    graph does (launched on a side handle inside the capture, waited for inside it), replay it twice on fresh data, check the sums."""
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist.init_process_group(backend="nccl", device_id=device)
    a = torch.zeros(1 << 20, device=device, dtype=torch.bfloat16)
    b = torch.zeros(3, device=device, dtype=torch.float64)
    dist.all_reduce(a)                                               # communicator warm-up, eager
    dist.all_reduce(b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        h = dist.all_reduce(a, op=dist.ReduceOp.SUM, async_op=True)  # a gradient bucket: launched from inside "backward" ...
        a2 = a * 1.0
        h.wait()                                                     # ... waited for before the optimizer reads it
        dist.all_reduce(b)                                           # the step's scalar exchange
        out = a + a2 * 0.0
    ok = True
    for it in range(2):
        a.fill_(float(rank + 1 + it))
        b.fill_(float(rank + it))
        g.replay()
        torch.cuda.synchronize()
        want_a = sum(r + 1 + it for r in range(world))
        want_b = sum(r + it for r in range(world))
        ok = ok and bool((out.float() == want_a).all()) and bool((b == want_b).all())
    dist.destroy_process_group()
    raise SystemExit(0 if ok else 3)


def captured_collectives_ok(world, rank, local_rank, timeout=180):
    """Can THIS node replay RCCL collectives captured in a hipGraph with `world` ranks?  Asked of a CHILD process per rank (its own
    rendezvous on MASTER_PORT + 1): a capture that fails cannot be retried in the process it failed in (trainer.TrainStep docstring), so
    the question must not be asked in the process that then has to capture the real step.  Every rank's answer is folded (min) over
    the bench's own process group by the caller.  VERDICT r4 item 7c: `--gpus N` must not die in its first capture."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = str(rank), str(world), str(local_rank)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # the children rendezvous among themselves: rank 0's child hosts the store on the new port
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dp-selftest-child"], env=env, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if r.returncode != 0 and rank == 0:
            sys.stderr.write("bench: captured-collective self-test failed (rc %d): %s\n" % (r.returncode, r.stderr.decode(errors="replace")[-600:]))
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        return False


def main():
    if "--dp-selftest-child" in sys.argv:
        dp_selftest_child()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default=None, help="model size (default: base; large for cfg5)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU (micro-)batch (default 32; 8 for cfg3; 4 for cfg4 / cfg5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=0, help="CPU-leg batch (0: per workload, CPU_SAMPLE)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU-leg timed steps (0: per workload)")
    ap.add_argument("--profile-gemm", type=int, default=1, help="instrumented steps after the timed region")
    ap.add_argument("--profile-park-cycles", type=float, default=1.5e8,
                    help="spin-kernel cycles in front of the instrumented step (the host must finish enqueueing before the GPU starts)")
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS),
                    help="cfg2 (headline: image_patch_embed, bias-free), cfg2b (image_resnet101 + biased attention, 196+252 -> 64), "
                         "cfg4 (video 8x224x224 -> 1568 tokens + 32 text -> 32, micro-batch 4), cfg3 (two-task step: cfg2b + text "
                         "128/128, micro-batch 8 each) or cfg5 (OFA-large, seven modality micro-batches of 4 per update)")
    ap.add_argument("--no-pack", action="store_true",
                    help="cfg2 only: run the padded batch (every sample computed at 448 + 64 positions, as the reference does) instead "
                         "of packing the non-pad positions")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--dp-graph", default=None, choices=["full", "split"],
                    help="N > 1: 'full' (default) captures the bucketed RCCL all-reduces inside the step graph, overlapped with "
                         "backward; 'split' = two graphs around an eager all-reduce")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1: nccl (= RCCL, the product path); gloo only to smoke-test the N > 1 code path "
                         "on a one-GPU box together with OFA_BENCH_DEVICE=0 (all ranks on one device)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="compute dtype: bf16 (the BASELINE configuration) or fp16 with the reference's dynamic loss scaler "
                         "(its own default trainer precision)")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher plumbing only: init the process group, one all-reduce, print the JSON skeleton (runs without a GPU)")
    args = ap.parse_args()
    _HALF_NOW[0] = HALF[args.dtype]
    if args.batch is None:
        args.batch = {"cfg4": 4, "cfg5": 4, "cfg3": 8}.get(args.workload, 32)
    if args.arch is None:
        args.arch = "large" if args.workload == "cfg5" else "base"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch.distributed as dist
    if args.launch_check:
        if world > 1:
            dist.init_process_group(backend="gloo")
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": "launch-check", "n_gpus": int(t.item()), "world": world, "backend": "gloo"}))
        if world > 1:
            dist.destroy_process_group()
        return
    if "OFA_BENCH_DEVICE" in os.environ:                          # test hook: several ranks on one GPU (gloo only)
        local_rank = int(os.environ["OFA_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dp_selftest = None
    if world > 1:
        want_full = args.backend == "nccl" and not args.no_graph and args.dp_graph in (None, "full")
        if want_full and os.environ.get("OFA_BENCH_SKIP_DP_SELFTEST") != "1":
            # BEFORE this process touches RCCL or captures anything: can captured collectives replay here?  (a child process per rank)
            dp_selftest = captured_collectives_ok(world, rank, local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
        dist.all_reduce(torch.zeros(1, device=device))            # communicator warm-up (distributed/utils.py:240-241)
        if dp_selftest is not None:
            flag = torch.tensor([1.0 if dp_selftest else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)           # one rank's failure decides for all: the modes must agree
            dp_selftest = bool(flag.item() > 0.5)
            if not dp_selftest:
                if rank == 0:
                    sys.stderr.write("bench: captured RCCL collectives do not replay on this node -> dp_graph = split (two graphs around "
                                     "an eager bucket-by-bucket all-reduce)\n")
                args.dp_graph = "split"

    from ofasys_amd import kernels as K
    from ofasys_amd.trainer import TrainStep
    model, d = build(args, device)
    if world > 1:                                                  # identical initial weights on every rank
        for p in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(p.data, 0)
    from ofasys_amd import ops
    ops.manual_seed(1 + rank)                                      # dropout streams differ per rank (fairseq: seed + rank)
    trainer = TrainStep(model, lr=1e-4, clip_norm=1.0, use_graph=not args.no_graph, dp_graph=args.dp_graph,
                        loss_scale={"init_scale": 128.0} if args.dtype == "fp16" else None)
    Ts_text, Tt, nvis, desc = WORKLOADS[args.workload]
    # a few distinct batches of the same structure: replays copy each new batch into the graph's static inputs
    # ragged row packing: cfg-2 and (since round 3: the position bias lives inside the attention kernels) the default biased
    # configuration cfg-2b; cfg-4's padding sits in the MIDDLE of a row (all-zero frames), which the position-indexed rel-pos ids
    # of a packed row cannot express -- it runs padded
    packed = args.workload in ("cfg2", "cfg2b", "cfg3", "cfg5") and not args.no_pack      # (cfg3 / cfg5: their cfg-2b-shaped micro-batch)
    batches = [make_step(d, args, args.batch, rank + 97 * i, device, packed) for i in range(4)]
    ntok = sum(b[1] for b in batches) / len(batches)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_i = 0

    def step():
        nonlocal step_i
        trainer.train_step(batches[step_i % len(batches)][0])
        step_i += 1

    if trainer.use_graph:                   # setup: eager priming steps + the one-time graph capture of every batch STRUCTURE
        for b in batches:                   # (packed batches fall into a few row-count buckets, each with its own hipGraph)
            for _ in range(trainer.graph_warmup + 1):
                trainer.train_step(b[0])
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    toks = torch.tensor([float(ntok)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(toks, op=dist.ReduceOp.SUM)
    dt = float(tmax)
    total_tokens = float(toks)
    ms_per_step = dt / args.steps * 1e3
    graph_mode = "eager"
    for e in trainer._graphs.values():
        if "graphs" in e:
            graph_mode = {1: "one hipGraph (collectives captured)" if world > 1 else "one hipGraph", 2: "two hipGraphs + eager all-reduce"}[len(e["graphs"])]
    trainer.check()

    # data-parallel exchange, measured on every rank (the collectives need all of them): each bucket's all-reduce ALONE, and how
    # long the compute stream still waits for the exchange once backward has been enqueued in overlapped eager steps
    dp = None
    if world > 1:
        red = trainer.reducer
        alone = red.time_buckets_alone()
        red.profile, red.exposed_events = True, []
        for _ in range(3):
            trainer.train_step(batches[0][0], eager=True)
        torch.cuda.synchronize()
        exposed = [a.elapsed_time(b) for a, b in red.exposed_events]
        red.profile = False
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                                      # (gloo smoke runs)
            rccl = f"n/a ({type(e).__name__})"
        tot_alone = sum(alone)
        exp_ms = sum(exposed[1:]) / max(len(exposed) - 1, 1) if exposed else None
        dp = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
              "dp_graph": trainer.dp_graph, "captured_collective_selftest": dp_selftest, "buckets": len(red.buckets), "bucket_bytes": red.bucket_sizes(),
              "bucket_allreduce_ms_alone": alone, "allreduce_ms_alone_total": tot_alone,
              "buckets_launched_inside_backward": red.last_early, "launch_order": red.last_launch_order,
              "exposed_wait_ms_per_step": exp_ms,
              "overlap_frac": (1.0 - exp_ms / tot_alone) if (exp_ms is not None and tot_alone > 0) else None,
              "how": "bucket_allreduce_ms_alone: blocking all-reduce of each gradient-arena bucket with nothing else running (HIP "
                     "events, 3 reps); exposed_wait_ms_per_step: HIP events on the compute stream around the bucket waits at the end "
                     "of backward in overlapped EAGER steps (the timed region replays the same launches from one hipGraph); "
                     "overlap_frac = 1 - exposed / alone"}

    # dominant kernel: every MFMA GEMM launch of one step bracketed by HIP events on the launch stream
    prof = None
    if args.profile_gemm > 0:
        K.gemm_profile_begin()
        for _ in range(args.profile_gemm):
            # an eager step is host-bound (~600 launches): park the stream behind a spin kernel first so that every launch and its
            # two events are QUEUED when the GPU reaches them -- the kernels then run back to back exactly as in the replayed graph
            # and each event pair brackets one kernel, not the host's dispatch gap
            torch.cuda._sleep(int(args.profile_park_cycles))
            trainer.train_step(batches[0][0], eager=True)          # HIP events around each launch: not inside a graph
        torch.cuda.synchronize()
        prof = K.gemm_profile_end()
        # what an event pair costs by itself in this regime (dispatch of ONE kernel between two queued events): the same bracket around
        # a one-element reduction, 200 times, stream parked the same way
        nulls = []
        x1 = torch.zeros(1, device=device)
        torch.cuda._sleep(int(args.profile_park_cycles) // 8)
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.sum_f32(x1)
            e1.record()
            nulls.append((e0, e1))
        torch.cuda.synchronize()
        null_us = sorted(a.elapsed_time(b) * 1e3 for a, b in nulls)[len(nulls) // 2]
        prof["null_bracket_us"] = null_us

    if rank == 0:
        cfg = model.cfg
        dims = (cfg.encoder.embed_dim, cfg.encoder.attention_heads, cfg.encoder.ffn_embed_dim, cfg.encoder.layers, cfg.decoder.layers)
        fwd_exec = None
        if args.workload in STEP_MICRO:
            gm = TRUNK_GMAC[cfg.adaptor.image_resnet.resnet_type]
            fwd_step_padded = sum(micro_fwd_flops(dims, len(d), ds, gm, executed=False) for ds in batches[0][2])
            fwd_exec = sum(sum(micro_fwd_flops(dims, len(d), ds, gm, executed=True) for ds in b[2]) for b in batches) / len(batches)
            fwd = fwd_step_padded / args.batch               # (per "sample" = one row of every micro-batch)
        elif args.workload == "cfg2":
            fwd = fwd_flops_per_sample(*dims, 257 + Ts_text, Tt, len(d))
            # the same formulas at every sample's OWN lengths: what a ragged (packed) step actually has to compute
            fwd_exec = sum(fwd_flops_per_sample(*dims, 257 + sl, tl, len(d)) for _, _, (sls, tls) in batches
                           for sl, tl in zip(sls, tls)) / len(batches)
        else:          # ResNet-101 stride-16 trunk ~ 6.9 GMAC per 224x224 image + Linear(1024, D), biased attention
            frames = 8 if args.workload == "cfg4" else 1
            trunk = frames * (2 * 6.9e9 + 2 * 196 * 1024 * cfg.encoder.embed_dim)
            fwd = fwd_flops_per_sample(*dims, nvis + Ts_text, Tt, len(d), patch_tokens=0, bias=True) + trunk
            # what the step EXECUTES: every sample at its own lengths (packed), and the position-bias products ONCE per batch --
            # positions are the same for every sample, so pos_q / pos_k / pos_q pos_k^T are built from one row (ops.SharedBias)
            # where SURVEY 8d's formula, like the reference, counts them per sample
            pos_once = fwd_flops_per_sample(*dims, nvis + Ts_text, Tt, len(d), patch_tokens=0, bias=True) - \
                fwd_flops_per_sample(*dims, nvis + Ts_text, Tt, len(d), patch_tokens=0, bias=False)
            lens = [[(sl, tl) for sl, tl in zip(sls, tls)] if packed else [(Ts_text, Tt)] * args.batch for _, _, (sls, tls) in batches]
            fwd_exec = sum(sum(fwd_flops_per_sample(*dims, nvis + sl, tl, len(d), patch_tokens=0, bias=False) + trunk for sl, tl in ls)
                           + pos_once for ls in lens) / len(batches)
        step_flops_padded = 3 * fwd * args.batch                   # backward = 2x forward, padded shape
        # a packed step is priced at the flops of the positions it computes (never at the padded count it skips)
        step_flops = 3 * fwd_exec if fwd_exec and (packed or args.workload != "cfg2") else step_flops_padded
        step_tflops = step_flops / (ms_per_step * 1e-3) / 1e12
        traffic, step_bytes, traffic_src = pmc_traffic(args.workload)
        roof = {"bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "ofa::gemm_mfma_kernel + ofa::gemm_big_kernel + ofa::gemm_ring_kernel (" + ("bf16 v_mfma_f32_32x32x16_bf16" if args.dtype == "bf16" else "fp16 v_mfma_f32_32x32x16_f16") + ", all instantiations)",
                "step_achieved": step_tflops, "step_frac": step_tflops / PEAK_BF16_TFLOPS, "step_flops": step_flops,
                "step_flops_padded_shape": step_flops_padded,
                "step_flops_basis": (("non-pad positions only (ragged row packing: each sample at its own lengths)" if packed else
                                      "padded shape (every sample at the longest lengths)") +
                                     ("" if args.workload == "cfg2" else "; position-bias products counted once per batch (they are "
                                      "computed once: ops.SharedBias), SURVEY 8d counts them per sample"))}
        if step_bytes:
            roof["hbm"] = {"step_bytes": step_bytes, "achieved_GBps": step_bytes / (ms_per_step * 1e-3) / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                           "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                           "how": "sum over every kernel of the step of rocprofv3 PMC FETCH_SIZE + WRITE_SIZE bytes (committed profile, "
                                  "same command) / this run's step time"}
        if prof and prof["time_ms"] > 0:
            raw_ms = prof["time_ms"]
            # `achieved` / `frac`: the RAW in-situ brackets (what the contract asks for: algorithmic flops / HIP-event time of the launches,
            # measured live).  A bracket also holds dispatch + event cost (~5 us per launch: the same two events around a kernel that does
            # nothing measure it), so the raw figure is a LOWER bound; the committed rocprofv3 trace of this command (roofline.rocprof:
            # kernel durations only) sits a few % above it, and the bracket-calibrated figure (side fields) above that.  VERDICT r4 item 9.
            cal_ms = max(raw_ms - prof["launches"] * max(prof["null_bracket_us"] - 1.0, 0.0) * 1e-3, 0.5 * raw_ms)
            ach = prof["flops"] / (raw_ms * 1e-3) / 1e12
            ach_cal = prof["flops"] / (cal_ms * 1e-3) / 1e12
            roof.update({"achieved": ach, "frac": ach / PEAK_BF16_TFLOPS, "launches_per_step": prof["launches"] // args.profile_gemm,
                         "gemm_ms_per_step": raw_ms / args.profile_gemm, "event_bracket_overhead_us": prof["null_bracket_us"],
                         "achieved_events_calibrated": ach_cal, "frac_events_calibrated": ach_cal / PEAK_BF16_TFLOPS,
                         "gemm_ms_per_step_events_calibrated": cal_ms / args.profile_gemm,
                         "avg_launch_us": raw_ms * 1e3 / max(prof["launches"], 1),
                         "gemm_flops_per_step": prof["flops"] / args.profile_gemm,
                         "algorithmic_bytes_per_launch": prof["bytes"] / max(prof["launches"], 1),
                         "how": "instrumented eager step right after the timed region, the stream parked behind a spin kernel "
                                "until the host has enqueued the whole step: every MFMA-GEMM launch of the step is bracketed IN SITU "
                                "by two HIP events on its launch stream (one launch each, no relaunch, caches as the step leaves "
                                "them; includes the split-K reduce where ofa_gemm runs one; a layer's grouped weight-gradient "
                                "launch is one launch, its slab fold is a FoldQueue kernel outside this family time -- see "
                                "roofline.rocprof for the profiler's figure incl. reduces and folds).  `achieved` / `frac` are the raw "
                                "bracketed times; *_events_calibrated subtract the measured cost of an EMPTY bracket per launch "
                                "(event_bracket_overhead_us - 1 us: the same two events around a one-element kernel)"})
            fam_ms, all_ms, src = rocprof_gemm_ms(args.workload)
            if src and not fam_ms:
                roof["rocprof_refused"] = src
            if fam_ms:
                f = prof["flops"] / args.profile_gemm / (fam_ms * 1e-3) / 1e12
                roof["rocprof"] = {"gemm_family_ms_per_step": fam_ms, "all_kernels_ms_per_step": all_ms, "achieved": f,
                                   "frac": f / PEAK_BF16_TFLOPS, "source": src,
                                   "how": "rocprofv3 --kernel-trace --stats of this command (committed): durations of the replayed "
                                          "graph's gemm_* + splitk_reduce + fold_batched kernels; flops = this run's GEMM flops"}
        else:
            roof.update({"achieved": step_tflops, "frac": step_tflops / PEAK_BF16_TFLOPS})
        out = {
            "metric": "multimodal tokens/sec (enc+dec train step)", "value": total_tokens * args.steps / dt,
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "tokens_per_sec_per_gpu": total_tokens * args.steps / dt / world,
            "config": {"workload": desc + f", OFA-{args.arch} enc-dec train step (fwd+CE+bwd+allreduce+clip+Adam)",
                       "micro_batches_per_step": len(batches[0][0]),
                       "arch": args.arch, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "padded_positions_per_sample": nvis + Ts_text + Tt, "nonpad_tokens_per_step": total_tokens,
                       "vocab": len(d), "parallelism": f"dp{world}", "random_init": True, "step_mode": graph_mode, "ragged_row_packing": packed,
                       "distinct_batches": len(batches)},
            "roofline": roof,
            "source_id": source_id(),      # the product sources of this run (tools/build_id.py); the quoted profile summaries carry the same id
        }
        if dp is not None:
            out["dp"] = dp
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, model, d)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
