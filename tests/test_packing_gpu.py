"""GPU: ragged row packing (ofasys_amd/packing.py) against the padded path it replaces.  No non-pad output of the reference
depends on a padded position (padding is zeroed, masked as a key and ignored by the criterion: model/transformer.py:110-112,
multihead_attention.py:319-326, cross_entropy.py:27-41), so the packed stack must reproduce the padded stack's logits at every
non-pad position and its gradients; the padded stack itself is pinned to the reference by tests/test_model_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle.cases import VOCAB_EXTRA, make_target
from tests.model_util import build_model, make_slots

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")]
DEV = "cuda"
V = 4 + VOCAB_EXTRA
CASE = {"arch": "tiny", "active": {"text"}, "overrides": {"use_self_attn_bias": False, "entangle_position_embedding": True, "dropout": 0.0},
        "adaptor_overrides": {"text": {"entangle_position_embedding": True}}}


# the reference's DEFAULT configuration (use_self_attn_bias: abs-pos + rel-pos bias in every attention): packed since round 3 -- the bias
# is computed inside the attention kernels from packed pos_q / pos_k rows and position-indexed bucket ids (ops.PosBias)
BIASED = {"arch": "tiny", "active": {"text"}, "overrides": {"dropout": 0.0}, "adaptor_overrides": {}}
BIASED_IMG = {"arch": "tiny", "active": {"text", "image_resnet"}, "overrides": {"dropout": 0.0},
              "adaptor_overrides": {"image_resnet": {"resnet_type": "resnet50"}}}


def _tok(key, shape, lengths=None, bos=False):
    return recipe.tokens("pack." + key, shape, V, lengths, bos=0 if bos else None)


def _batch(two_slots):
    B = 5
    src_a = _tok("srcA", (B, 37), [37, 12, 30, 1, 22])
    prev = _tok("prev", (B, 19), [19, 3, 11, 19, 7], bos=True)
    vals = [("TEXT", True, src_a, None)]
    if two_slots:                     # a second ragged source slot: the valid positions of a row are no longer a prefix
        vals.append(("STRUCT", True, _tok("srcB", (B, 9), [4, 9, 1, 9, 6]), None))
    vals.append(("TEXT", False, prev, None))
    enc_tokens = torch.cat([v for m, s, v, a in vals if s], 1)
    return vals, make_target(prev), enc_tokens.eq(1), prev.eq(1)


@pytest.mark.parametrize("two_slots,case_name", [(False, "bias-free"), (True, "bias-free"), (False, "biased"), ("image", "biased")])
def test_packed_forward_backward_equals_padded(two_slots, case_name):
    from ofasys_amd import ops
    from ofasys_amd.packing import build_pack_plan
    CASE = globals()["CASE"] if case_name == "bias-free" else (BIASED_IMG if two_slots == "image" else BIASED)
    if two_slots == "image":          # [IMAGE (16 positions, all valid)][TEXT ragged] -> [TEXT]: the valid positions stay a prefix
        vals, target, enc_mask, dec_mask = _batch(False)
        img = recipe.floats("pack.image", (enc_mask.shape[0], 3, 64, 64))
        vals = [("IMAGE", True, img, None)] + vals
        enc_mask = torch.cat([torch.zeros(enc_mask.shape[0], 16, dtype=torch.bool), enc_mask], 1)
    else:
        vals, target, enc_mask, dec_mask = _batch(two_slots)
    plan = build_pack_plan(enc_mask, dec_mask, bucket=64)
    assert plan.enc_prefix == (two_slots is not True) and plan.dec_prefix
    assert plan.enc_tokens == int((~enc_mask).sum()) and plan.dec_tokens == int((~dec_mask).sum())
    assert plan.enc_index.numel() % 64 == 0 and plan.enc_index.numel() < enc_mask.numel()       # really fewer rows
    assert all(int(o) % 8 == 0 for o in plan.enc_self.table[:, 0])
    res = {}
    for mode in ("padded", "packed"):
        model, d = build_model(CASE, DEV, torch.bfloat16)
        model.train()                                          # dropout 0: train mode only matters for the fused kernels' choice
        slots = make_slots(vals, DEV, torch.bfloat16)
        if mode == "padded":
            logits = model(slots)[0]
            tgt = target.to(DEV)
        else:
            p = plan.to(DEV)
            logits = model(slots, pack=p)[0]
            assert logits.shape == (1, plan.dec_index.numel(), len(d))
            idx = p.dec_index
            tgt = target.to(DEV).reshape(-1)[idx.clamp_min(0)].masked_fill(idx < 0, d.pad()).view(1, -1)
        loss = ops.cross_entropy_sum(logits, tgt, d.pad())
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (logits.detach().float().cpu(), float(loss), {k: p_.grad.detach().float().cpu() for k, p_ in model.named_parameters()
                                                                   if p_.grad is not None})
    lp, losp, gp = res["padded"]
    lk, losk, gk = res["packed"]
    # logits at every non-pad decoder position
    di = plan.dec_index
    valid = di >= 0
    want = lp.reshape(-1, lp.shape[-1])[di[valid]]
    got = lk[0][valid]
    scale = float(want.abs().max())
    # (same arithmetic per row; with several ragged slots the keys of a sample sit in different 32-key blocks than in the padded
    #  layout, so the online-softmax partial sums round differently: a couple of bf16 ulps)
    assert float((got - want).abs().max()) <= 2e-2 * scale
    assert float(lk[0][~valid].abs().max()) < 1e30                        # filler rows: finite (they are inert)
    assert abs(losp - losk) <= 2e-3 * abs(losp)
    gmax = max(float(g.norm()) for g in gp.values())
    assert set(gp) == set(gk)
    for k in gp:
        a, b = gp[k], gk[k]
        t = 1e-1 if ".embed_images." in k else 3e-2                       # (bf16 through the BatchNorm trunk: see tests/test_model_gpu.py)
        assert float((a - b).norm()) <= t * float(a.norm()) + 2e-3 * gmax, k


def test_packed_train_step_and_graph_replay():
    """TrainStep on ragged batches: the pack plan is part of the step's static inputs (new batches of the same bucketed row count
    replay the same hipGraph), and a run of steps matches the padded run (bf16, dropout 0) within rounding."""
    from ofasys_amd.packing import build_pack_plan
    from ofasys_amd.trainer import TrainStep
    runs = {}
    for mode in ("padded", "packed-eager", "packed-graph"):
        model, d = build_model(CASE, DEV, torch.bfloat16)
        tr = TrainStep(model, lr=1e-3, clip_norm=1.0, use_graph=mode == "packed-graph", graph_warmup=1)
        losses = []
        for step in range(6):
            g = np.random.Generator(np.random.Philox(key=100 + step))
            B = 6
            sl = g.integers(5, 40, B)
            tl = g.integers(2, 20, B)
            sl[0], tl[0] = 39, 19                                           # same padded shape every step
            src = recipe.tokens(f"pack.s{step}", (B, 39), V, sl.tolist())
            prev = recipe.tokens(f"pack.p{step}", (B, 19), V, tl.tolist(), bos=0)
            sample = {"slots": make_slots([("TEXT", True, src, None), ("TEXT", False, prev, None)], DEV, torch.bfloat16),
                      "target": make_target(prev).to(DEV)}
            if mode != "padded":
                sample["pack"] = build_pack_plan(src.eq(1), prev.eq(1), bucket=256).to(DEV)
            losses.append(float(tr.train_step([sample])["stats"][1]))
        torch.cuda.synchronize()
        runs[mode] = (losses, tr)
    lp, lke, lkg = runs["padded"][0], runs["packed-eager"][0], runs["packed-graph"][0]
    assert lke == lkg                                                        # replays == eager, bit for bit
    tr = runs["packed-graph"][1]
    assert sum(1 for e in tr._graphs.values() if "graphs" in e) == 1        # one bucketed structure -> one graph for all batches
    for a, b in zip(lp, lke):
        assert abs(a - b) <= 2e-2 * abs(a)


def test_packing_refuses_what_it_cannot_run():
    from ofasys_amd.packing import build_pack_plan
    vals, target, enc_mask, dec_mask = _batch(False)
    plan = build_pack_plan(enc_mask, dec_mask, bucket=64).to(DEV)
    # a rel-pos bias over packed rows needs "packed row r == padded position r": two ragged source slots break that
    vals2, _, enc_mask2, dec_mask2 = _batch(True)
    plan2 = build_pack_plan(enc_mask2, dec_mask2, bucket=64).to(DEV)
    assert not plan2.enc_prefix
    model, d = build_model(BIASED, DEV, torch.bfloat16)
    with pytest.raises(NotImplementedError, match="prefix"):
        model(make_slots(vals2, DEV, torch.bfloat16), pack=plan2)
    model, d = build_model(CASE, DEV, torch.float32)                         # fp32: no fused ragged attention kernel
    with pytest.raises(NotImplementedError, match="fused"):
        model(make_slots(vals, DEV), pack=plan)


def test_ragged_attention_backward_reads_nothing_behind_lse_and_delta():
    """Regression of round 5's replayed-graph fault (profiles/round6_graph_fault_root_cause.txt): the dK/dV kernel fetched the softmax
    statistics of a whole 32-row query block, unclamped -- for the last head of a last sample that ends in the bucket's final rows,
    up to 31 floats past the END of the [heads, rows] lse / delta buffers.  Here lse closes a freshly mapped allocator segment (the
    next address is normally unmapped: an overrun kills the process with a GPU memory access fault) and the result must equal the
    run with a roomy lse bit for bit."""
    from ofasys_amd import kernels as K
    from ofasys_amd.packing import Segments
    heads, D, R = 4, 256, 64
    lens = [24, 40]                                   # sample 1 = rows 24 .. 63: its second 32-row block spans rows 56 .. 87 > R
    table = torch.tensor([[0, 24, 0, 24], [24, 40, 24, 40]], dtype=torch.int32, device=DEV)
    seg = Segments(table, 2, R, R, 64, 64)
    g = torch.Generator().manual_seed(5)
    q, k, v, do = (torch.randn(1, R, D, generator=g).to(torch.bfloat16).to(DEV) for _ in range(4))
    out, lse = K.attn_fwd(q, k, v, heads, 0.125, seg=seg)
    ref = K.attn_bwd(q, k, v, out, do, lse, heads, 0.125, seg=seg)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                          # no cached block can serve the next request: it gets a segment of its own
    big = torch.empty((12 << 20) + (2 << 20), dtype=torch.uint8, device=DEV)
    n = lse.numel() * 4
    tail = big[big.numel() - n:].view(torch.float32).view_as(lse)
    tail.copy_(lse)
    got = K.attn_bwd(q, k, v, out, do, tail, heads, 0.125, seg=seg)
    torch.cuda.synchronize()
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a, b)
