"""CPU: the oracle restatement (oracle/restate.py) against golden vectors produced by RUNNING THE REFERENCE
(oracle/gen_golden.py).  This is what pins the oracle (the reference ships no tests)."""
import numpy as np
import pytest
import torch

from oracle import restate
from oracle.cases import CASES, ORACLE_CASES
from tests.golden_util import case_inputs, drop_keep_rows, load_golden, oracle_cfg, oracle_slots, rel_err, state_from_golden

TOL = 2e-5   # fp32 vs fp32, different op order only (reference noise floor 2e-6)


@pytest.mark.parametrize("name", list(CASES) + list(ORACLE_CASES))
def test_forward_backward_matches_reference(name):
    torch.set_num_threads(8)
    case = {**CASES, **ORACLE_CASES}[name]
    # a "half" case is a reference run in fp16 (modal_ffn only exists there): the fp32 oracle meets it at fp16's rounding
    TOL, GTOL = (4e-3, 2e-2) if case.get("half") else (globals()["TOL"], 1e-4)
    g = load_golden(name)
    state = state_from_golden(g)
    params = {k: v.requires_grad_(True) for k, v in state.items()
              if v.is_floating_point() and not k.endswith(("version", "running_mean", "running_var"))}
    # tied embedding: one tensor under both names, as in the reference (adaptor/general.py:193-221)
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    cfg = oracle_cfg(case)
    cfg.drop_keep = drop_keep_rows(g)          # (stochastic-depth cases: the reference's recorded per-sample draws)
    vals, target = case_inputs(case)
    assert np.array_equal(target.numpy(), g["target"])
    rec = {}
    logits, extra = restate.model_forward(state, cfg, oracle_slots(vals), rec)
    loss, n = restate.cross_entropy(logits, target)
    assert n == int(g["sample_size"][0]) and not cfg.drop_keep
    assert rel_err(logits.detach(), g["logits"]) < TOL
    assert rel_err(loss.detach(), g["loss"][0]) < TOL
    assert rel_err(extra["attn"].detach(), g["attn"]) < TOL
    big = case["arch"] in ("base", "large")
    enc = rec["encoder_out"].detach()
    assert rel_err(enc.reshape(-1)[::97] if big else enc, g["encoder_out"]) < TOL
    for k in g:
        if k.startswith("rec."):
            mine = rec[k[4:]].detach()
            if big and mine.numel() > 200000:
                mine = mine.reshape(-1)[::97]
            assert rel_err(mine, g[k]) < TOL, k
    loss.backward()
    gn = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]))
    for k, want in gn.items():
        p = state[k]
        if k == "decoder.adaptor.embed_tokens.weight":
            continue  # same tensor as the encoder's
        if want < 0:
            assert p.grad is None or float(p.grad.norm()) == 0.0, k
            continue
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        assert abs(got - want) <= GTOL * want + (1e-2 if case.get("half") else 1e-5), (k, got, want)  # k_proj.bias grads are exactly 0 in maths
    for k in g:
        if k.startswith("grad."):
            assert rel_err(state[k[5:]].grad, g[k]) < 5 * TOL, k
        if k.startswith("buffer."):           # BatchNorm running statistics after one training-mode forward
            assert rel_err(state[k[7:]].detach().double(), g[k].astype(np.float64)) < TOL, k


@pytest.mark.parametrize("shape", __import__("oracle.edge_cases", fromlist=["SHAPES"]).SHAPES, ids=lambda s: f"b{s[0]}s{s[1]}t{s[2]}")
def test_edge_shapes_match_reference(shape):
    """The oracle on the degenerate / ragged shapes the HIP path is compared with it on (tests/test_model_gpu.py::
    test_ragged_and_degenerate_shapes_match_oracle): one token, batch of one at odd lengths, single-token rows, an almost entirely
    padded row -- against the reference run on the same inputs (tests/golden/edge_shapes.npz, oracle/gen_edge_golden.py)."""
    from oracle import edge_cases as EC
    torch.set_num_threads(8)
    G = load_golden("edge_shapes")
    case = CASES["tiny_text"]
    state = state_from_golden(load_golden("tiny_text"))
    params = {k: v.requires_grad_(True) for k, v in state.items() if v.is_floating_point() and not k.endswith("version")}
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    src, prev, target = EC.inputs(shape)
    k = EC.key(shape)
    assert np.array_equal(target.numpy(), G[k + ".target"])
    logits, _ = restate.model_forward(state, oracle_cfg(case), [restate.OSlot("TEXT", True, src, None), restate.OSlot("TEXT", False, prev, None)])
    loss, _ = restate.cross_entropy(logits, target)
    assert rel_err(logits.detach(), G[k + ".logits"]) < TOL
    assert rel_err(loss.detach(), G[k + ".loss"][0]) < TOL
    loss.backward()
    want = dict(zip([str(n) for n in G["grad_norm_keys"]], G[k + ".grad_norms"]))
    scale = max(want.values())
    checked = 0
    for n, w in want.items():
        g = params[n].grad
        got = float(g.double().norm()) if g is not None else 0.0
        if w < 0:
            assert got == 0.0, n
        else:
            assert abs(got - w) <= 1e-4 * w + 1e-6 * scale, (n, got, w)
            checked += 1
    assert checked > 100


def test_maximum_positions_match_reference():
    """1024 source / 1024 target tokens (the edge of the position and bucket tables), forward: the oracle against the reference's run
    (every 53rd logit and attention value of tests/golden/edge_shapes.npz) -- the GPU test compares the HIP path with the oracle there."""
    from oracle import edge_cases as EC
    torch.set_num_threads(8)
    G = load_golden("edge_shapes")
    state = state_from_golden(load_golden("tiny_text"))
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    src, prev = EC.max_position_inputs()
    with torch.no_grad():
        logits, extra = restate.model_forward(state, oracle_cfg(CASES["tiny_text"]),
                                              [restate.OSlot("TEXT", True, src, None), restate.OSlot("TEXT", False, prev, None)])
    assert abs(float(logits.abs().max()) - float(G["maxpos.logits_absmax"][0])) <= TOL * float(G["maxpos.logits_absmax"][0])
    scale = float(G["maxpos.logits_absmax"][0])
    assert float((logits.reshape(-1)[::EC.MAX_POS_STRIDE] - torch.from_numpy(G["maxpos.logits"])).abs().max()) < TOL * scale
    assert rel_err(extra["attn"].reshape(-1)[::EC.MAX_POS_STRIDE], G["maxpos.attn"]) < TOL


def test_token_bucket_bit_exact():
    g = load_golden("tiny_text")
    b = restate.make_token_bucket_position(256, 1024)
    import zlib
    assert zlib.crc32(b.contiguous().numpy().tobytes()) == int(g["token_rp_bucket_crc"][0])
    assert np.array_equal(b[:300:7, :300:7].numpy(), g["token_rp_bucket_corner"])
    assert int(b.min()) == 0 and int(b.max()) == 510


def test_box_bins_bit_exact():
    g = load_golden("box_bins")
    for row, want in zip(g["coords"], g["bins"]):
        assert restate.box_to_bins(row, int(g["max_image_size"][0]), int(g["num_bins"][0])) == list(want)


def test_fused_softmax_semantics():
    g = load_golden("fused_softmax")
    x = torch.from_numpy(g["x"])
    s = float(g["scale"][0])
    assert rel_err(restate.scaled_softmax(x, s), g["y"]) < 1e-6
    assert rel_err(restate.scaled_masked_softmax(x, torch.from_numpy(g["mask"]), s), g["y_masked"]) < 1e-6
    # backward formula against autograd
    xx = x.clone().requires_grad_(True)
    y = torch.softmax(xx * s, -1)
    dy = torch.randn_like(y)
    y.backward(dy)
    assert rel_err(restate.scaled_softmax_bwd(dy, y.detach(), s), xx.grad) < 1e-5
    # causal variant: rows sum to one over the visible prefix, zeros above the diagonal
    c = restate.scaled_upper_triang_masked_softmax(x[0, :, :8, :8].contiguous(), s)
    assert torch.all(torch.triu(c, 1) == 0) and torch.allclose(c.sum(-1), torch.ones(3, 8), atol=1e-6)
    assert restate.get_batch_per_block(128, 128, 2, 4) == 8 and restate.get_batch_per_block(64, 448, 2, 4) == 4


@pytest.mark.parametrize("name", ["plain", "range", "range_mask", "drop"])
def test_label_smoothed_cross_entropy_matches_reference(name):
    g = load_golden("ls_cross_entropy")
    eps, cs, ce, dw = [float(v) for v in g[name + ".cfg"]]
    x = torch.from_numpy(g["logits"]).clone().requires_grad_(True)
    tg = torch.from_numpy(g[name + ".target"])
    crange = None if cs < 0 else (int(cs), int(ce))
    sm = torch.from_numpy(g[name + ".sample_mask"]).bool() if (name + ".sample_mask") in g else None
    loss, nll, ntok = restate.label_smoothed_cross_entropy(x, tg, eps, 1, crange, sm, dw)
    loss.backward()
    assert ntok == int(g[name + ".ntokens"][0])
    assert rel_err(loss.detach(), g[name + ".loss"][0]) < 1e-6 and rel_err(nll.detach(), g[name + ".nll"][0]) < 1e-6
    assert rel_err(x.grad, g[name + ".dlogits"]) < 1e-5


def _incremental_oracle(state, cfg, vals, V):
    """Drive oracle/restate.py's incremental decoder through the scenario of oracle/incremental_case.py."""
    from oracle.incremental_case import BEAM_ORDER, NEW_ORDER, REORDER_AT, STEPS, beam_prefix
    from oracle.restate import OSlot
    enc = restate.encoder_forward(state, cfg, [s for s in oracle_slots(vals) if s.is_src])
    enc = restate.reorder_encoder_out(enc, torch.tensor(BEAM_ORDER))
    prev = beam_prefix(V)
    inc, logits, extra = {}, [], None
    for t in range(STEPS):
        out, extra = restate.decoder_step(state, cfg, [OSlot("TEXT", False, prev[:, :t + 1], None)], enc, inc)
        logits.append(out[:, -1])
        if t == REORDER_AT:
            order = torch.tensor(NEW_ORDER)
            restate.reorder_incremental_state(inc, order)
            enc = restate.reorder_encoder_out(enc, order)
            prev = prev.index_select(0, order)
    full, _ = restate.decoder_forward(state, cfg, [OSlot("TEXT", False, prev, None)], enc)
    return torch.stack(logits), extra["attn"], full[:, -1], inc


def test_incremental_decoding_matches_reference():
    """KV-cache decoding + beam reorder against the reference run step by step."""
    from oracle.cases import VOCAB_EXTRA
    torch.set_num_threads(8)
    case = CASES["tiny_text"]
    g = load_golden("tiny_text")
    gi = load_golden("tiny_text_incremental")
    state = state_from_golden(g)
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    vals, _ = case_inputs(case)
    with torch.no_grad():
        logits, attn, full_last, inc = _incremental_oracle(state, oracle_cfg(case), vals, 4 + VOCAB_EXTRA)
    assert rel_err(logits, gi["logits"]) < TOL
    assert rel_err(attn, gi["attn"]) < TOL
    assert rel_err(full_last, gi["full_last"]) < TOL
    assert rel_err(logits[-1], full_last) < TOL                      # incremental == teacher-forced on the final beams
    assert rel_err(inc[(0, "self")]["prev_key"], gi["prev_key_l0"]) < TOL
    assert rel_err(inc[(0, "self")]["prev_value"], gi["prev_value_l0"]) < TOL


def test_incremental_decoding_with_finished_beams_matches_reference():
    """Pad tokens inside the prefix (finished hypotheses): the cached key-padding mask takes part in every later step."""
    from oracle.cases import VOCAB_EXTRA
    from oracle.incremental_case import BEAM_ORDER, STEPS, padded_prefix
    from oracle.restate import OSlot
    torch.set_num_threads(8)
    case = CASES["tiny_text"]
    gi = load_golden("tiny_text_incremental")
    state = state_from_golden(load_golden("tiny_text"))
    state["decoder.adaptor.embed_tokens.weight"] = state["encoder.adaptor.embed_tokens.weight"]
    vals, _ = case_inputs(case)
    cfg = oracle_cfg(case)
    with torch.no_grad():
        enc = restate.encoder_forward(state, cfg, [s for s in oracle_slots(vals) if s.is_src])
        enc = restate.reorder_encoder_out(enc, torch.tensor(BEAM_ORDER))
        prev, inc, logits = padded_prefix(4 + VOCAB_EXTRA), {}, []
        for t in range(STEPS):
            out, _ = restate.decoder_step(state, cfg, [OSlot("TEXT", False, prev[:, :t + 1], None)], enc, inc)
            logits.append(out[:, -1])
    assert rel_err(torch.stack(logits), gi["logits_padded"]) < TOL
    assert np.array_equal(inc[(1, "self")]["prev_key_padding_mask"].float().numpy(), gi["kpm_padded_l1"])


def test_fused_softmax_backward_restatement_matches_reference():
    """restate.scaled_*softmax* + scaled_softmax_bwd against autograd through the reference's forward_torch_softmax, fp32 and
    fp16 (tests/golden/fused_softmax_bwd.npz).  In fp16 the extension's backward reads the ROUNDED fp16 probabilities
    (scaled_masked_softmax.h:329-423) while the torch fallback differentiates through fp32 probabilities: <= 2e-3 apart."""
    g = load_golden("fused_softmax_bwd")
    scale = float(g["scale"][0])
    x32, dy32 = torch.from_numpy(g["x"]), torch.from_numpy(g["dy"])
    mask = torch.from_numpy(g["mask"])
    b, np_, sq, _ = x32.shape
    for tag, dt, tol in (("f32", torch.float32, 2e-6), ("f16", torch.float16, 2e-3)):
        x, dy = x32.to(dt), dy32.to(dt)
        ys = {"plain": restate.scaled_softmax(x, scale), "masked": restate.scaled_masked_softmax(x, mask, scale),
              "causal": restate.scaled_upper_triang_masked_softmax(x.view(-1, sq, sq), scale).view(b, np_, sq, sq)}
        for name, y in ys.items():
            assert y.dtype == dt
            assert rel_err(y.float(), g[f"{tag}.{name}.y"].astype(np.float32)) < (1e-6 if dt == torch.float32 else 1e-3), (tag, name)
            dx = restate.scaled_softmax_bwd(dy, y, scale).to(dt)
            assert rel_err(dx.float(), g[f"{tag}.{name}.dx"].astype(np.float32)) < tol, (tag, name)
            if name == "causal":
                assert float(torch.triu(dx.float(), 1).abs().max()) == 0.0
