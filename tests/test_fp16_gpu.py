"""fp16 as a compute dtype (the reference trainer's default precision, config/default_trainer.yaml:7-25): the kernels the train
step uses against plain fp32 PyTorch references of the same ops on the SAME fp16 inputs, the model against the reference-run
golden outputs, and the train step under the dynamic loss scaler.  Tolerances: fp16 rounds at 2^-11 (bf16: 2^-8)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import _attn_ref, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = torch.float16


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ofasys_amd import kernels
    return kernels


@pytest.mark.parametrize("M,N,K_", [(64, 64, 64), (200, 136, 72), (534, 264, 768), (77, 64, 1032), (256, 3072, 768), (1000, 208, 264),
                                    (4032, 3072, 256), (3320, 3848, 1216)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_fp16(K, M, N, K_, ta, tb):
    """MFMA GEMM on fp16 operands (v_mfma_f32_32x32x16_f16): every layout, LDS-DMA and register-staged loops, the 128x128 /
    ring / eight-wave tiles the planner picks, column bias, alpha, fp32 output, accumulation, split-K."""
    torch.manual_seed(1)
    a = torch.randn((K_, M) if ta else (M, K_), device=DEV).to(H)
    b = torch.randn((N, K_) if tb else (K_, N), device=DEV).to(H)
    bias = torch.randn(N, device=DEV).to(H)
    prod = (a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())
    ref = (prod + bias.float()) * 0.5
    out = K.gemm(a, b, ta, tb, bias=bias, alpha=0.5)
    assert out.dtype == H and rel(out, ref) < 2e-3
    assert rel(K.gemm(a, b, ta, tb, bias=bias, alpha=0.5, out_f32=True), ref) < 1e-3
    assert rel(K.gemm(a, b, ta, tb, bias=bias, alpha=0.5, force_simple=True), ref) < 2e-3
    acc = torch.ones(M, N, device=DEV, dtype=H)
    K.gemm(a, b, ta, tb, out=acc, accumulate=True)
    assert rel(acc.float() - 1, prod) < 4e-3
    if ta:                                                      # the weight-gradient form: split-K + immediate reduce
        big = torch.randn(4096, M if M % 8 == 0 else 264, device=DEV).to(H)
        xx = torch.randn(4096, 256, device=DEV).to(H)
        assert rel(K.gemm(big, xx, True, False), big.float().t() @ xx.float()) < 2e-3


@pytest.mark.parametrize("rows,cols", [(130, 768), (67, 3072), (5000, 768)])
@pytest.mark.parametrize("gelu", [False, True])
def test_layernorm_fp16(K, rows, cols, gelu):
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device=DEV).to(H)
    g = (1 + 0.1 * torch.randn(cols, device=DEV)).to(H)
    b = (0.1 * torch.randn(cols, device=DEV)).to(H)
    dy = torch.randn(rows, cols, device=DEV).to(H)
    xr, gr, br = (t.float().requires_grad_(True) for t in (x, g, b))
    yr = F.layer_norm(F.gelu(xr) if gelu else xr, (cols,), gr, br, 1e-5)
    yr.backward(dy.float())
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5, fuse_gelu=gelu)
    dx, dg, db, dbias = K.layernorm_bwd(dy, x, g, mean, rstd, fuse_gelu=gelu, want_dbias=gelu)
    big = 4 if rows > 1000 else 1
    assert y.dtype == H and rel(y, yr) < 2e-3
    assert rel(dx, xr.grad) < 4e-3
    assert rel(dg, gr.grad) < 4e-3 * big and rel(db, br.grad) < 4e-3 * big
    if gelu:
        assert rel(dbias, xr.grad.sum(0)) < 4e-3 * big


def test_residual_join_fp16(K):
    """y = res + LN_a(x), z = LN_b(y) (dropout 0) against fp32 torch on the same fp16 inputs."""
    torch.manual_seed(2)
    rows, cols = 300, 768
    x, res = (torch.randn(rows, cols, device=DEV).to(H) for _ in range(2))
    ga, ba, gb, bb = ((1 + 0.1 * torch.randn(cols, device=DEV)).to(H), (0.1 * torch.randn(cols, device=DEV)).to(H),
                      (1 + 0.1 * torch.randn(cols, device=DEV)).to(H), (0.1 * torch.randn(cols, device=DEV)).to(H))
    yr = res.float() + F.layer_norm(x.float(), (cols,), ga.float(), ba.float(), 1e-5).to(H).float()
    zr = F.layer_norm(yr.to(H).float(), (cols,), gb.float(), bb.float(), 1e-5)
    y, z, stats, _ = K.join_fwd(x, res, (ga, ba), (gb, bb), 1e-5, 0.0, 0, 0, None)
    assert y.dtype == H and rel(y, yr) < 2e-3 and rel(z, zr) < 3e-3


@pytest.mark.parametrize("B,heads,T,S,causal,use_bias,use_kpm", [
    (2, 4, 45, 45, True, True, True),
    (1, 12, 130, 130, False, True, True),
    (2, 3, 64, 267, False, False, True),
    (2, 2, 300, 131, True, True, True),
    (3, 2, 448, 448, False, False, False),
])
def test_fused_attention_fp16(K, B, heads, T, S, causal, use_bias, use_kpm):
    torch.manual_seed(5)
    D = heads * 64
    q, k, v = (torch.randn(B, n, D, device=DEV).to(H) for n in (T, S, S))
    bias = torch.randn(B * heads, T, S, device=DEV).to(H) if use_bias else None
    kpm = None
    if use_kpm:
        kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        kpm[-1, S - 5:] = True
    c = (1 + 0.2 * torch.randn(heads, device=DEV)).to(H)
    scale = (64 * 2) ** -0.5
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    br = bias.float().requires_grad_(True) if use_bias else None
    cr = c.float().requires_grad_(True)
    ref = _attn_ref(qr, kr, vr, heads, scale, br, kpm, cr, causal)
    dout = torch.randn(B, T, D, device=DEV).to(H)
    ref.backward(dout.float())
    out, lse = K.attn_fwd(q, k, v, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal)
    assert out.dtype == H and rel(out, ref) < 3e-3
    dq, dk, dv, dbias, delta = K.attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal,
                                           need_dbias=use_bias)
    assert rel(dq, qr.grad) < 5e-3 and rel(dk, kr.grad) < 5e-3 and rel(dv, vr.grad) < 5e-3
    if use_bias:
        assert rel(dbias, br.grad) < 5e-3


def test_embedding_criterion_adam_fp16(K):
    from ofasys_amd import ops
    from tests.golden_util import load_golden
    torch.manual_seed(3)
    V, D, n = 1000, 768, 333
    w = torch.randn(V, D, device=DEV).to(H)
    ids = torch.randint(0, V, (n,), device=DEV)
    assert torch.equal(K.embedding_fwd(w, ids), w[ids])
    # label-smoothed cross entropy on fp16 logits against the reference-run golden vectors (tests/golden/ls_cross_entropy.npz)
    g = load_golden("ls_cross_entropy")
    for name in ("plain", "range_mask"):
        eps, cs, ce, dw = [float(v) for v in g[name + ".cfg"]]
        x = torch.from_numpy(g["logits"]).to(DEV).to(H).requires_grad_(True)
        tg = torch.from_numpy(g[name + ".target"]).to(DEV)
        sm = torch.from_numpy(g[name + ".sample_mask"]).bool().to(DEV) if (name + ".sample_mask") in g else None
        loss, nll, ntok = ops.label_smoothed_cross_entropy(x, tg, 1, eps, None if cs < 0 else (int(cs), int(ce)), sm, dw)
        loss.backward()
        assert abs(float(loss) - float(g[name + ".loss"][0])) <= 3e-3 * abs(float(g[name + ".loss"][0]))
        assert x.grad.dtype == H and rel(x.grad, torch.from_numpy(g[name + ".dlogits"]).to(DEV)) < 5e-3
    # Adam on the fp32 master, model copy written in fp16
    nparam = 4096 * 3 + 4
    p0 = torch.randn(nparam, device=DEV)
    gr = torch.randn(nparam, device=DEV).to(H)
    master, m, vv, model = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), p0.to(H)
    K.adam_step(master, m, vv, gr, model, None, 1e-2, 0.9, 0.98, 1e-8, 0.01, 1)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    pr.grad = gr.float()
    opt.step()
    assert rel(master, pr.detach()) < 1e-5 and torch.equal(model, master.to(H))
    out = torch.zeros(1, device=DEV)
    K.sumsq(gr, out)
    assert abs(float(out) - float(gr.float().pow(2).sum())) / float(out) < 1e-5


def test_conv_batchnorm_fp16(K):
    from ofasys_amd import ops
    torch.manual_seed(4)
    B, Cin, Cout, Hh, Ww = 2, 64, 128, 14, 14
    x = torch.randn(B * Hh * Ww, Cin, device=DEV).to(H).requires_grad_(True)
    w = (0.05 * torch.randn(Cout, Cin, 3, 3, device=DEV)).to(H).requires_grad_(True)
    y, Ho, Wo = ops.conv2d(x, w, None, B, Hh, Ww, stride=1, pad=1)
    xr = x.detach().float().view(B, Hh, Ww, Cin).permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    assert rel(y, yr.permute(0, 2, 3, 1).reshape(-1, Cout)) < 3e-3
    bn = torch.nn.BatchNorm2d(Cout).to(DEV)
    bnh = torch.nn.BatchNorm2d(Cout).to(DEV)
    bnh.weight.data = bnh.weight.data.to(H)
    bnh.bias.data = bnh.bias.data.to(H)
    z = ops.batch_norm(y, bnh, relu=True)
    zr = F.relu(bn(yr))
    assert z.dtype == H and rel(z, zr.permute(0, 2, 3, 1).reshape(-1, Cout)) < 5e-3
    dz = torch.randn_like(z)
    z.backward(dz)
    zr.backward(dz.float().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2))
    assert rel(x.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin)) < 1e-2
    assert rel(w.grad, wr.grad) < 1e-2


def test_model_fp16_against_golden_and_train_step_under_the_loss_scaler():
    """The tiny multi-slot case in fp16: forward + loss against the reference-run golden values (fp32), then TrainStep with the
    reference's dynamic loss scaler (init scale 128): finite updates, a falling loss, a scale that grows after `scale_window`
    clean steps, and an overflow that halves it and skips the update."""
    from oracle.cases import CASES
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    from tests.golden_util import case_inputs, load_golden, rel_err
    from tests.model_util import build_model, make_slots
    name = "tiny_multislot"
    case, g = CASES[name], load_golden(name)
    model, d = build_model(case, DEV, H)
    model.eval()
    vals, target = case_inputs(case)
    slots = make_slots(vals, DEV, H)
    logits, extra, enc = model(slots, return_encoder_out=True)
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    assert logits.dtype == H
    assert rel_err(logits.detach().float().cpu(), g["logits"]) < 1e-2           # (bf16 on this case: ~2e-2)
    assert rel_err(loss.detach().float().cpu(), g["loss"][0]) < 5e-3
    model.train()
    step = TrainStep(model, lr=1e-3, clip_norm=1.0, loss_scale={"init_scale": 128.0, "scale_window": 4})
    sample = {"slots": slots, "target": target.to(DEV)}
    losses, scales = [], []
    for _ in range(10):
        last = step.train_step([sample])
        st = last["stats"].tolist()
        losses.append(st[1] / max(st[0], 1.0))
        scales.append(float(last["loss_scale"]))
    assert all(math.isfinite(x) for x in losses) and losses[-1] < losses[0]
    assert scales[0] >= 128.0 and max(scales) > 128.0                  # grew after scale_window clean updates
    cur, skipped0 = float(step._ls[0]), float(step._sched[4])
    step._ls[0] = 2.0 ** 40                                            # a scale far beyond fp16's range: the next step overflows
    last = step.train_step([sample])
    torch.cuda.synchronize()
    assert float(step._sched[4]) == skipped0 + 1 and float(step._ls[0]) < 2.0 ** 40
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_modal_ffn_matches_the_reference_fp16_run():
    """modal_ffn (one FFN expert per modality, transformer_layer.py:116-130): the model in fp16 against tests/golden/tiny_modal_ffn.npz,
    a HALF-precision run of the reference itself (its modal_for_ffn only runs in fp16) -- logits, loss, cross-attention map,
    every parameter's gradient norm (unused experts: zero) and five expert gradients in full; then the same step through
    TrainStep's hipGraph (the routing is built on the host from the slot layout: nothing in it syncs)."""
    from oracle.cases import CASES
    from ofasys_amd import ops
    from ofasys_amd.trainer import TrainStep
    from tests.golden_util import case_inputs, load_golden, rel_err
    from tests.model_util import build_model, make_slots
    name = "tiny_modal_ffn"
    case, g = CASES[name], load_golden(name)
    model, d = build_model(case, DEV, H)
    model.eval()
    vals, target = case_inputs(case)
    slots = make_slots(vals, DEV, H)
    logits, extra, enc = model(slots, return_encoder_out=True)
    loss = ops.cross_entropy_sum(logits, target.to(DEV), d.pad())
    model.zero_grad()
    loss.backward()
    assert rel_err(logits.detach().float().cpu(), g["logits"]) < 6e-3
    assert rel_err(loss.detach().float().cpu(), g["loss"][0]) < 3e-3
    assert rel_err(extra["attn"][0].float().cpu(), g["attn"]) < 6e-3
    params = dict(model.named_parameters())
    gn = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]))
    scale = max(gn.values())
    for k, want in gn.items():
        if k == "decoder.adaptor.embed_tokens.weight":
            continue
        p = params[k]
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        if want <= 0:                                  # unused: the shared fc1 / fc2 (no gradient), experts of absent modalities (zero)
            assert got == 0.0, k
        else:
            assert abs(got - want) <= 3e-2 * want + 1e-3 * scale, (k, got, want)
    for k in g:
        if k.startswith("grad."):
            got, want = params[k[5:]].grad.float().cpu().double(), torch.from_numpy(g[k]).double()
            assert float((got - want).abs().max()) <= 3e-2 * float(want.abs().max()) + 1e-4 * scale, k
    # state-dict schema: the experts are there under the reference's names
    keys = set(model.state_dict().keys())
    want_keys = {str(x).split("|")[0] for x in g["state_keys"]}
    assert {k for k in want_keys if "experts_fc" in k} <= keys
    # graph-captured train step.  (The autograd graph of the forward above must be gone first: a live graph keeps the parameters'
    # AccumulateGrad nodes bound to the stream it ran on, and torch would route the captured backward through that stream.)
    del logits, extra, enc, loss
    model.zero_grad(set_to_none=True)
    model.train()
    step = TrainStep(model, lr=1e-3, clip_norm=1.0, use_graph=True, graph_warmup=1, loss_scale={"init_scale": 128.0})
    sample = {"slots": slots, "target": target.to(DEV)}
    losses = []
    for _ in range(6):
        st = step.train_step([sample])["stats"].tolist()
        losses.append(st[1] / max(st[0], 1.0))
    assert any("graphs" in e for e in step._graphs.values()), "the modal_ffn step was not captured"
    assert all(math.isfinite(x) for x in losses) and losses[-1] < losses[0]
