"""CPU: host-side logic of the mirrored interface -- state-dict schema, slot routing, integer paths, flat arena,
and the data-parallel bucket reducer under a 2-rank gloo group."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from oracle.cases import CASES
from tests.golden_util import load_golden
from tests.model_util import build_model


@pytest.mark.parametrize("name", ["tiny_text", "base_patch", "tiny_resnet", "tiny_video", "tiny_audio", "tiny_resnet_droppath",
                                  "tiny_audio_chmask", "tiny_text_noscale", "tiny_text_postln", "tiny_multislot_shared"])
def test_state_dict_schema_matches_reference(name):
    g = load_golden(name)
    model, _ = build_model(CASES[name])
    mine = [f"{k}|{tuple(v.shape)}|{str(v.dtype)}" for k, v in model.state_dict().items()]
    assert mine == [str(x) for x in g["state_keys"]]          # same keys, shapes, dtypes AND order as the reference
    assert model.encoder.adaptor.embed_tokens.weight is model.decoder.adaptor.embed_tokens.weight


def test_resnet_drop_path_rates_follow_the_reference_s_assignment():
    """module/resnet.py:198-231: a stage's first block is built without a rate, block i >= 1 takes linspace(0, rate, blocks)[i]; the
    number of active DropPath modules is the number of draws the reference made (golden: one recorded row per call)."""
    g = load_golden("tiny_resnet_droppath")
    model, _ = build_model(CASES["tiny_resnet_droppath"])
    bb = model.encoder.adaptor.image_resnet.embed_images
    rates = [[round(b.drop_path.drop_prob, 6) for b in layer] for layer in (bb.layer1, bb.layer2, bb.layer3)]
    assert rates[0] == [0.0, 0.15, 0.3] and rates[1] == [0.0, 0.1, 0.2, 0.3]
    assert rates[2] == [0.0] + [round(float(v), 6) for v in torch.linspace(0, 0.3, 6)[1:]]
    assert sum(r > 0 for layer in rates for r in layer) == g["droppath_keep"].shape[0]
    assert g["droppath_keep"].shape[1] == 4 and 0 < float(g["droppath_keep"].mean()) < 1      # some rows kept, some dropped


def test_activation_checkpointing_flags_are_accepted_and_change_nothing():
    """checkpoint_activations / offload_activations (model/ofa.py:349-350, model/transformer.py:50-51, 68-72): same module tree and
    state-dict schema as the plain model; one warning says that the activations stay resident."""
    import copy
    import warnings
    from ofasys_amd.model import transformer as T
    plain, _ = build_model(CASES["tiny_text"])
    case = copy.deepcopy(CASES["tiny_text"])
    case["overrides"] = dict(case["overrides"], checkpoint_activations=True, checkpoint_adaptor_activations=True)
    T._CKPT_NOTED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model, _ = build_model(case)
    assert sum("activations are kept resident" in str(x.message) for x in w) == 1
    assert list(model.state_dict()) == list(plain.state_dict())
    assert [type(m) for m in model.modules()] == [type(m) for m in plain.modules()]


def test_layerdrop_survivors_match_the_reference_s_draws():
    """model/transformer.kept_layers against the reference's LayerDropModuleList iterated on the same seeds (tests/golden/
    layerdrop.json, oracle/gen_layerdrop_golden.py): same survivors, same number of random numbers consumed per iteration (the
    second iteration of a seed matches too), every layer in evaluation mode."""
    import json
    from ofasys_amd.model.transformer import kept_layers
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "layerdrop.json")))
    assert len(G["cases"]) >= 12
    for c in G["cases"]:
        layers = list(range(c["n"]))
        torch.manual_seed(c["seed"])
        assert kept_layers(layers, c["p"], True) == c["train_first"], c
        if c["p"] > 0:                      # (p == 0 keeps everything without drawing; the reference draws and keeps everything)
            assert kept_layers(layers, c["p"], True) == c["train_second"], c
        assert kept_layers(layers, c["p"], False) == c["eval"] == layers


def test_pack_plan_layout_is_a_bijection_on_the_non_pad_positions():
    """packing.build_pack_plan (host logic of the ragged layout): every non-pad position of the padded batch appears exactly once in
    the packed index, in order, each sample on an 8-row boundary; the inverse map undoes it; rows round up to the bucket; the segment
    tables carry (offset, length) of queries and keys; `*_prefix` tells whether packed row r of a sample is padded position r."""
    from ofasys_amd.packing import ALIGN, build_pack_plan
    g = torch.Generator().manual_seed(3)
    for trial in range(20):
        B, Ts, Tt = int(torch.randint(1, 9, (1,), generator=g)), int(torch.randint(1, 70, (1,), generator=g)), int(torch.randint(1, 40, (1,), generator=g))
        if trial % 2:                                  # right-padded rows (a prefix of every row is valid)
            el = torch.randint(1, Ts + 1, (B,), generator=g)
            enc = torch.arange(Ts)[None, :] >= el[:, None]
        else:                                          # two ragged slots side by side: padding in the middle of a row
            enc = torch.rand(B, Ts, generator=g) < 0.3
            enc[:, 0] = False
        dl = torch.randint(1, Tt + 1, (B,), generator=g)
        dec = torch.arange(Tt)[None, :] >= dl[:, None]
        plan = build_pack_plan(enc, dec, bucket=64, dec_bucket=32)
        for mask, index, inverse, seg, bucket in ((enc, plan.enc_index, plan.enc_inverse, plan.enc_self, 64),
                                                  (dec, plan.dec_index, plan.dec_inverse, plan.dec_self, 32)):
            T = mask.shape[1]
            want = torch.nonzero(~mask.reshape(-1)).squeeze(1)
            got = index[index >= 0]
            assert torch.equal(got, want)                                       # every valid position once, in (sample, position) order
            assert index.numel() % bucket == 0 and index.numel() >= bucket
            assert torch.equal(inverse[want], torch.nonzero(index >= 0).squeeze(1)) and int((inverse >= 0).sum()) == want.numel()
            tab = seg.table
            assert tab.dtype == torch.int32 and tab.shape == (B, 4)
            for b in range(B):
                off, n = int(tab[b, 0]), int(tab[b, 1])
                assert off % ALIGN == 0 and n == int((~mask[b]).sum())
                assert torch.equal(index[off:off + n] - b * T, torch.nonzero(~mask[b]).squeeze(1))
                assert b + 1 == B or int(tab[b + 1, 0]) >= off + n
            assert seg.max_q >= int(tab[:, 1].max()) and seg.max_q % 32 == 0
        assert plan.enc_tokens == int((~enc).sum()) and plan.dec_tokens == int((~dec).sum())
        assert torch.equal(plan.cross.table[:, :2], plan.dec_self.table[:, :2]) and torch.equal(plan.cross.table[:, 2:], plan.enc_self.table[:, 2:])
        assert plan.dec_prefix and plan.enc_prefix == all(bool((~enc[b, :int((~enc[b]).sum())]).all()) for b in range(B))


def test_integer_paths_bit_exact():
    g = load_golden("tiny_text")
    model, d = build_model(CASES["tiny_text"])
    b = model.encoder.adaptor.text.token_rp_bucket
    assert zlib.crc32(b.contiguous().numpy().tobytes()) == int(g["token_rp_bucket_crc"][0])
    assert np.array_equal(b[:300:7, :300:7].numpy(), g["token_rp_bucket_corner"])
    gb = load_golden("box_bins")
    d.add_bins(int(gb["num_bins"][0]))
    for row, want in zip(gb["coords"], gb["bins"]):
        assert [t - d.bin_start for t in d.box_to_tokens(row, int(gb["max_image_size"][0]))] == list(want)


def test_bucket_tables_bit_exact_at_every_adaptor_size():
    """make_token_bucket_position / make_image_bucket_position (re-derived as 1-D distance / 2-D offset lookups, adaptor/text.py and
    adaptor/image_resnet.py here) against the CRC-32 of the tables the REFERENCE's own builders return (tests/golden/bucket_tables.json,
    oracle/gen_bucket_crc.py): text 256 / 1024, the audio adaptor's 1024 / 4096 -- where one band edge is decided by float32 rounding --
    and a few more sizes; the oracle's restatement is held to the same CRCs."""
    import json
    import os
    from ofasys_amd.adaptor.image_resnet import make_image_bucket_position
    from ofasys_amd.adaptor.text import make_token_bucket_position
    from oracle import restate
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bucket_tables.json")))
    for e in G["token"]:
        for fn in (make_token_bucket_position, restate.make_token_bucket_position):
            t = fn(e["bucket_size"], e["max_position"])
            assert str(t.dtype) == e["dtype"] and zlib.crc32(t.contiguous().numpy().tobytes()) == e["crc32"], (fn.__module__, e)
    for e in G["image"]:
        for fn in (make_image_bucket_position, restate.make_image_bucket_position):
            t = fn(e["bucket_size"], e["num_relative_distance"])
            assert str(t.dtype) == e["dtype"] and zlib.crc32(t.contiguous().numpy().tobytes()) == e["crc32"], (fn.__module__, e)


def test_slot_attributes_and_adaptor_routing():
    from ofasys_amd import ModalityType, Slot
    model, _ = build_model(CASES["base_patch"])
    ga = model.encoder.adaptor
    s = Slot(ModalityType.IMAGE, True, None, attributes="adaptor=image_patch_embed,foo")
    assert s.has_attr("foo") and s.get_attr("adaptor") == "image_patch_embed" and s.attr2kwargs()["foo"] is True
    assert ga.get_adaptor(s) is ga.image_patch_embed
    for mod in (ModalityType.TEXT, ModalityType.BOX, ModalityType.STRUCT, ModalityType.MOTION, ModalityType.PHONE,
                ModalityType.CATEGORY):
        assert ga.get_adaptor(Slot(mod, True, None)) is ga.text          # adaptor/general.py:36-46
    assert "image_patch_embed" in model.decoder.adaptor.name2adaptor and not hasattr(model.decoder, "cross_pos_q_linear")
    assert [m.name for m in ModalityType] == ["TEXT", "IMAGE", "BOX", "AUDIO", "MOTION", "PHONE", "VIDEO", "STRUCT", "CATEGORY"]


def test_config_isolation_between_models():
    # the reference leaks adaptor configs between models through shared dataclass defaults; we must not
    m1, _ = build_model(CASES["tiny_text"])
    m2, _ = build_model(CASES["base_patch"])
    assert m1.cfg.adaptor.text.embed_dim == 256 and m2.cfg.adaptor.text.embed_dim == 768
    assert m1.cfg.encoder_embed_dim == 256 and m2.cfg.decoder_attention_heads == 12


def test_flat_param_arena():
    from ofasys_amd.trainer import FlatParams
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in lin.parameters()]
    fp = FlatParams(lin)
    for p, b, o in zip(lin.parameters(), before, fp.offsets):
        assert torch.equal(p.detach(), b) and o % 8 == 0
        assert p.data_ptr() == fp.flat.data_ptr() + o * 4 and p.grad.data_ptr() == fp.grad.data_ptr() + o * 4
    lin(torch.randn(2, 5)).sum().backward()
    assert float(fp.grad.abs().sum()) > 0
    fp.zero_grad()
    assert float(fp.grad.abs().sum()) == 0


def test_flat_param_arena_keeps_spatial_conv_weights_in_gemm_order():
    """Spatial convolution weights sit in the arena as [Cout][kh][kw][Cin] (channels_last strides of the torch shape): the im2col GEMM
    reads the parameter and the weight-gradient GEMM writes its arena gradient without a permuted copy.  Values, shapes and the
    state dict are unchanged; autograd's own accumulation (CPU here) lands in the right elements."""
    from ofasys_amd.trainer import FlatParams
    net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, padding=1, bias=False), torch.nn.Conv2d(16, 8, 1, bias=False),
                              torch.nn.Conv2d(3, 8, 7, bias=False))        # 3x3: eligible; 1x1 and 3*7*7 = 147 (not a multiple of 8): plain
    before = {k: v.clone() for k, v in net.state_dict().items()}
    fp = FlatParams(net)
    w3, w1, w7 = net[0].weight, net[1].weight, net[2].weight
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]) and v.shape == before[k].shape
    assert w3.permute(0, 2, 3, 1).is_contiguous() and not w3.is_contiguous()
    assert w3.grad.permute(0, 2, 3, 1).is_contiguous() and w3.grad.shape == w3.shape
    assert w1.is_contiguous() and w7.is_contiguous()
    o3 = fp.offsets[0]
    assert torch.equal(fp.flat[o3:o3 + w3.numel()].view(16, 3, 3, 8), before["0.weight"].permute(0, 2, 3, 1))
    x = torch.randn(2, 8, 5, 5)
    ref = torch.nn.functional.conv2d(x, before["0.weight"].clone().requires_grad_(True), padding=1)
    wr = before["0.weight"].clone().requires_grad_(True)
    torch.nn.functional.conv2d(x, wr, padding=1).sum().backward()
    net[0](x).sum().backward()
    assert torch.allclose(net[0](x), ref) and torch.allclose(w3.grad, wr.grad, atol=1e-5)
    assert torch.allclose(fp.grad[o3:o3 + w3.numel()].view(16, 3, 3, 8), wr.grad.permute(0, 2, 3, 1), atol=1e-5)
    fp.grad.zero_()
    w3.grad = None                                  # autograd replaced it: zero_grad re-points it at the arena, same layout
    fp.zero_grad()
    assert w3.grad.data_ptr() == fp.grad.data_ptr() + o3 * 4 and w3.grad.permute(0, 2, 3, 1).is_contiguous()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.randn_like(v) for k, v in sd.items()})      # loading keeps the arena views
    assert w3.data_ptr() == fp.flat.data_ptr() + o3 * 4 and w3.permute(0, 2, 3, 1).is_contiguous()


def test_flat_param_arena_leaves_the_patch_embedding_projection_contiguous():
    """the [D, 3, p, p] projection of image_patch_embed is read by ops.PatchEmbedFn as a [D, 3*p*p] VIEW (columns in
    (c, ph, pw) order), so the arena must not store it channels_last like the im2col convolutions' weights."""
    from ofasys_amd.trainer import FlatParams
    model, _ = build_model(CASES["base_patch"], "cpu", torch.float32)
    FlatParams(model)
    w = next(m for m in model.modules() if hasattr(m, "_ofa_plain_conv_weights")).proj.weight
    assert w.dim() == 4 and w.is_contiguous() and w.grad.is_contiguous()
    assert w.reshape(w.shape[0], -1).data_ptr() == w.data_ptr()               # a view, no copy


def test_owner_token_is_not_reused_like_id():
    """process-wide plan caches are keyed on a token object held by the owner, not on id(owner) (recycled after gc)."""
    import copy
    import gc
    from ofasys_amd import ops
    a = torch.nn.Linear(2, 2)
    ta = ops.owner_token(a)
    assert ops.owner_token(a) is ta and hash(ta) == hash(ops.owner_token(a))
    b = copy.deepcopy(a)
    assert ops.owner_token(b) is not ta and ops.owner_token(b) != ta
    key = ("text", ta)
    del a
    gc.collect()
    c = torch.nn.Linear(2, 2)                                                  # may land on a's address: its token still differs
    assert ("text", ops.owner_token(c)) != key


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd.distributed import GradBucketReducer, all_reduce_scalars
    from ofasys_amd.trainer import FlatParams
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    unused = torch.nn.Linear(4, 4)           # never touched by backward (find_unused_parameters case)
    model.add_module("unused", unused)
    fp = FlatParams(model)
    red = GradBucketReducer(fp.params, fp.grad, fp.offsets, None, bucket_bytes=256)     # several buckets
    assert len(red.buckets) > 2
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(6, 16, generator=g) for _ in range(2)]
    for step in range(2):                     # step 0 learns the contribution counts, step 1 overlaps bucket launches
        fp.zero_grad()
        red.begin_step("2mb")
        for x in xs:                          # two micro-batches accumulate; buckets fire on the LAST contribution
            model[3](model[2](model[1](model[0](x)))).sum().backward()
        if step == 1:
            assert any(red._launched)             # buckets fired from inside backward (overlap armed)
        red.finish()
    n = all_reduce_scalars(torch.tensor([6.0 * 2], dtype=torch.float64))
    q.put((rank, fp.grad.clone().numpy(), float(n)))       # by value: a shared-memory tensor can outlive its worker
    dist.destroy_process_group()


def test_dp_bucket_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    res = [(r, torch.from_numpy(g), n) for r, g, n in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]) and res[0][2] == 24.0
    # reference: single-process sum of both ranks' gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        for _ in range(2):
            model(torch.randn(6, 16, generator=g)).sum().backward()
    want = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 8)) for p in model.parameters()])
    got = res[0][1]
    assert torch.allclose(got[: want.numel()], want, atol=1e-5) and float(got[want.numel():].abs().sum()) == 0.0


def _dp_mixed_mode_worker(rank, world, port, q):
    """Rank 0 runs its second step ARMED (counts learned for its structure key), rank 1 sees a NEW key in that step (learning:
    everything goes out at finish()) -- what per-rank padding does to sample_structure under DP.  The module registered last is
    used first, so its bucket (index 0) completes LAST in backward: completion order != index order."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd.distributed import GradBucketReducer
    from ofasys_amd.trainer import FlatParams
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(16, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 16)
    model = torch.nn.ModuleList([a, b, c])
    fp = FlatParams(model)
    red = GradBucketReducer(fp.params, fp.grad, fp.offsets, None, bucket_bytes=512)
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(4, 16, generator=g)
    orders = []
    for step in range(2):
        fp.zero_grad()
        red.begin_step("same" if rank == 0 else f"padded-to-{step}")
        b(a(c(x))).sum().backward()
        if step == 1:
            assert (red.expected is not None) == (rank == 0)
            if rank == 0:
                assert 0 < red._next or not any(red._ready), "armed rank"
                assert red.launch_order == sorted(red.launch_order)
        red.finish()
        orders.append(list(red.last_launch_order))
    q.put((rank, fp.grad.clone().numpy(), orders))
    dist.destroy_process_group()


def test_dp_collective_order_is_rank_invariant_across_modes():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_mixed_mode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, g1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(g0, g1)
    for _, _, orders in res:
        for o in orders:
            assert o == list(range(len(o))) and len(o) >= 3          # every step, every mode: bucket 0, 1, 2, ...
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(16, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 16)
    for rank in range(2):
        x = torch.randn(4, 16, generator=torch.Generator().manual_seed(7 + rank))
        b(a(c(x))).sum().backward()
    want = torch.cat([p.grad.reshape(-1) for m in (a, b, c) for p in m.parameters()])
    assert torch.allclose(g0[: want.numel()], want, atol=1e-5)


def _dp_layerdrop_worker(rank, world, port, q, dynamic):
    """LayerDrop under DP: every rank keeps its OWN subset of layers each step, so a parameter's contribution count
    changes from step to step under one structure key.  dynamic=True is what TrainStep passes for such a model."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd.distributed import GradBucketReducer
    from ofasys_amd.trainer import FlatParams
    torch.manual_seed(0)
    layers = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(4)])
    fp = FlatParams(layers)
    red = GradBucketReducer(fp.params, fp.grad, fp.offsets, None, bucket_bytes=512)
    keep = [[(0, 2), (0, 1, 2, 3), (1, 2, 3)], [(1, 3), (0, 1, 2), (0, 3)]][rank]     # step 1 keeps layers the (learning) step 0 had dropped
    x = torch.randn(4, 16, generator=torch.Generator().manual_seed(7 + rank))
    err, grads = None, []
    try:
        for step in range(3):
            fp.zero_grad()
            red.begin_step("same-structure", dynamic=dynamic)
            h = x
            for i in keep[step]:
                h = layers[i](h)
            h.sum().backward()
            red.finish()
            grads.append(fp.grad.clone().numpy())
            assert red.last_launch_order == list(range(len(red.buckets)))
            assert not dynamic or (red.expected is None and not red.knows("same-structure"))
    except RuntimeError as e:
        err = str(e)
    q.put((rank, grads, err))
    if err is None:
        dist.destroy_process_group()


@pytest.mark.parametrize("dynamic", [True, False])
def test_dp_reducer_with_layerdrop_never_arms(dynamic):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 1000 + (0 if dynamic else 1)
    procs = [ctx.Process(target=_dp_layerdrop_worker, args=(r, 2, port, q, dynamic)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():                              # the armed variant: one rank raised, its peer waits in a collective
            p.terminate()
    if not dynamic:                                   # what the advice describes: a kept layer the learned step had dropped
        assert any(e and "were learned" in e for _, _, e in res)
        return
    assert all(e is None for _, _, e in res)
    torch.manual_seed(0)
    layers = [torch.nn.Linear(16, 16) for _ in range(4)]
    keep = [[(0, 2), (0, 1, 2, 3), (1, 2, 3)], [(1, 3), (0, 1, 2), (0, 3)]]
    for step in range(3):
        for m in layers:
            m.zero_grad()
        for rank in range(2):
            h = torch.randn(4, 16, generator=torch.Generator().manual_seed(7 + rank))
            for i in keep[rank][step]:
                h = layers[i](h)
            h.sum().backward()
        want = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for m in layers for p in m.parameters()])
        g0, g1 = torch.from_numpy(res[0][1][step]), torch.from_numpy(res[1][1][step])
        assert torch.equal(g0, g1) and torch.allclose(g0[: want.numel()], want, atol=1e-5)


def _bn_exchange_worker(rank, world, port, q):
    """The one exchange of a SyncBatchNorm layer, over CPU-hosted partials: [2C + 1] fp64 (sum x, sum x^2, row count) per rank."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd import ops
    x = torch.randn(40, 8, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    mine = x[:13] if rank == 0 else x[13:]                          # uneven row blocks
    sums = torch.cat([mine.sum(0), (mine * mine).sum(0), torch.tensor([float(mine.shape[0])], dtype=torch.float64)])
    ops._bn_all_reduce(sums, None)
    q.put((rank, sums.numpy()))
    dist.destroy_process_group()


def test_sync_bn_exchange_over_two_gloo_ranks_gives_whole_batch_statistics():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 1000
    procs = [ctx.Process(target=_bn_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0], res[1])
    s = torch.from_numpy(res[0])
    C = 8
    R = float(s[2 * C])
    assert R == 40.0
    mean, var = s[:C] / R, s[C:2 * C] / R - (s[:C] / R) ** 2          # what bn_finalize_kernel computes from the reduced sums
    x = torch.randn(40, 8, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    bn = torch.nn.BatchNorm1d(C, momentum=0.1, eps=1e-3).double().train()
    y = bn(x)
    assert torch.allclose(mean, x.mean(0)) and torch.allclose(var, x.var(0, unbiased=False))
    assert torch.allclose(bn.running_mean, 0.1 * mean) and torch.allclose(bn.running_var, 0.9 + 0.1 * var * R / (R - 1))
    assert torch.allclose(y, (x - mean) / torch.sqrt(var + 1e-3))


def test_tool_scripts_compile():
    """tools/*.py (profiling / analysis helpers referenced by DESIGN.md) must at least parse."""
    import glob
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(scripts) >= 15
    for s in scripts:
        py_compile.compile(s, doraise=True)


def test_bench_self_launches_ranks_when_no_launcher_started_it():
    """`python bench.py --gpus 2` run BARE (no torch.distributed.run, no WORLD_SIZE): bench.py re-executes itself under
    torch.distributed.run with a 127.0.0.1 rendezvous, one rank per GPU, and rank 0 prints one JSON line.  --launch-check stops
    after the process-group round trip (gloo; the train step itself needs GPUs)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world"] == 2


class _SyncProbe(torch.autograd.Function):
    """Stands in for a SyncBatchNorm layer: one all-reduce on the layers' communicator in forward and one in backward."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        ctx.group = group
        s = x.detach().sum().reshape(1).clone()
        dist.all_reduce(s, group=group)
        return x + 0.0 * s

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        s = dy.sum().reshape(1).clone()
        dist.all_reduce(s, group=ctx.group)
        return dy + 0.0 * s, None


def _dp_syncbn_worker(rank, world, port, q, dynamic):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd import ops
    from ofasys_amd.distributed import GradBucketReducer
    from ofasys_amd.trainer import FlatParams
    bn = torch.nn.BatchNorm2d(4)
    bn._ofa_sync = True
    (group,) = ops._bn_sync_group(bn)
    own = group is not None and group is not dist.group.WORLD and ops._bn_sync_group(bn)[0] is group     # dedicated, created once
    torch.manual_seed(0)
    layers = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(4)])
    fp = FlatParams(layers)
    red = GradBucketReducer(fp.params, fp.grad, fp.offsets, None, bucket_bytes=512)
    grads, early = [], []
    for step in range(3):
        # rank 0 sees ONE structure (armed from step 1 on), rank 1 a new one every step (padded lengths differ): never armed
        sig = "same" if rank == 0 else f"len{step}"
        fp.zero_grad()
        red.begin_step(sig, dynamic=dynamic)
        h = torch.randn(4, 16, generator=torch.Generator().manual_seed(7 + rank + 10 * step))
        for i, lin in enumerate(layers):
            h = lin(h)
            if i == 1:
                h = _SyncProbe.apply(h, group)
        h.sum().backward()
        red.finish()
        grads.append(fp.grad.clone().numpy())
        early.append(red.last_early)
    q.put((rank, grads, early, own))
    dist.destroy_process_group()


@pytest.mark.parametrize("dynamic", [True, False])
def test_sync_bn_collectives_next_to_bucket_reducer_with_ranks_in_different_modes(dynamic):
    """SyncBatchNorm's exchanges run on their OWN communicator (ops.sync_bn_process_group), so a rank that launches gradient
    buckets from inside backward (structure learned) next to a rank that launches them at finish() (still learning: its padded lengths
    changed) interleaves the two kinds of collectives differently without mismatching either sequence; TrainStep additionally never
    arms the reducer when the model holds such layers (dynamic = True).  Both settings must give the summed gradients on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34500 + os.getpid() % 1000 + (0 if dynamic else 1)
    procs = [ctx.Process(target=_dp_syncbn_worker, args=(r, 2, port, q, dynamic)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][3] and res[1][3]                                    # the dedicated group, the same object on every call
    if dynamic:
        assert res[0][2] == [0, 0, 0] and res[1][2] == [0, 0, 0]      # nobody launched from inside backward
    else:
        assert res[0][2][1] > 0 and res[1][2] == [0, 0, 0]            # rank 0 armed, rank 1 not: the situation of the advice
    torch.manual_seed(0)
    layers = [torch.nn.Linear(16, 16) for _ in range(4)]
    for step in range(3):
        for m in layers:
            m.zero_grad()
        for rank in range(2):
            h = torch.randn(4, 16, generator=torch.Generator().manual_seed(7 + rank + 10 * step))
            for lin in layers:
                h = lin(h)
            h.sum().backward()
        want = torch.cat([p.grad.reshape(-1) for m in layers for p in m.parameters()])
        g0, g1 = torch.from_numpy(res[0][1][step]), torch.from_numpy(res[1][1][step])
        assert torch.equal(g0, g1) and torch.allclose(g0[: want.numel()], want, atol=1e-5)


def test_owner_token_entries_leave_the_plan_cache_with_their_owner():
    """process-wide caches keyed by an owner token drop the entries of a collected module (weakref.finalize)."""
    import gc
    from ofasys_amd import ops
    m = torch.nn.Linear(2, 2)
    key = ("text", ops.owner_token(m))
    other = ("text", ops.owner_token(torch.nn.Linear(2, 2)))          # (its owner is already gone: purged on collection)
    gc.collect()
    ops.SegmentPlan._cache[key] = object()
    keep = torch.nn.Linear(2, 2)
    key2 = ("image", ops.owner_token(keep), (3, 4))
    ops.SegmentPlan._cache[key2] = object()
    assert key in ops.SegmentPlan._cache and other not in ops.SegmentPlan._cache
    del m
    gc.collect()
    assert key not in ops.SegmentPlan._cache and key2 in ops.SegmentPlan._cache
    del ops.SegmentPlan._cache[key2]


def test_fan_out_backward_takes_the_pairwise_path_for_gradients_add_n_would_refuse():
    """FanOutFn.backward hands strided / mixed-dtype gradients to plain adds instead of ofa_add_n (contiguous, aligned, one dtype)."""
    from ofasys_amd import ops
    x = torch.randn(6, 8, requires_grad=True)
    a, b, c = ops.FanOutFn.apply(x, 3)
    ga = torch.randn(8, 6).t()                                         # not contiguous
    gb = torch.randn(6, 8, dtype=torch.float64)                        # another dtype
    (a * ga).sum().backward(retain_graph=True, inputs=[x])
    g1 = x.grad.clone()
    x.grad = None
    ((a * ga).sum() + (b.double() * gb).sum() + c.sum()).backward(inputs=[x])
    assert torch.allclose(g1, ga) and torch.allclose(x.grad, ga + gb.float() + 1.0, atol=1e-6)


def _dp_four_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd.distributed import GradBucketReducer
    from ofasys_amd.trainer import FlatParams
    torch.manual_seed(0)
    layers = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(6)])
    fp = FlatParams(layers)
    red = GradBucketReducer(fp.params, fp.grad, fp.offsets, None, bucket_bytes=600)
    grads, early, orders = [], [], []
    # every rank pads its own batches: the structure key (shapes are part of it) repeats on rank 0 at once, on rank 1 every other step,
    # on rank 2 every third step, never on rank 3 -- so in one and the same step some ranks launch buckets from inside backward
    # (armed: structure learned) and others launch all of them at finish()
    period = [1, 2, 3, 10 ** 6][rank]
    for step in range(6):
        fp.zero_grad()
        red.begin_step(("len", step % period))
        h = torch.randn(3 + rank, 16, generator=torch.Generator().manual_seed(100 * rank + step))      # uneven batch sizes too
        for lin in layers:
            h = torch.tanh(lin(h))
        h.sum().backward()
        red.finish()
        grads.append(fp.grad.clone().numpy())
        early.append(red.last_early)
        orders.append(list(red.last_launch_order))
    q.put((rank, grads, early, orders, len(red.buckets)))
    dist.destroy_process_group()


def test_dp_reducer_four_ranks_with_uneven_structures_per_rank():
    """four gloo ranks whose step structures repeat at different rates, so that armed ranks (buckets launched from
    inside backward) and learning ranks (everything at finish()) meet in the same step, with different batch sizes per rank.  The
    sequence of collectives is the bucket index on every rank in every step, and every rank ends with the sum of the four gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_four_rank_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(4)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nb = res[0][4]
    assert nb >= 4
    for rank, _, early, orders, _ in res:
        assert all(o == list(range(nb)) for o in orders), (rank, orders)
    assert max(res[0][2]) > 0 and res[3][2] == [0] * 6                     # rank 0 overlapped, rank 3 never could
    assert any(res[0][2][s] > 0 and res[3][2][s] == 0 for s in range(6))   # ... in the SAME step
    torch.manual_seed(0)
    layers = [torch.nn.Linear(16, 16) for _ in range(6)]
    for step in range(6):
        for m in layers:
            m.zero_grad()
        for rank in range(4):
            h = torch.randn(3 + rank, 16, generator=torch.Generator().manual_seed(100 * rank + step))
            for lin in layers:
                h = torch.tanh(lin(h))
            h.sum().backward()
        want = torch.cat([p.grad.reshape(-1) for m in layers for p in m.parameters()])
        for rank in range(4):
            g = torch.from_numpy(res[rank][1][step])
            assert torch.allclose(g[: want.numel()], want, atol=1e-5), (rank, step)
        assert all(np.array_equal(res[0][1][step], res[r][1][step]) for r in range(1, 4))


def _dp_sharded_worker(rank, world, port, q):
    """Two steps of a toy data-parallel training loop in BOTH exchange forms on the same ranks: all-reduce -> every rank updates the
    whole arena, and reduce-scatter -> every rank updates the pieces it owns -> all-gather.  The update is an elementwise stand-in for
    ofa_adam_step (a function of parameter, gradient, two moments and a global clip coefficient from the gradient norm)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofasys_amd.distributed import GradBucketReducer
    from ofasys_amd.trainer import FlatParams
    out = {}
    for shard in (False, True):
        torch.manual_seed(0)
        layers = torch.nn.ModuleList([torch.nn.Linear(12, 12) for _ in range(5)] + [torch.nn.Linear(12, 7)])
        fp = FlatParams(layers)
        red = GradBucketReducer(fp.params, fp.grad, fp.offsets, None, bucket_bytes=700, shard=shard)
        assert len(red.buckets) >= 3
        red.overlap = False        # every bucket at finish(): the gradients are rounded to integers between backward and the exchange
        m, v = torch.zeros_like(fp.flat), torch.zeros_like(fp.flat)
        norms = []
        for step in range(3):
            fp.zero_grad()
            red.begin_step("s")
            # integer-valued data: every sum below is exact in fp32, so the two forms must agree BIT FOR BIT whatever order a
            # collective adds the ranks in
            h = torch.randint(-2, 3, (4, 12), generator=torch.Generator().manual_seed(17 * rank + step)).float()
            for lin in layers:
                h = torch.tanh(lin(h))
            (h.sum() * 256).backward()
            with torch.no_grad():
                fp.grad.round_()                             # integer-valued gradients on every rank BEFORE the exchange
            red.finish()
            owned = red.owned_ranges()
            gsq = torch.zeros(1, dtype=torch.float64)        # (fp64: sums of squares of integers stay exact in any order)
            for lo, hi, counted in owned:
                if counted:
                    gsq += fp.grad[lo:hi].double().pow(2).sum()
            if shard:
                dist.all_reduce(gsq)
            norms.append(float(gsq))
            coef = 1.0 / max(float(gsq.sqrt()), 1.0)
            with torch.no_grad():
                for lo, hi, _ in owned:
                    g = fp.grad[lo:hi] * coef
                    m[lo:hi].mul_(0.5).add_(g, alpha=0.5)
                    v[lo:hi].mul_(0.75).add_(g * g, alpha=0.25)
                    fp.flat[lo:hi].sub_(0.125 * m[lo:hi] / (v[lo:hi].sqrt() + 1.0))
                    fp.flat[lo:hi].mul_(8).round_().div_(8)             # keep the parameters on a grid: the next step's sums stay exact
            red.gather_params(fp.flat)
        cover = sorted((lo, hi) for r in range(world) for b in range(len(red.buckets))
                       for lo, hi in [red.piece(b, r)[:2], (red.piece(b, r)[2], red.buckets[b][1])] if hi > lo) if shard else None
        out[shard] = (fp.flat.clone().numpy(), norms, cover, fp.numel, [(lo, hi) for lo, hi, _ in owned])
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_exchange_equals_all_reduce_form(world):
    """reduce_scatter_tensor -> owners update 1 / world of the arena -> all_gather_into_tensor over arena slices
    (GradBucketReducer(shard=True)) against the all-reduce form, 2 and 4 gloo ranks: the same parameters on every rank and in both
    forms, bit for bit; the same global gradient norm; the ranks' pieces and the replicated tails tile the arena exactly once."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + os.getpid() % 1000 + world
    procs = [ctx.Process(target=_dp_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = res[0][False]
    for r in range(world):
        for shard in (False, True):
            flat, norms, cover, numel, owned = res[r][shard]
            assert np.array_equal(flat, ref[0]), (r, shard)
            assert norms == ref[1], (r, shard, norms, ref[1])
        assert len(res[r][True][4]) > len(res[r][False][4]) == 1
    cover, numel = res[0][True][2], res[0][True][3]
    pieces = sorted(set(cover))
    # every rank's piece once, every tail `world` times (replicated): as a set, a partition of [0, numel)
    assert pieces[0][0] == 0 and pieces[-1][1] == numel and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    assert float(np.abs(ref[0]).sum()) > 0 and ref[1][0] > 0


def test_pointer_audit_classifies_pins_by_pool_and_owner():
    """lib.PointerAudit (the capture-time pointer audit) on a synthetic allocator snapshot: only tensors inside
    default-pool segments are pinned, each once per storage, labelled with the first C-ABI call that used them; `foreign` leaves
    out what the engine owns."""
    from ofasys_amd.lib import PointerAudit

    class FakeStorage:
        def __init__(self, addr, nbytes):
            self._a, self._n = addr, nbytes

        def data_ptr(self):
            return self._a

        def nbytes(self):
            return self._n

    class FakeTensor:
        is_cuda = True

        def __init__(self, addr, shape=(4,), storage=None):
            self._a, self.shape, self.dtype = addr, shape, torch.float32
            self._s = storage or FakeStorage(addr, 16)

        def data_ptr(self):
            return self._a

        def untyped_storage(self):
            return self._s

    snap = [{"address": 0x1000, "total_size": 0x1000, "segment_pool_id": (0, 0), "blocks": []},
            {"address": 0x8000, "total_size": 0x2000, "segment_pool_id": (0, 1), "blocks": []},        # a graph's private pool
            {"address": 0x4000, "total_size": 0x1000, "segment_pool_id": (0, 0), "blocks": []}]
    au = PointerAudit.from_snapshot(snap)
    assert au.in_default_pool(0x1000) and au.in_default_pool(0x1fff) and not au.in_default_pool(0x2000)
    assert au.in_default_pool(0x4800) and not au.in_default_pool(0x8000) and not au.in_default_pool(0x0)
    param = FakeTensor(0x1100)
    cache = FakeTensor(0x4100, shape=(7, 7))
    st = FakeStorage(0x4200, 64)
    view_a, view_b = FakeTensor(0x4200, storage=st), FakeTensor(0x4210, storage=st)
    priv = FakeTensor(0x8100)
    for t in (param, cache, view_a):
        au.see(t)
    au.label("ofa_first")
    for t in (view_b, priv, cache):
        au.see(t)
    au.label("ofa_second")
    assert len(au.pins) == 3 and {id(t) for t in au.tensors()} == {id(param), id(cache), id(view_a)}
    rep = au.foreign([param, None])
    assert sorted(rep) == sorted([("ofa_first", (7, 7), "float32", 16), ("ofa_first", (4,), "float32", 64)])
