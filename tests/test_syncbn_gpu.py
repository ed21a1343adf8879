"""GPU: `sync_bn` (reference adaptor/image_resnet.py:53-56, 87-90 -> module/layer.py:26-27 nn.SyncBatchNorm).

torch's SyncBatchNorm refuses CPU tensors, so the reference cannot record a fixture in the build container.  What pins the path
instead is the identity it exists for: BatchNorm synchronised over R ranks that hold row blocks of one batch IS plain BatchNorm
over the whole batch -- and plain BatchNorm is pinned to the reference by the tiny_resnet goldens (tests/test_model_gpu.py).  Two
processes share the one GPU of the test box (gloo group, as test_bench_two_ranks_end_to_end_on_one_gpu does), hold UNEVEN row
blocks, and must reproduce the single-process layer: outputs, input gradients, running buffers, and parameter gradients that sum
to the single-process ones."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROWS, SPLIT, C = 1536, 640, 64          # rank 0 holds 640 rows, rank 1 holds 896


def _data(dtype):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(ROWS, C, generator=g) * 1.5 + 0.3
    dy = torch.randn(ROWS, C, generator=g)
    res = torch.randn(ROWS, C, generator=g)
    w = torch.rand(C, generator=g) + 0.5
    b = torch.randn(C, generator=g) * 0.1
    return [t.to(dtype) for t in (x, dy, res, w, b)]


def _layer(w, b, sync):
    bn = torch.nn.BatchNorm2d(C, momentum=0.1, eps=1e-3).cuda().to(w.dtype)
    with torch.no_grad():
        bn.weight.copy_(w)
        bn.bias.copy_(b)
    if sync:
        bn._ofa_sync = True
    bn.train()
    return bn


def _run(bn, x, dy, res, relu, with_res):
    from ofasys_amd import ops
    x = x.cuda().requires_grad_(True)
    r = res.cuda().requires_grad_(True) if with_res else None
    y = ops.batch_norm(x, bn, relu=relu, residual=r)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    out = {"y": y.detach().float().cpu(), "dx": x.grad.float().cpu(), "dw": bn.weight.grad.float().cpu(), "db": bn.bias.grad.float().cpu(),
           "rm": bn.running_mean.float().cpu(), "rv": bn.running_var.float().cpu()}
    if with_res:
        out["dres"] = r.grad.float().cpu()
    return out


def _worker(rank, world, port, q, dtype_name):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = getattr(torch, dtype_name)
    x, dy, res, w, b = _data(dtype)
    sl = slice(0, SPLIT) if rank == 0 else slice(SPLIT, ROWS)
    outs = {}
    for relu, with_res in ((False, False), (True, False), (True, True)):
        o = _run(_layer(w, b, True), x[sl], dy[sl], res[sl], relu, with_res)
        outs[(relu, with_res)] = {k: v.numpy() for k, v in o.items()}
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name,tol", [("float32", 2e-5), ("bfloat16", 1.6e-2)])
def test_sync_bn_over_two_ranks_equals_plain_bn_over_the_whole_batch(dtype_name, tol):
    import torch.multiprocessing as mp
    dtype = getattr(torch, dtype_name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, dtype_name)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x, dy, rs, w, b = _data(dtype)
    for relu, with_res in ((False, False), (True, False), (True, True)):
        want = _run(_layer(w, b, False), x, dy, rs, relu, with_res)          # one process, the whole batch, plain BatchNorm
        r0, r1 = res[0][(relu, with_res)], res[1][(relu, with_res)]

        def err(a, ref):
            return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        for k in ["y", "dx"] + (["dres"] if with_res else []):
            got = torch.cat([torch.from_numpy(r0[k]), torch.from_numpy(r1[k])])
            assert err(got, want[k]) < tol, (relu, with_res, k, err(got, want[k]))
        for k in ("dw", "db"):                                                 # rank-local sums: the gradient exchange adds them
            got = torch.from_numpy(r0[k]) + torch.from_numpy(r1[k])
            assert err(got, want[k]) < 2 * tol, (relu, with_res, k, err(got, want[k]))
        for k in ("rm", "rv"):                                                 # every rank holds the GLOBAL running statistics
            for r in (r0, r1):
                assert err(torch.from_numpy(r[k]), want[k]) < tol, (relu, with_res, k)
        print(f"MEASURED sync_bn {dtype_name} relu={relu} residual={with_res}: y / dx within {tol:.0e} of the whole-batch layer")


def test_image_resnet_adaptor_builds_sync_layers_with_the_reference_eps():
    """cfg.sync_bn: every BatchNorm of the trunk is a SynBatchNorm2d (eps 1e-3, module/layer.py:26-27); state-dict keys unchanged."""
    from oracle.cases import CASES
    from tests.model_util import build_model
    case = dict(CASES["tiny_resnet"])
    plain, _ = build_model(case, "cpu")
    case["adaptor_overrides"] = {**case["adaptor_overrides"], "image_resnet": {**case["adaptor_overrides"].get("image_resnet", {}), "sync_bn": True}}
    sync, _ = build_model(case, "cpu")
    assert list(plain.state_dict().keys()) == list(sync.state_dict().keys())
    bns = [m for m in sync.encoder.adaptor.image_resnet.embed_images.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert bns and all(getattr(m, "_ofa_sync", False) and m.eps == 1e-3 and m.momentum == 0.1 for m in bns)
    assert all(m.eps == 1e-5 for m in plain.encoder.adaptor.image_resnet.embed_images.modules() if isinstance(m, torch.nn.BatchNorm2d))
