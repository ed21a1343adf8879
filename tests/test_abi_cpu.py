"""CPU: the C-ABI library loads and exports every symbol include/ofasys_amd.h declares; no compute without a GPU."""
import ctypes
import os

import pytest
import torch

from ofasys_amd import lib as L


def test_header_parses_and_library_exports_every_symbol():
    protos = L.parse_header()
    assert len(protos) >= 40
    for must in ("ofa_gemm", "ofa_attn_fwd", "ofa_attn_bwd", "ofa_layernorm_fwd", "ofa_layernorm_bwd",
                 "ofa_scaled_softmax_fwd", "ofa_scaled_masked_softmax_fwd", "ofa_scaled_upper_triang_masked_softmax_fwd",
                 "ofa_scaled_softmax_bwd", "ofa_scaled_masked_softmax_bwd", "ofa_scaled_upper_triang_masked_softmax_bwd",
                 "ofa_get_batch_per_block", "ofa_embedding_bwd", "ofa_cross_entropy_fwd", "ofa_adam_step", "ofa_version",
                 "ofa_last_error", "ofa_im2col_patch", "ofa_bias_block_add"):
        assert must in protos
    assert os.path.exists(L.LIB_PATH), "build with `python __graft_entry__.py` first"
    cdll = ctypes.CDLL(L.LIB_PATH)
    for name in protos:
        getattr(cdll, name)          # AttributeError == declared but not exported
    h = L.lib()
    assert h.cdll.ofa_version() >= 100


def test_host_only_entry_points():
    h = L.lib()
    from oracle import restate
    for a in [(128, 128, 2, 4), (64, 448, 2, 4), (4, 16, 1, 1), (8, 4096, 1, 1), (16, 100, 3, 3)]:
        assert h.cdll.ofa_get_batch_per_block(*a) == restate.get_batch_per_block(*a)
    assert h.cdll.ofa_layernorm_bwd_ws_rows() > 0


def test_gemm_group_plan_is_host_only():
    """ofa_gemm_group_plan: one K-slice length for the group, at most 256 workgroups of 256 x 256 tiles in total."""
    import ctypes as C
    from ofasys_amd import kernels as K
    h = L.lib()

    def plan(shapes, dt=L.BF16):
        arr = (K._GroupItem * len(shapes))()
        for it, (m, n, k) in zip(arr, shapes):
            it.a, it.b, it.lda, it.ldb, it.m, it.n, it.k = 4096, 8192, m, n, m, n, k
        h.call("ofa_gemm_group_plan", C.addressof(arr), len(shapes), dt)
        return [it.splits for it in arr]

    assert plan([(768, 3072, 13312), (3072, 768, 13312), (2304, 768, 13312), (768, 768, 13312)]) == [2, 2, 2, 2]
    sp = plan([(2304, 768, 3072), (768, 768, 3072), (1536, 768, 13312), (768, 3072, 3072), (3072, 768, 3072)])
    tiles = [27, 9, 18, 36, 36]
    assert sum(t * s for t, s in zip(tiles, sp)) <= 256 and sp[2] > sp[0] >= 1        # the long contraction gets more slices
    assert plan([(256, 256, 64)]) == [1]
    assert plan([(4096, 4096, 1024), (4096, 4096, 512)]) == [1, 1]                     # more than one round already: no slicing
    assert plan([(768, 768, 100)]) == [1]                                              # any row count (round 4: zero rows inside the kernel)
    with pytest.raises(L.OfaError, match="gemm_group"):
        plan([(772, 768, 128)])                                                        # m % 8
    with pytest.raises(L.OfaError, match="gemm_group"):
        plan([(768, 768, 128)], L.F32)
    with pytest.raises(L.OfaError, match="gemm_group"):
        plan([(256, 256, 64)] * 17)
    # two base-size encoder layers in one group: a round of one-slice products (no slabs: ops._Wgrads.FLUSH_TILES)
    assert plan([(768, 3072, 13312), (3072, 768, 13312), (2304, 768, 13312), (768, 768, 13312)] * 2) == [1] * 8


def test_shipped_join_backward_is_the_row_per_wave_kernel():
    """Since round 6 the row-per-wave residual-join BACKWARD is the product's kernel (round 5 held it back for a fault that turned out to
    be the dK/dV attention kernel's: profiles/round6_graph_fault_root_cause.txt): 16-bit rows of 256 k columns hand one dropout keep bit
    per element from the forward to the backward, and the partial rows are sized for the row kernel; other shapes keep the split-row
    kernel and no keep bits."""
    h = L.lib()
    assert h.cdll.ofa_join_keep_bytes(13312, 768, L.BF16) == 13312 * 768 // 8 * 4 // 3 or h.cdll.ofa_join_keep_bytes(13312, 768, L.BF16) >= 13312 * 768 // 8
    assert h.cdll.ofa_join_keep_bytes(100, 1024, L.F16) >= 100 * 1024 // 8
    assert h.cdll.ofa_join_keep_bytes(100, 1000, L.BF16) == 0 and h.cdll.ofa_join_keep_bytes(100, 768, L.F32) == 0
    assert h.cdll.ofa_join_bwd_slots(13312, 768, L.BF16) == 256 and h.cdll.ofa_join_bwd_slots(60, 768, L.BF16) == 8     # 8 rows per block
    assert h.cdll.ofa_join_bwd_slots(60, 768, L.F32) == 20                                                               # split-row kernel


def test_status_codes_not_asserts():
    h = L.lib()
    # argument validation happens before any launch, so it is observable without a GPU
    with pytest.raises(L.OfaError, match="layernorm"):
        h.call("ofa_layernorm_fwd", None, None, None, None, None, None, 4, 6, 1e-5, L.BF16, None)   # cols % 8 != 0
    with pytest.raises(L.OfaError, match="sk"):
        h.call("ofa_scaled_softmax_fwd", 1, 1, 1.0, 1, 1, 4, 5000, L.F32, None)                       # sk > 4096
    with pytest.raises(L.OfaError, match="sk"):                                                        # same precondition, backward
        h.call("ofa_scaled_masked_softmax_bwd", 1, 1, 1, 1.0, 1, 1, 4, 5000, L.F16, None)
    with pytest.raises(L.OfaError, match="dtype"):
        h.call("ofa_scaled_upper_triang_masked_softmax_bwd", 1, 1, 1, 1.0, 1, 8, 7, None)
    with pytest.raises(L.OfaError, match="dtype"):                                                     # an unknown dtype code
        h.call("ofa_layernorm_fwd", 1, 1, 1, 1, 1, 1, 4, 8, 1e-5, 7, None)
    with pytest.raises(L.OfaError, match="bf16"):
        h.call("ofa_attn_fwd", 1, 1, 1, None, None, None, L.F32, 1, None, 1, 1, 32, 32, 32, 64, 64, 64, 1.0, 0, None, 0, 0, L.F32, None)


def test_no_cpu_fallback():
    x = torch.randn(4, 8)
    with pytest.raises(L.OfaError, match="no CPU fallback"):
        L.ptr(x)
    from ofasys_amd import ops
    with pytest.raises(L.OfaError):
        ops.layer_norm(x, torch.ones(8), torch.zeros(8))
