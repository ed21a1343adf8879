"""GPU: the adaptor plugin contract beyond the built-in adaptors.

A user registers an adaptor with `@register_config("ofasys.adaptor", name, Cfg)` AFTER the package was imported, routes a slot to
it with the `adaptor=<name>` attribute, and its forward() may fill `AdaptorOutput.self_attn_bias` itself with one [B, A, n, n]
matrix PER SAMPLE -- the post-hook then leaves it alone (reference adaptor/base.py:183-189) and the general adaptor sums it into
the layer bias block-wise (adaptor/general.py:265-280).  The built-in adaptors never do this, so no golden vector of the reference
covers it; what pins it here is the reference's own batch semantics: a batch of B samples must give, for every sample, what that
sample gives alone -- and a batch of ONE takes the shared-bias path that the golden vectors do pin (tests/test_model_gpu.py)."""
import pytest
import torch

from oracle.cases import CASES
from tests.golden_util import case_inputs, rel_err
from tests.model_util import build_model, make_slots

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _register():
    from ofasys_amd import register_config
    from ofasys_amd.adaptor.text import TextAdaptor, TextAdaptorConfig
    from ofasys_amd.configure import ConfigStore
    if "text_own_bias" in ConfigStore().names("ofasys.adaptor"):
        return

    @register_config("ofasys.adaptor", "text_own_bias", TextAdaptorConfig)
    class OwnBiasTextAdaptor(TextAdaptor):
        """A text adaptor whose attention bias depends on the SAMPLE: the usual bucketed rel-pos values plus, per head, a learned
        weight (row 0 of the layer's table) wherever two positions hold the same token."""

        def forward(self, slot, **kwargs):
            out = super().forward(slot, **kwargs)
            tok = slot.value
            B, T = tok.shape
            same = (tok[:, :, None] == tok[:, None, :]).to(out.embed.dtype)                     # [B, T, T]
            out.self_attn_bias = []
            for idx in range(len(self.token_rel_pos_table_list)):
                values = self.get_rel_pos_bias(B, T, idx)                                       # [T, T, A]
                w = self.token_rel_pos_table_list[idx].weight[0].to(out.embed.dtype)            # [A]
                out.self_attn_bias.append(values.permute(2, 0, 1).unsqueeze(0) + w.view(1, -1, 1, 1) * same.unsqueeze(1))
            return out


def _step(model, d, vals, target, dtype, rows):
    from ofasys_amd import ops
    sel = [(m, is_src, v[rows], "adaptor=text_own_bias" if is_src else a) for m, is_src, v, a in vals]
    slots = make_slots(sel, DEV, dtype)
    logits = model(slots)[0]
    loss = ops.cross_entropy_sum(logits, target[rows].to(DEV), d.pad())
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
    return logits.detach().float(), float(loss), grads


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_custom_adaptor_with_its_own_per_sample_attention_bias(dtype, tol):
    _register()
    case = dict(CASES["tiny_text"])
    case["active"] = set(case["active"]) | {"text_own_bias"}
    model, d = build_model(case, DEV, dtype)
    model.eval()
    vals, target = case_inputs(CASES["tiny_text"])
    assert "text_own_bias" in model.encoder.adaptor.name2adaptor
    assert any(k.startswith("encoder.adaptor.text_own_bias.") for k in model.state_dict())
    both = _step(model, d, vals, target, dtype, slice(0, 2))
    alone = [_step(model, d, vals, target, dtype, slice(b, b + 1)) for b in range(2)]
    # the samples have different lengths: compare each one's non-pad decoder rows
    tgt_len = [int(target[b].ne(d.pad()).sum()) for b in range(2)]
    for b in range(2):
        n = tgt_len[b]
        e = rel_err(both[0][b, :n].cpu(), alone[b][0][0, :n].cpu())
        print(f"MEASURED per-sample bias {dtype}: sample {b} logits vs alone {e:.2e} (bound {tol:.0e})")
        assert e < tol, (b, e)
    assert abs(both[1] - (alone[0][1] + alone[1][1])) <= tol * abs(both[1])
    worst = 0.0
    sums = {k: alone[0][2].get(k, 0) + alone[1][2].get(k, 0) for k in both[2]}
    scale = max(float(r.norm()) for r in sums.values() if torch.is_tensor(r))
    for k, g in both[2].items():
        ref = sums[k]
        if not torch.is_tensor(ref):
            continue
        denom = float(ref.norm())
        if denom > 1e-2 * scale:
            worst = max(worst, float((g - ref).norm()) / denom)
        # (parameters whose gradient is mathematically ~0 -- a key-side position bias shifts every score of a row alike -- hold noise)
        assert float((g - ref).norm()) <= (5 * tol) * denom + 2e-3 * scale, (k, float((g - ref).norm()) / max(denom, 1e-30))
    print(f"MEASURED per-sample bias {dtype}: worst gradient deviation (batch vs sum of singles) {worst:.2e}")
    # the per-sample term is live: its table rows get a gradient through the custom adaptor
    k0 = "encoder.adaptor.text_own_bias.token_rel_pos_table_list.0.weight"
    assert float(both[2][k0][0].abs().sum()) > 0
    # and the bias really differed per sample (otherwise the test proves nothing): the token-equality patterns are not equal
    src = next(v for m, s, v, a in vals if s)
    assert not torch.equal(src[0][:, None] == src[0][None, :], src[1][:, None] == src[1][None, :])
