"""Build the ofasys_amd model for a golden case and fill it from the shared weight recipe."""
import torch

from oracle import recipe
from oracle.cases import VOCAB_EXTRA


def build_model(case, device=None, dtype=torch.float32):
    from ofasys_amd import Dictionary, GeneralistModel
    d = Dictionary()
    for i in range(VOCAB_EXTRA):
        d.add_symbol(f"<text>_{i}")
    m = GeneralistModel()
    m.cfg.arch = case["arch"]
    m.__init__(m.cfg)
    for k, v in case["overrides"].items():
        setattr(m.cfg, k, v)
    for a in case["active"]:
        getattr(m.cfg.adaptor, a).is_active = True
    for a, kv in case["adaptor_overrides"].items():
        for k, v in kv.items():
            setattr(getattr(m.cfg.adaptor, a), k, v)
    m.initialize(d)
    recipe.fill_state(m.state_dict())
    if device is not None:
        m = m.to(device)
    if dtype != torch.float32:
        m = m.to(dtype)
    return m, d


def make_slots(vals, device, float_dtype=torch.float32):
    from ofasys_amd import ModalityType, Slot
    out = []
    for mod, is_src, v, attrs in vals:
        if isinstance(v, dict):                      # audio: {"fbank", "fbank_lengths", "mask_indices"}
            v = {k: (t.to(device).to(float_dtype) if t.is_floating_point() else t.to(device)) for k, t in v.items()}
            out.append(Slot(ModalityType[mod], is_src, v, attributes=attrs))
            continue
        v = v.to(device)
        if v.is_floating_point():
            v = v.to(float_dtype)
        out.append(Slot(ModalityType[mod], is_src, v, attributes=attrs))
    return out
