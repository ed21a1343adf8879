"""Generate tests/golden/tiny_text_incremental.npz by RUNNING THE REFERENCE decoder step by step (build container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_incremental_golden.py
The reference GeneralistModel (tiny, recipe weights, eval mode) encodes the `tiny_text` sources once, the encoder output
is expanded to two beams per sentence (reorder_encoder_out), then the decoder is called with a growing target prefix and
an `incremental_state` (KV cache) exactly as SequenceGenerator does (generator/sequence_generator.py:258-275, 300-306);
after step REORDER_AT the beams are permuted with reorder_incremental_state / reorder_encoder_out.  Stored: per-step
logits, the last step's attention, and the layer-0 self-attention cache as the reference keeps it.  Only data is stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe  # noqa: E402
from oracle.cases import CASES, VOCAB_EXTRA, make_value  # noqa: E402
from oracle.incremental_case import BEAM_ORDER, NEW_ORDER, REORDER_AT, STEPS, beam_prefix, padded_prefix  # noqa: E402
from oracle.ref_import import build_reference_model, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "tiny_text_incremental.npz")


def main():
    install()
    import ofasys  # noqa: F401
    from ofasys import ModalityType
    from ofasys.preprocessor import Slot
    case = CASES["tiny_text"]
    model, d = build_reference_model(case["arch"], VOCAB_EXTRA, case["active"], case["overrides"], case["adaptor_overrides"])
    recipe.fill_state(model.state_dict())
    model.eval()
    V = len(d)
    src = [Slot(ModalityType[m], True, make_value(spec, V), attributes=a) for m, s, spec, a in case["slots"] if s]
    prev = beam_prefix(V)
    with torch.no_grad():
        enc = model.encoder(src)
        enc = model.encoder.reorder_encoder_out(enc, torch.tensor(BEAM_ORDER))
        inc = {}
        logits, attn = [], None
        for t in range(STEPS):
            out, extra = model.decoder([Slot(ModalityType.TEXT, False, prev[:, :t + 1])], encoder_out=enc, incremental_state=inc)
            assert out.shape[1] == 1
            logits.append(out[:, -1].clone())
            attn = extra["attn"][0]
            if t == REORDER_AT:
                order = torch.tensor(NEW_ORDER)
                model.decoder.reorder_incremental_state_scripting(inc, order)
                enc = model.encoder.reorder_encoder_out(enc, order)
                prev = prev.index_select(0, order)
        full, _ = model.decoder([Slot(ModalityType.TEXT, False, prev)], encoder_out=enc)     # teacher-forced, same final beams
        buf = model.decoder.layers[0].self_attn._get_input_buffer(inc)
        # scenario 2: finished beams (pad tokens in the prefix), no reorder
        enc2 = model.encoder.reorder_encoder_out(model.encoder(src), torch.tensor(BEAM_ORDER))
        prev2, inc2, logits2 = padded_prefix(V), {}, []
        for t in range(STEPS):
            out, _ = model.decoder([Slot(ModalityType.TEXT, False, prev2[:, :t + 1])], encoder_out=enc2, incremental_state=inc2)
            logits2.append(out[:, -1].clone())
        buf2 = model.decoder.layers[1].self_attn._get_input_buffer(inc2)
    np.savez_compressed(OUT, logits=torch.stack(logits).numpy(), attn=attn.numpy(), full_last=full[:, -1].numpy(),
                        prev_key_l0=buf["prev_key"].numpy(), prev_value_l0=buf["prev_value"].numpy(),
                        logits_padded=torch.stack(logits2).numpy(),
                        kpm_padded_l1=buf2["prev_key_padding_mask"].float().numpy())
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB; max |incremental - full| at the last step:",
          float((logits[-1] - full[:, -1]).abs().max()))


if __name__ == "__main__":
    main()
