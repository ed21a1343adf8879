"""The incremental-decoding scenario shared by oracle/gen_incremental_golden.py (reference), the oracle test and the HIP
parity test.  TEST INFRASTRUCTURE."""
from oracle import recipe

STEPS = 8
BEAM_ORDER = [0, 0, 1, 1]          # two beams per source sentence
REORDER_AT = 3                     # after this step ...
NEW_ORDER = [1, 0, 3, 3]           # ... the beams are permuted (within their sentence, as beam search does)


def beam_prefix(vocab):
    """Forced target tokens of the four beams: [4, STEPS], bos first, no padding."""
    return recipe.tokens("input.inc_prev", (len(BEAM_ORDER), STEPS), vocab, None, bos=0)


def padded_prefix(vocab, pad=1):
    """A second scenario: beams 1 and 2 finish early -- their forced tokens are <pad> from step 4 / 6 on, so the cached
    key-padding mask (multihead_attention.py:356-391) takes part; no reorder."""
    t = recipe.tokens("input.inc_prev_pad", (len(BEAM_ORDER), STEPS), vocab, None, bos=0)
    t[1, 4:] = pad
    t[2, 6:] = pad
    return t
