"""Edge shapes of the text path (tiny arch, default flags): shared by oracle/gen_edge_golden.py (which runs the REFERENCE on them),
tests/test_oracle_golden.py (oracle vs that fixture, CPU) and tests/test_model_gpu.py (HIP path vs the oracle on the same inputs).
TEST INFRASTRUCTURE.  (B, Ts, Tt, source lengths, target lengths)."""
from oracle import recipe
from oracle.cases import VOCAB_EXTRA, make_target

SHAPES = [
    (1, 1, 1, [1], [1]),                       # one source token, one target position (bos only)
    (1, 33, 65, [33], [65]),                   # batch of one, lengths that are no multiple of any tile
    (3, 5, 2, [5, 1, 3], [2, 1, 2]),           # rows with a single real token next to longer ones
    (2, 129, 31, [129, 2], [31, 30]),          # one row almost entirely padding
    (2, 128, 128, [128, 116], [128, 119]),     # BASELINE.json configs[0]'s own shape: tiny arch, bsz 2, seq_len 128 / 128 (row 1 padded)
]


def key(shape):
    B, Ts, Tt, _, _ = shape
    return f"b{B}_s{Ts}_t{Tt}"


def inputs(shape):
    B, Ts, Tt, slen, tlen = shape
    V = 4 + VOCAB_EXTRA
    src = recipe.tokens(f"input.edge_src{B}{Ts}", (B, Ts), V, slen)
    prev = recipe.tokens(f"input.edge_prev{B}{Tt}", (B, Tt), V, tlen, bos=0)
    return src, prev, make_target(prev)


MAX_POS_STRIDE = 53           # the 1024 x 1024 run keeps every 53rd logit / attention value: the file stays small


def max_position_inputs():
    """1024 source and 1024 target tokens: the edge of max_source_positions / max_target_positions and of the bucket tables."""
    V = 4 + VOCAB_EXTRA
    src = recipe.tokens("input.max_src", (1, 1024), V, [1024])
    prev = recipe.tokens("input.max_prev", (1, 1024), V, [1024], bos=0)
    return src, prev
