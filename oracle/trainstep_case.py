"""The 2-task x 2-micro-batch train step that pins the update arithmetic ("per-task sample_size,
summed grads before/after multiply_grads(world/sum sample_size), grad-norm, clip coefficient, post-Adam params").

TEST INFRASTRUCTURE.  Shared by oracle/gen_trainstep_golden.py (which runs the REFERENCE's own criterion, clip_grad_norm_
and Adam on it, build container only) and by the parity tests (oracle restatement on CPU, Trainer.train_step on the GPU).
"""
from oracle.cases import make_target, make_value

ARCH = "tiny"
ACTIVE = {"text"}
OVERRIDES = {"dropout": 0.0}          # train mode, deterministic
HYPER = dict(lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, clip_norm=0.25)
STEPS = 2

# tasks[t][i] = slot specs of micro-batch i of task t (engine/trainer.py:747-766: `samples[task_id]` is the delayed-update list)
TASKS = [
    [   # task A: text -> text
        [("TEXT", True, ("tok", "ts.a0.src", (2, 14), [14, 9]), None),
         ("TEXT", False, ("tok", "ts.a0.prev", (2, 10), [10, 7]), None)],
        [("TEXT", True, ("tok", "ts.a1.src", (3, 9), [9, 9, 4]), None),
         ("TEXT", False, ("tok", "ts.a1.prev", (3, 8), [5, 8, 6]), None)],
    ],
    [   # task B: box + text + struct -> text (BOX / STRUCT route to the text adaptor, adaptor/general.py:36-46)
        [("BOX", True, ("tok", "ts.b0.box", (2, 4), None), None),
         ("TEXT", True, ("tok", "ts.b0.src", (2, 6), [6, 3]), None),
         ("STRUCT", True, ("tok", "ts.b0.struct", (2, 5), None), None),
         ("TEXT", False, ("tok", "ts.b0.prev", (2, 7), [7, 4]), None)],
        [("BOX", True, ("tok", "ts.b1.box", (2, 4), None), None),
         ("TEXT", True, ("tok", "ts.b1.src", (2, 8), [5, 8]), None),
         ("STRUCT", True, ("tok", "ts.b1.struct", (2, 3), None), None),
         ("TEXT", False, ("tok", "ts.b1.prev", (2, 5), [5, 5]), None)],
    ],
]

# parameters stored in full (gradient before / after the multiply, Adam moments and the parameter after each update)
FULL = [
    "encoder.layers.0.self_attn.c_attn",
    "encoder.adaptor.text.token_rel_pos_table_list.1.weight",
    "encoder.adaptor.pos_q_linear.weight",
    "decoder.cross_pos_k_linear.bias",
    "decoder.layers.3.ffn_layernorm.weight",
    "encoder.layers.2.self_attn.q_proj.weight",
    "decoder.layers.1.fc2.weight",
    "decoder.layers.0.encoder_attn.out_proj.bias",
    "decoder.adaptor.text.embed_positions.weight",
    "encoder.adaptor.embed_tokens.weight",
    "encoder.layer_norm.bias",
]


def micro_batch(specs, vocab):
    """-> ([(modality, is_src, value, attributes)], target) for one micro-batch."""
    vals, prev = [], None
    for mod, is_src, spec, attrs in specs:
        v = make_value(spec, vocab)
        if not is_src:                               # make_value gives bos only to the key "prev"
            v[:, 0] = 0
            prev = v
        vals.append((mod, is_src, v, attrs))
    return vals, make_target(prev)


def sample(t):
    """Big tensors are stored as a strided sample of their flattened form (keeps the fixture small); the tests apply the same."""
    return t.reshape(-1)[::17] if t.numel() > 20000 else t
