"""Golden-case definitions shared by oracle/gen_golden.py (which runs the REFERENCE on them, build container
only) and by the parity tests (which run the oracle restatement and the HIP path on the same recipe inputs).

TEST INFRASTRUCTURE.  A slot spec is (modality, is_src, value-spec, attributes); value-spec is
("tok", key, shape, row_lengths|None) or ("img", key, shape).
"""
import torch

from oracle import recipe

VOCAB_EXTRA = 200
AUDIO_EMBED_DIM = 256          # the tiny arch's embed_dim (the channel-mask case)

CASES = {
    # cfg-1 family: text -> text, tiny, default flags (biased attention everywhere)
    "tiny_text": dict(
        arch="tiny", active={"text"}, overrides={}, adaptor_overrides={},
        slots=[("TEXT", True, ("tok", "src", (2, 16), [16, 11]), None),
               ("TEXT", False, ("tok", "prev", (2, 12), [9, 12]), None)],
        full_grads=["encoder.layers.0.self_attn.c_attn", "encoder.adaptor.text.token_rel_pos_table_list.1.weight",
                    "encoder.adaptor.pos_q_linear.weight", "decoder.cross_pos_k_linear.bias",
                    "decoder.layers.3.ffn_layernorm.weight", "encoder.layers.2.self_attn.q_proj.weight",
                    "decoder.adaptor.text.embed_positions.weight", "encoder.adaptor.embed_tokens.weight"],
    ),
    # the two non-default switches of the adaptor post-hook (adaptor/base.py:168, 174-176): embed_scale = sqrt(D)
    # (no_scale_embedding=False) and scale_embedding_gradient = 0.5 (the gradient into the embedding / type tables is halved)
    "tiny_text_embscale": dict(
        arch="tiny", active={"text"}, overrides={},
        adaptor_overrides={"text": {"no_scale_embedding": False, "scale_embedding_gradient": 0.5}},
        slots=[("TEXT", True, ("tok", "src", (2, 16), [16, 11]), None),
               ("TEXT", False, ("tok", "prev", (2, 12), [9, 12]), None)],
        full_grads=["encoder.adaptor.embed_tokens.weight", "encoder.adaptor.text.type_embedding.weight",
                    "encoder.adaptor.text.layernorm_embedding.weight", "encoder.layers.0.self_attn.q_proj.weight"],
    ),
    # a non-default FFN activation (activation_fn, module/utils.py get_activation_fn): relu
    "tiny_text_relu": dict(
        arch="tiny", active={"text"}, overrides={"activation_fn": "relu"}, adaptor_overrides={},
        slots=[("TEXT", True, ("tok", "src", (2, 16), [16, 11]), None),
               ("TEXT", False, ("tok", "prev", (2, 12), [9, 12]), None)],
        full_grads=["encoder.layers.0.fc1.weight", "decoder.layers.3.ffn_layernorm.weight", "encoder.layers.2.fc2.bias",
                    "encoder.adaptor.embed_tokens.weight"],
    ),
    # the layer's optional normalisations switched the other way: no attn_ln / ffn_layernorm / c_attn (scale_attn, scale_fc,
    # scale_heads = False) and the FFN residual scaled per channel (scale_resids = True, w_resid; transformer_layer.py:60-69, 204-205)
    "tiny_text_noscale": dict(
        arch="tiny", active={"text"}, overrides={"scale_attn": False, "scale_fc": False, "scale_heads": False, "scale_resids": True},
        adaptor_overrides={},
        slots=[("TEXT", True, ("tok", "src", (2, 16), [16, 11]), None),
               ("TEXT", False, ("tok", "prev", (2, 12), [9, 12]), None)],
        full_grads=["encoder.layers.0.w_resid", "decoder.layers.3.w_resid", "encoder.layers.2.fc2.bias",
                    "decoder.layers.1.encoder_attn.out_proj.weight", "encoder.adaptor.embed_tokens.weight"],
    ),
    # post-LN layers (encoder / decoder normalize_before = False: LayerNorm after each residual add, no final stack LayerNorm,
    # transformer_layer.py:183-184, 207-208, 435-436, 468-469, 493-494; model/transformer.py:61-64, 255-258)
    "tiny_text_postln": dict(
        arch="tiny", active={"text"}, overrides={"encoder_normalize_before": False, "decoder_normalize_before": False},
        adaptor_overrides={},
        slots=[("TEXT", True, ("tok", "src", (2, 16), [16, 11]), None),
               ("TEXT", False, ("tok", "prev", (2, 12), [9, 12]), None)],
        full_grads=["encoder.layers.0.self_attn_layer_norm.weight", "decoder.layers.3.final_layer_norm.bias",
                    "decoder.layers.0.encoder_attn_layer_norm.weight", "encoder.layers.3.fc1.weight",
                    "encoder.adaptor.embed_tokens.weight"],
    ),
    # three source slots incl. BOX-as-tokens: slot order != ModalityType order, block-diagonal rel-pos bias
    "tiny_multislot": dict(
        arch="tiny", active={"text"}, overrides={}, adaptor_overrides={},
        slots=[("BOX", True, ("tok", "box", (3, 4), None), None),
               ("TEXT", True, ("tok", "srcA", (3, 7), [7, 5, 3]), None),
               ("STRUCT", True, ("tok", "srcB", (3, 5), None), None),
               ("TEXT", False, ("tok", "prev", (3, 6), [6, 4, 6]), None)],
        full_grads=["encoder.adaptor.text.token_rel_pos_table_list.0.weight", "decoder.layers.0.encoder_attn.c_attn",
                    "encoder.adaptor.text.type_embedding.weight"],
    ),
    # one relative-position table for all layers (share_attn_bias), a non-default attention scale (attn_scale_factor, also the
    # position-bias projections' scaling), and the text adaptor without type embedding / embedding LayerNorm / position LayerNorm
    # (adaptor/base.py:59-66, 141-143, 172-180)
    "tiny_multislot_shared": dict(
        arch="tiny", active={"text"}, overrides={"share_attn_bias": True, "attn_scale_factor": 1.5},
        adaptor_overrides={"text": {"add_type_embedding": False, "layernorm_embedding": False, "layernorm_position": False}},
        slots=[("BOX", True, ("tok", "box", (3, 4), None), None),
               ("TEXT", True, ("tok", "srcA", (3, 7), [7, 5, 3]), None),
               ("STRUCT", True, ("tok", "srcB", (3, 5), None), None),
               ("TEXT", False, ("tok", "prev", (3, 6), [6, 4, 6]), None)],
        full_grads=["encoder.adaptor.text.token_rel_pos_table_list.0.weight", "decoder.adaptor.text.token_rel_pos_table_list.0.weight",
                    "encoder.adaptor.pos_q_linear.weight", "decoder.cross_pos_k_linear.bias",
                    "encoder.adaptor.text.embed_positions.weight"],
    ),
    # modal_ffn (one FFN expert per modality).  The reference's modal_for_ffn ends in x.half() (transformer_layer.py:129), so it
    # only runs in an fp16 model: the fixture is a HALF-precision reference run on CPU ("half": True)
    "tiny_modal_ffn": dict(
        arch="tiny", active={"text"}, overrides={"modal_ffn": True}, adaptor_overrides={}, half=True,
        slots=[("BOX", True, ("tok", "box", (3, 4), None), None),
               ("TEXT", True, ("tok", "srcA", (3, 7), [7, 5, 3]), None),
               ("STRUCT", True, ("tok", "srcB", (3, 5), None), None),
               ("TEXT", False, ("tok", "prev", (3, 6), [6, 4, 6]), None)],
        full_grads=["encoder.layers.0.experts_fc1.0.weight", "encoder.layers.1.experts_fc2.2.weight",
                    "encoder.layers.0.experts_fc1.7.bias", "decoder.layers.0.experts_fc1.0.weight",
                    "decoder.layers.1.experts_fc2.0.bias", "encoder.layers.0.fc1.weight"],
    ),
    # cfg-2 family: image_patch_embed + text -> text, base, the adaptor's only working corner
    "base_patch": dict(
        arch="base", active={"text", "image_patch_embed"},
        overrides={"use_self_attn_bias": False, "entangle_position_embedding": True},
        adaptor_overrides={"text": {"entangle_position_embedding": True},
                           "image_patch_embed": {"entangle_position_embedding": True}},
        slots=[("IMAGE", True, ("img", "image", (2, 3, 224, 224)), ["adaptor=image_patch_embed"]),
               ("TEXT", True, ("tok", "src", (2, 10), [10, 6]), None),
               ("TEXT", False, ("tok", "prev", (2, 8), [8, 5]), None)],
        full_grads=["encoder.adaptor.image_patch_embed.cls_token", "encoder.adaptor.image_patch_embed.proj.bias",
                    "encoder.layers.5.attn_ln.weight", "decoder.layers.0.self_attn.c_attn",
                    "decoder.layers.5.encoder_attn.out_proj.bias", "encoder.adaptor.text.type_embedding.weight"],
    ),
    # cfg-2b family: default IMAGE adaptor (ResNet backbone, BatchNorm in TRAIN mode, 2-D rel-pos bias) + text -> text.
    # dropout = 0 so that train mode is deterministic; resnet50 and a 64x64 image keep the CPU reference run short
    "tiny_resnet": dict(
        arch="tiny", active={"text", "image_resnet"}, overrides={"dropout": 0.0},
        adaptor_overrides={"image_resnet": {"resnet_type": "resnet50"}}, train=True,
        slots=[("IMAGE", True, ("img", "image", (2, 3, 64, 64)), None),
               ("TEXT", True, ("tok", "src", (2, 9), [9, 6]), None),
               ("TEXT", False, ("tok", "prev", (2, 7), [7, 4]), None)],
        full_grads=["encoder.adaptor.image_resnet.embed_images.conv1.weight",
                    "encoder.adaptor.image_resnet.embed_images.layer1.0.downsample.0.weight",
                    "encoder.adaptor.image_resnet.embed_images.layer2.1.conv2.weight",
                    "encoder.adaptor.image_resnet.embed_images.layer3.5.bn3.weight",
                    "encoder.adaptor.image_resnet.embed_images.bn1.bias",
                    "encoder.adaptor.image_resnet.image_proj.weight",
                    "encoder.adaptor.image_resnet.image_rel_pos_table_list.2.weight",
                    "encoder.adaptor.image_resnet.embed_image_positions.weight"],
        block_grads=["layer3.5", "layer3.0", "layer2.0", "layer1.0"],
        buffers=["encoder.adaptor.image_resnet.embed_images.bn1.running_mean",
                 "encoder.adaptor.image_resnet.embed_images.layer3.5.bn3.running_var",
                 "encoder.adaptor.image_resnet.embed_images.layer2.0.downsample.1.running_mean",
                 "encoder.adaptor.image_resnet.embed_images.layer1.2.bn2.num_batches_tracked"],
    ),
    # resnet_drop_path_rate > 0 (adaptor/image_resnet.py:49-52): per-sample stochastic depth of the residual branches, training mode.
    # The keep draws are random inputs: gen_golden.py seeds torch ("drop_seed") and records them per DropPath call
    # ("droppath_keep", [calls, B]); the oracle and the HIP model replay them.  B = 4 so that most draws keep some rows and drop others
    "tiny_resnet_droppath": dict(
        arch="tiny", active={"text", "image_resnet"}, overrides={"dropout": 0.0},
        adaptor_overrides={"image_resnet": {"resnet_type": "resnet50", "resnet_drop_path_rate": 0.3}}, train=True, drop_seed=3,
        slots=[("IMAGE", True, ("img", "image", (4, 3, 64, 64)), None),
               ("TEXT", True, ("tok", "src", (4, 6), [6, 4, 5, 2]), None),
               ("TEXT", False, ("tok", "prev", (4, 5), [5, 3, 4, 5]), None)],
        full_grads=["encoder.adaptor.image_resnet.embed_images.layer2.1.conv2.weight",
                    "encoder.adaptor.image_resnet.embed_images.layer3.5.bn3.weight",
                    "encoder.adaptor.image_resnet.embed_images.layer1.2.bn3.bias",
                    "encoder.adaptor.image_resnet.image_proj.bias"],
        buffers=["encoder.adaptor.image_resnet.embed_images.layer3.5.bn3.running_var"],
    ),
    # cfg-4 family: video clip (3 frames of 64x64, the middle frame of row 1 all-zero = padding) + text -> text
    "tiny_video": dict(
        arch="tiny", active={"text", "video_image_sequence"}, overrides={"dropout": 0.0},
        adaptor_overrides={"image_resnet": {"resnet_type": "resnet50"}}, train=True,
        slots=[("VIDEO", True, ("vid", "video", (2, 3, 3, 64, 64), [(1, 1)]), None),
               ("TEXT", True, ("tok", "src", (2, 5), [5, 3]), None),
               ("TEXT", False, ("tok", "prev", (2, 6), [6, 4]), None)],
        full_grads=["encoder.adaptor.video_image_sequence.embed_frame_positions.weight",
                    "encoder.adaptor.video_image_sequence.video_rel_pos_table_list.1.weight",
                    "encoder.adaptor.image_resnet.image_rel_pos_table_list.0.weight",
                    "encoder.adaptor.image_resnet.image_proj.bias",
                    "encoder.adaptor.video_image_sequence.type_embedding.weight"],
        buffers=["encoder.adaptor.image_resnet.embed_images.layer3.0.bn1.running_mean"],
    ),
    # cfg-5 family: audio fbank [B,T,80] with ragged lengths (row 1 ends up with one padded frame) + mask_emb rows
    "tiny_audio": dict(
        arch="tiny", active={"text", "audio_fbank"}, overrides={"dropout": 0.0}, adaptor_overrides={}, train=False,
        slots=[("AUDIO", True, ("fbank", "audio", (2, 50, 80), [50, 37], [(0, 2), (0, 3), (1, 5)]), ["use_mask"]),
               ("TEXT", True, ("tok", "src", (2, 4), [4, 3]), None),
               ("TEXT", False, ("tok", "prev", (2, 6), [6, 5]), None)],
        full_grads=["encoder.adaptor.audio_fbank.subsample.conv.0.weight", "encoder.adaptor.audio_fbank.subsample.conv.2.bias",
                    "encoder.adaptor.audio_fbank.subsample.out.0.bias", "encoder.adaptor.audio_fbank.mask_emb",
                    "encoder.adaptor.audio_fbank.audio_rel_pos_table_list.1.weight",
                    "encoder.adaptor.audio_fbank.embed_audio_positions.weight"],
    ),
    # audio_fbank.mask_channel_prob > 0 (adaptor/audio.py:462-464): drawn channels zeroed over all frames AFTER the mask_emb rows
    "tiny_audio_chmask": dict(
        arch="tiny", active={"text", "audio_fbank"}, overrides={"dropout": 0.0},
        adaptor_overrides={"audio_fbank": {"mask_channel_prob": 0.1}}, train=False,
        slots=[("AUDIO", True, ("fbank", "audio", (2, 50, 80), [50, 37], [(0, 2), (0, 3), (1, 5)],
                                [(0, 7), (0, 8), (0, 200), (1, 31), (1, 255)]), ["use_mask"]),
               ("TEXT", True, ("tok", "src", (2, 4), [4, 3]), None),
               ("TEXT", False, ("tok", "prev", (2, 6), [6, 5]), None)],
        full_grads=["encoder.adaptor.audio_fbank.subsample.out.0.bias", "encoder.adaptor.audio_fbank.subsample.conv.2.bias",
                    "encoder.adaptor.audio_fbank.mask_emb"],
    ),
    # cfg-5 family: OFA-large (D=1024, 16 heads, 12+12 layers, model/ofa.py:604-610), text + box-as-tokens -> text, short T
    "large_multislot": dict(
        arch="large", active={"text"}, overrides={}, adaptor_overrides={},
        slots=[("TEXT", True, ("tok", "src", (2, 12), [12, 8]), None),
               ("BOX", True, ("tok", "box", (2, 4), None), None),
               ("TEXT", False, ("tok", "prev", (2, 9), [9, 6]), None)],
        full_grads=["encoder.layers.11.self_attn.c_attn", "encoder.adaptor.text.token_rel_pos_table_list.7.weight",
                    "decoder.layers.11.ffn_layernorm.weight", "decoder.cross_pos_q_linear.bias",
                    "decoder.layers.6.encoder_attn.out_proj.bias", "encoder.adaptor.text.type_embedding.weight",
                    "decoder.layer_norm.weight"],
    ),
}


# Cases with a stored golden file that pin the ORACLE only (tests/test_oracle_golden.py, CPU): the HIP path meets the oracle on these
# routes in tests/test_configs_gpu.py (cfg-5: struct / motion micro-batches) rather than through a fixture of its own.
ORACLE_CASES = {
    # every token-carrying modality the general adaptor routes to the text adaptor (adaptor/general.py:36-46): MOTION, PHONE, CATEGORY
    # and STRUCT source slots next to TEXT, with modal_ffn so that each modality's OWN expert runs (half precision as tiny_modal_ffn)
    "tiny_token_modalities": dict(
        arch="tiny", active={"text"}, overrides={"modal_ffn": True}, adaptor_overrides={}, half=True,
        slots=[("MOTION", True, ("tok", "motion", (2, 6), [6, 4]), None),
               ("PHONE", True, ("tok", "phone", (2, 5), None), None),
               ("CATEGORY", True, ("tok", "cat", (2, 2), None), None),
               ("STRUCT", True, ("tok", "struct", (2, 4), [4, 3]), None),
               ("TEXT", True, ("tok", "src", (2, 5), [5, 2]), None),
               ("TEXT", False, ("tok", "prev", (2, 6), [6, 5]), None)],
        full_grads=["encoder.layers.0.experts_fc1.4.weight", "encoder.layers.1.experts_fc2.5.bias",
                    "encoder.layers.0.experts_fc1.8.bias", "encoder.adaptor.text.type_embedding.weight"],
    ),
}


# Cases WITHOUT a stored golden file: the GPU test compares with the oracle directly (tests/test_bench_parity_gpu.py); listed here so
# that oracle/ref_bf16_gap.py can measure the REFERENCE's own bf16-vs-fp32 gap on exactly these inputs (the bf16 tolerance's basis).
EXTRA_CASES = {
    # OFA-large + one IMAGE slot through the default adaptor with the reference's default trunk (resnet152), train-mode BatchNorm
    "large_image": dict(
        arch="large", active={"text", "image_resnet"}, overrides={"dropout": 0.0}, adaptor_overrides={}, train=True,
        slots=[("IMAGE", True, ("img", "large.image", (2, 3, 224, 224)), None),
               ("TEXT", True, ("tok", "large.src", (2, 20), [20, 13]), None),
               ("TEXT", False, ("tok", "prev", (2, 16), [16, 9]), None)],
        full_grads=[],
    ),
}


def make_value(spec, vocab):
    if spec[0] == "tok":
        _, key, shape, lengths = spec
        return recipe.tokens("input." + key, shape, vocab, lengths, bos=0 if key == "prev" else None)
    if spec[0] == "fbank":           # ("fbank", key, [B,T,80], lengths, [(row, subsampled frame) masked][, [(row, channel) zeroed]])
        _, key, shape, lengths, masked = spec[:5]
        v = recipe.floats("input." + key, shape)
        for r, n in enumerate(lengths):
            v[r, n:] = 0.0
        t2 = ((shape[1] - 3) // 2 + 1 - 3) // 2 + 1
        mi = torch.zeros(shape[0], t2, dtype=torch.bool)
        for r, t in masked:
            mi[r, t] = True
        out = {"fbank": v, "fbank_lengths": torch.tensor(lengths, dtype=torch.long), "mask_indices": mi}
        if len(spec) > 5:                        # [B, C] channel mask over the adaptor's embed_dim features (tiny: 256)
            mc = torch.zeros(shape[0], AUDIO_EMBED_DIM, dtype=torch.bool)
            for r, c in spec[5]:
                mc[r, c] = True
            out["mask_channel_indices"] = mc
        return out
    if spec[0] == "vid":                         # ("vid", key, [B,3,F,H,W], [(row, frame) set to exactly zero])
        _, key, shape, zero_frames = spec
        v = recipe.floats("input." + key, shape)
        for r, f in zero_frames:
            v[r, :, f] = 0.0
        return v
    _, key, shape = spec
    return recipe.floats("input." + key, shape)


def make_target(prev, pad=1, eos=2):
    """target = prev shifted left, eos appended at each row's end, pad elsewhere."""
    target = torch.full_like(prev, pad)
    for r in range(prev.shape[0]):
        n = int(prev[r].ne(pad).sum())
        target[r, : n - 1] = prev[r, 1:n]
        target[r, n - 1] = eos
    return target
