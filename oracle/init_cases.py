"""Model configurations whose INITIAL weights are pinned against the reference (SURVEY.md section 8a row a17:
module/initialize.py:10-40 init_bert_params, module/layer.py:8-23 Embedding / Linear, multihead_attention.py:93-111,
module/resnet.py:180-185 kaiming / BatchNorm constants, adaptor tables).  Shared by oracle/gen_init_golden.py (runs the REFERENCE,
build container only) and tests/test_init_cpu.py (builds ofasys_amd's model with the same seed).  TEST INFRASTRUCTURE."""

SEED = 1
VOCAB_EXTRA = 200

CASES = {
    # every adaptor of the hot path active on the tiny arch (resnet50 trunk keeps it quick)
    "tiny_all_adaptors": dict(arch="tiny", active=["text", "image_resnet", "video_image_sequence", "audio_fbank"], overrides={},
                              adaptor_overrides={"image_resnet": {"resnet_type": "resnet50"}}),
    # the benchmarked cfg-2 model (bench.build): base, image_patch_embed corner
    "base_patch": dict(arch="base", active=["text", "image_patch_embed"],
                       overrides={"use_self_attn_bias": False, "entangle_position_embedding": True},
                       adaptor_overrides={"text": {"entangle_position_embedding": True},
                                          "image_patch_embed": {"entangle_position_embedding": True}}),
    # the benchmarked cfg-2b model: base + the default image_resnet (resnet101) adaptor, default biased attention
    "base_resnet101": dict(arch="base", active=["text", "image_resnet"], overrides={},
                           adaptor_overrides={"image_resnet": {"resnet_type": "resnet101"}}),
}


def tensor_record(t):
    """What the fixture stores per state-dict entry: shape, moments (fp64) and a digest of the exact bytes."""
    import hashlib
    import torch
    t = t.detach().cpu().contiguous()
    f = t.double() if t.numel() else torch.zeros(1, dtype=torch.float64)
    return {"shape": list(t.shape), "dtype": str(t.dtype).replace("torch.", ""), "mean": float(f.mean()),
            "std": float(f.std(unbiased=False)) if t.numel() > 1 else 0.0, "absmax": float(f.abs().max()),
            "sha1": hashlib.sha1(t.numpy().tobytes()).hexdigest()[:16]}
