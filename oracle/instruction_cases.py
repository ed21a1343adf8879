"""Instruction templates / format calls whose parse is recorded from the reference (oracle/gen_instruction_golden.py) and
replayed against ofasys_amd.Instruction (tests/test_instruction_cpu.py).  TEST INFRASTRUCTURE."""
TEMPLATES = [
    "[IMAGE:image_url] what does the image describe? -> [TEXT:caption]",
    'what is the complete text of " [TEXT:sentence,mask_ratio=0.3] "? -> [TEXT:sentence]',
    "can text1 [TEXT:sent1] imply text2 [TEXT:sent2]? ->  can text1 [TEXT:sent1,no_loss] imply text2 [TEXT:sent2,no_loss]? "
    "[TEXT:label,closed_set]",
    "[IMAGE:img,adaptor=image_patch_embed] which region does the text \" [TEXT:cap] \" describe? region: [BOX:b] -> [BOX:out]",
    " -> [TEXT]",
    "[AUDIO:wav] [VIDEO] x -> y [TEXT:t,max_length=5,noise_ratio=0.2] z",
    "[STRUCT:db,preprocess=table] [MOTION:m] [PHONE:p] [CATEGORY:c] -> [TEXT:sql]",
    "[TEXT:a] and again [TEXT:a] -> [TEXT:b]",
]
SPLITS = [("train", False), ("valid", True)]
# (template index, positional args, keyword args)
FORMATS = [
    (7, ["x"], {"b": "y"}),
    (7, [], {"a": "q", "b": "r", "extra": 5}),
    (7, ["x", "y", "z"], {}),
    (0, [], {"image_url": "IMG", "caption": "a cat", "id": 17}),
    (0, ["IMG"], {}),
    (1, [], {"sentence": "hello world", "label": 1}),
]
BAD = ["no arrow here [TEXT:a]", "a -> b -> c"]


def slot_record(s):
    return [s.modality.name, bool(s.is_src), s.value, s.global_position, s.column_name, s.attributes, bool(s.is_plaintext),
            s.split, bool(s.decoder_plain_with_loss)]
