"""Generate tests/golden/bucket_tables.json by CALLING THE REFERENCE's own table builders (adaptor/text.py:20-30 make_token_bucket_position,
adaptor/image_resnet.py:25-40 make_image_bucket_position) at the sizes the adaptors use (text 256/1024, audio 1024/4096, video frames
max_position-sized, a few small ones) and storing the CRC-32 of each int64 table.  Build container only.  TEST INFRASTRUCTURE: data only."""
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_import import install  # noqa: E402


def main():
    install()
    from ofasys.adaptor.text import make_token_bucket_position
    from ofasys.adaptor.image_resnet import make_image_bucket_position
    out = {"token": [], "image": []}
    for bs, mp in [(256, 1024), (1024, 4096), (16, 64), (8, 128), (256, 512), (128, 2048), (64, 1024)]:
        t = make_token_bucket_position(bs, mp)
        out["token"].append({"bucket_size": bs, "max_position": mp, "dtype": str(t.dtype), "crc32": zlib.crc32(t.contiguous().numpy().tobytes())})
    for bs in (42, 7, 16):
        n = (2 * bs - 1) ** 2 + 3
        t = make_image_bucket_position(bs, n)
        out["image"].append({"bucket_size": bs, "num_relative_distance": n, "dtype": str(t.dtype), "crc32": zlib.crc32(t.contiguous().numpy().tobytes())})
    path = os.path.join(ROOT, "tests", "golden", "bucket_tables.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
