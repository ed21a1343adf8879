"""Identity of the PRODUCT SOURCES a measurement was taken on: sha256 over the package's Python, the HIP / C++ sources and headers, the
C-ABI header, the library's Makefile and bench.py (sorted relative paths + contents; documents, tests, tools and profiles do not count).
The GPU box has no .git, so this -- not a commit hash -- is what tools/prof_summary.py and tools/pmc_traffic.py stamp into the profile
summaries and what bench.py stamps into its line and compares them with: a summary of other sources is refused.
    python tools/build_id.py        prints the 16-hex-digit id of the tree it lives in"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files(root=ROOT):
    pats = ["ofasys_amd/**/*.py", "ofasys_amd/csrc/*.hip", "ofasys_amd/csrc/*.h", "ofasys_amd/csrc/Makefile", "include/*.h", "bench.py"]
    out = set()
    for p in pats:
        out.update(f for f in glob.glob(os.path.join(root, p), recursive=True) if "/build/" not in f and "__pycache__" not in f)
    return sorted(out)


def source_id(root=ROOT):
    h = hashlib.sha256()
    for f in source_files(root):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_id())
