"""Import helper for the *reference* OFASys (test infrastructure, build container only).

TEST INFRASTRUCTURE -- never imported by the product path.  `/root/reference` does not exist
on the GPU box; this module is used only by `oracle/gen_golden.py` to emit the small committed
fixtures under `tests/golden/`.

The reference cannot be imported as-is (oss2, omegaconf, hydra, dacite,
torchvision, ... are absent).  We install a meta-path finder that serves permissive stub modules
for those third-party top-levels, with two real shims the hot path needs:
  * omegaconf.II            (used by ofasys/module/transformer_config.py:38-49)
  * dacite.from_dict        (used by ofasys/configure/configs.py:97-105 to read default_model.yaml)
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = "/root/reference"

_ABSENT = {
    "oss2", "omegaconf", "hydra", "dacite", "torchvision", "torchaudio", "cv2", "timm", "iopath", "ftfy",
    "nltk", "librosa", "soundfile", "av", "diffusers", "clip", "sacrebleu", "jsonlines", "sqlparse",
    "rapidfuzz", "inflect", "pypinyin", "g2p_en", "editdistance", "rouge_score", "matplotlib", "fairscale",
    "apex", "amp_C", "torch_xla", "pytorch_lightning", "taming", "kornia", "sklearn", "tensorboardX",
    "wandb", "common_io", "odps", "pycocoevalcap", "pycocotools", "func_timeout", "zhon", "jieba",
    "bitarray", "sentence_transformers", "fairseq",
}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        m = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == "omegaconf":
            module.II = lambda s: "${" + s + "}"
            module.MISSING = "???"
            module.DictConfig = type("DictConfig", (dict,), {})
            module.OmegaConf = MagicMock()
            module.open_dict = MagicMock()
        elif module.__name__ == "dacite":
            import dataclasses

            def from_dict(data_class, data, config=None):
                obj = data_class()
                for k, v in data.items():
                    cur = getattr(obj, k, None)
                    if dataclasses.is_dataclass(cur) and isinstance(v, dict):
                        for kk, vv in v.items():
                            setattr(cur, kk, vv)
                    else:
                        setattr(obj, k, v)
                return obj

            module.from_dict = from_dict
            module.Config = lambda **kw: None


_installed = False


def install():
    """Make `import ofasys` (the reference) work in this container."""
    global _installed
    if _installed:
        return
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def build_reference_model(arch, vocab_extra, active_adaptors, overrides=None, adaptor_overrides=None, seed=1):
    """Build a reference GeneralistModel with a synthetic Dictionary.

    Follows SURVEY.md section 8c: Dictionary specials bos=0,pad=1,eos=2,unk=3
    (ofasys/preprocessor/dictionary.py:21-53), then `vocab_extra` synthetic symbols.
    """
    install()
    import torch
    import ofasys  # noqa: F401
    from ofasys import GeneralistModel
    from ofasys.preprocessor import Dictionary

    d = Dictionary()
    for i in range(vocab_extra):
        d.add_symbol(f"<text>_{i}")
    torch.manual_seed(seed)
    m = GeneralistModel()
    m.cfg.arch = arch
    m.__init__(m.cfg)
    for k, v in (overrides or {}).items():
        if "." in k:
            a, b = k.split(".")
            setattr(getattr(m.cfg, a), b, v)
        else:
            setattr(m.cfg, k, v)
    # dataclass defaults are shared mutable instances: reset activation explicitly
    import dataclasses
    for f in dataclasses.fields(m.cfg.adaptor):
        if not f.name.startswith("_"):
            getattr(m.cfg.adaptor, f.name).is_active = f.name in active_adaptors
    for name, kv in (adaptor_overrides or {}).items():
        for k, v in kv.items():
            setattr(getattr(m.cfg.adaptor, name), k, v)
    m.initialize(d)
    return m, d


def force_drop_path_draws(model, keep_rows):
    """Replay recorded per-sample keep decisions ([calls, B] of 0/1, tests/golden/<case>.npz "droppath_keep") in the reference's
    active DropPath modules, in call order, instead of fresh torch.rand draws -- so that runs in different dtypes (the bf16 gap
    scripts) take the same stochastic-depth decisions.  The reference divides its input by keep_prob IN PLACE before it multiplies
    by the draw (module/droppath.py:57-59), so the hook only replaces the draw: output = (already divided) input * keep."""
    import torch
    rows = [torch.as_tensor(r) for r in keep_rows]
    state = {"next": 0}

    def hook(m, i, o):
        x = i[0]
        keep = rows[state["next"] % len(rows)].to(x.dtype)
        state["next"] += 1
        return x * keep.view([-1] + [1] * (x.ndim - 1))
    for m in model.modules():
        if type(m).__name__ == "DropPath" and m.drop_prob > 0.0:
            m.register_forward_hook(hook)
