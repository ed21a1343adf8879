"""Minimal counterpart of ofasys.configure for the hot path: the `@register_config(group, name, dataclass)` plugin
registry that adaptors register themselves in (reference: configure/config_store.py:22-72, 207-230).

Only what the path needs is here (no argparse/omegaconf bridge -- out of scope).
"""
import dataclasses
from dataclasses import dataclass, field, is_dataclass
from typing import Any


def ChoiceEnum(choices):
    """Enum class enforcing a list of string choices (reference: configure/constants.py:12-33); members compare equal to
    their string value."""
    import enum

    class _StrEnum(str, enum.Enum):
        def __str__(self):
            return self.value

        def __hash__(self):
            return hash(self.value)

    return _StrEnum("Choices", {k: k for k in choices})


@dataclass
class BaseDataclass:
    _name: Any = None

    @classmethod
    def from_dict(cls, d):
        obj = cls()
        for k, v in (d or {}).items():
            cur = getattr(obj, k, None)
            if is_dataclass(cur) and isinstance(v, dict):
                for kk, vv in v.items():
                    setattr(cur, kk, vv)
            else:
                setattr(obj, k, v)
        return obj

    def update(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        return self


@dataclass
class _Node:
    target: Any
    config: Any


class ConfigStore:
    """Singleton registry: group -> name -> (class, default config dataclass)."""
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
            cls._inst.repo = {}
        return cls._inst

    def store(self, group, name, target, config_cls):
        self.repo.setdefault(group, {})[name] = _Node(target, config_cls)

    def get(self, group, name):
        return self.repo[group][name]

    def names(self, group):
        return list(self.repo.get(group, {}).keys())

    def make_dataclass(self, group, cls_name, module, prefix_names=()):
        """Synthesise a dataclass with one field per registered plugin of `group` (config_store.py:207-230).
        Unlike the reference, every instance gets FRESH sub-configs (default_factory), so adaptor settings do not
        leak between models built in one process (SURVEY.md section 5, config gotcha)."""
        flds = []
        # field order as in the reference (config_store.py:210-218): `prefix_names` first, in that order, then the rest sorted by
        # name.  The general adaptor builds its adaptors in THIS order, which fixes both the state-dict order and the RNG stream
        # of the initial weights (tests/test_init_cpu.py: bit-identical initial state for the same seed).
        prefix_names = list(prefix_names)
        order = lambda kv: (prefix_names.index(kv[0]), "") if kv[0] in prefix_names else (len(prefix_names), kv[0])   # noqa: E731
        for name, node in sorted(self.repo.get(group, {}).items(), key=order):
            flds.append((name, node.config, field(default_factory=node.config)))
        # A plugin registered AFTER this dataclass was synthesised (a user's own adaptor, `@register_config("ofasys.adaptor", name,
        # Cfg)` in their script) still gets its config node: `cfg.adaptor.<name>` creates it on first access, and
        # `late_plugins(cfg)` lists them for the owner (OFAGeneralAdaptor builds them after the built-in ones).
        store = self

        def _late_config(self_, name):
            node = store.repo.get(group, {}).get(name) if not name.startswith("_") else None
            if node is None or node.config is None:
                raise AttributeError(f"{cls_name} has no field {name!r} (no plugin of that name is registered in {group!r})")
            cfg = node.config()
            object.__setattr__(self_, name, cfg)
            return cfg

        dc = dataclasses.make_dataclass(cls_name, flds, bases=(BaseDataclass,), namespace={"__getattr__": _late_config})
        dc.__module__ = module
        dc.__config_group__ = group
        return dc

    def late_plugins(self, cfg):
        """Names registered in cfg's group that are not fields of its (earlier synthesised) dataclass, in registration order."""
        have = {f.name for f in dataclasses.fields(cfg)}
        return [n for n in self.names(getattr(type(cfg), "__config_group__", "")) if n not in have]


def register_config(group, name, dataclass=None):
    """Class decorator: `@register_config("ofasys.adaptor", "text", TextAdaptorConfig)` (config_store.py:41-72)."""
    def deco(cls):
        ConfigStore().store(group, name, cls, dataclass)
        cls.__config_group__, cls.__config_name__ = group, name
        return cls
    return deco
