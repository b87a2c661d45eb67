"""CLI + YAML configuration (mirror of the reference's utils/config_init.py:20-62 and utils/function.py:82-140).

`--key value` pairs are typed (null / int / bool / float / str); values that name a YAML file are loaded;
the template grammar of the third-party `refconfig` package is reproduced as far as the reference's own
configs use it (config/exp/default.yaml, config/model/{naml,nrms}.yaml, config/data/mind.yaml):
    $$import: [relative yaml files]      deep-merged below the importing file
    ${a.b}            string interpolation of a dotted path in the root config
    ${key}$           typed substitution (the whole value)
    ${key:default}$   typed substitution with a default when `key` is absent
refconfig is not vendored anywhere, so this grammar is inferred from usage ("parity unpinned")."""
from __future__ import annotations

import os
import re
import sys
from typing import Any, Dict, List, Optional

import yaml

_REF = re.compile(r"\$\{([^}:]+)(?::([^}]*))?\}(\$?)")


class Obj:
    """attribute access over nested dicts; calling it returns the plain dict (oba.Obj semantics)."""

    def __init__(self, d):
        object.__setattr__(self, "_d", d)

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_d")
        if isinstance(d, dict) and k in d:
            v = d[k]
            return Obj(v) if isinstance(v, (dict, list)) else v
        return None

    def __setattr__(self, k, v):
        object.__getattribute__(self, "_d")[k] = v

    def __getitem__(self, k):
        v = object.__getattribute__(self, "_d")[k]
        return Obj(v) if isinstance(v, (dict, list)) else v

    def __iter__(self):
        for v in object.__getattribute__(self, "_d"):
            yield Obj(v) if isinstance(v, (dict, list)) else v

    def __call__(self):
        return object.__getattribute__(self, "_d")

    def __bool__(self):
        return bool(object.__getattribute__(self, "_d"))


def typed(value: str):
    if not isinstance(value, str):
        return value
    if value == "null":
        return None
    if value.isdigit() or (value.startswith("-") and value[1:].isdigit()):
        return int(value)
    if value.lower() == "true":
        return True
    if value.lower() == "false":
        return False
    try:
        return float(value)
    except ValueError:
        return value


def argparse(arguments: Optional[List[str]] = None) -> Dict[str, Any]:
    arguments = sys.argv[1:] if arguments is None else arguments
    kwargs, key = {}, None
    for arg in arguments:
        if key is not None:
            kwargs[key] = typed(arg)
            key = None
        else:
            assert arg.startswith("--"), f"Unexpected token {arg}, expecting a key starting with '--'."
            key = arg[2:]
    return kwargs


def _merge(base: dict, over: dict) -> dict:
    out = dict(base)
    for k, v in over.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def load_yaml(path: str) -> dict:
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    merged = {}
    for imp in cfg.pop("$$import", []) or []:
        merged = _merge(merged, load_yaml(os.path.join(os.path.dirname(path), imp)))
    return _merge(merged, cfg)


def _lookup(root: dict, dotted: str):
    cur = root
    for part in dotted.strip().split("."):
        if not isinstance(cur, dict) or part not in cur:
            raise KeyError(dotted)
        cur = cur[part]
    return cur


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if not isinstance(node, str):
        return node
    m = _REF.fullmatch(node)
    if m:                                            # the whole value is one reference: typed substitution
        try:
            v = _lookup(root, m.group(1))
        except KeyError:
            if m.group(2) is None:
                raise ValueError(f"config reference ${{{m.group(1)}}} is not defined and has no default")
            v = typed(m.group(2))
        return _resolve(v, root) if isinstance(v, str) and _REF.search(v) else v

    def sub(mm):
        try:
            return str(_lookup(root, mm.group(1)))
        except KeyError:
            if mm.group(2) is None:
                raise ValueError(f"config reference ${{{mm.group(1)}}} is not defined and has no default")
            return mm.group(2)
    return _REF.sub(sub, node)


class CommandInit:
    def __init__(self, required_args, default_args=None):
        self.required_args = required_args
        self.default_args = default_args or {}

    def parse(self, kwargs=None) -> Obj:
        kwargs = dict(kwargs) if kwargs else argparse()
        for arg in self.required_args:
            if arg not in kwargs:
                raise ValueError(f"miss argument {arg}")
        for arg, v in self.default_args.items():
            kwargs.setdefault(arg, v)
        root = {}
        for k, v in kwargs.items():                  # SMART: values naming a yaml file are loaded
            if isinstance(v, str) and v.endswith((".yaml", ".yml")):
                path = v if os.path.exists(v) else os.path.join(os.path.dirname(os.path.abspath(__file__)), v)
                if not os.path.exists(path):
                    raise ValueError(f"config file {v} not found")
                root[k] = load_yaml(path)
            else:
                root[k] = v
        for _ in range(4):                           # references may point at references
            root = _resolve(root, root)
        return Obj(root)


class ModelInit:
    """Name -> local checkpoint of a pretrained language model (mirror of the reference's utils/config_init.py:172-185,
    which keeps `bertbase = bert-base-uncased`-style lines in a `.model` file).  Looked up in `LEGO_MODEL_<NAME>` first,
    then in `./.model`; returns None when the name is unknown (the operator then explains what to configure)."""

    @classmethod
    def examples(cls):
        return "\n".join(["bertbase = bert-base-uncased", "bertlarge = bert-large-uncased"])

    @classmethod
    def get(cls, name: str):
        v = os.environ.get("LEGO_MODEL_" + name.upper())
        if v:
            return v
        if os.path.exists(".model"):
            for line in open(".model"):
                if "=" in line and not line.lstrip().startswith("#"):
                    k, val = line.split("=", 1)
                    if k.strip() == name:
                        return val.strip()
        return None
