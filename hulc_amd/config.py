"""Mini Hydra/OmegaConf-compatible config loader over PyYAML (hydra / omegaconf are not installed on either box).

Covers exactly the features the reference's `conf/` tree uses (SURVEY.md §5 "Config / flags"):
defaults lists with nested groups (`conf/config.yaml:1-11`, `conf/model/perceptual_encoder/gripper_cam.yaml:4-10`),
package relocation `/group@pkg: option` (`conf/model/hulc.yaml:14`), `override` / `_self_` entries, `# @package _group_`
headers, `${a.b}` interpolation and the `${now:...}` resolver, `???` / `??` mandatory markers, `_target_` +
`_recursive_: false` instantiation, and CLI overrides `group=option`, `key.sub=value`, `+key=value`, `~group`.

`instantiate` maps the reference's dotted `_target_` paths onto this package (TARGET_MAP) so the reference's own YAML files
work unchanged: `hulc.models.hulc.Hulc` -> `hulc_amd.hulc.Hulc`.
"""
from __future__ import annotations

import copy
import datetime
import importlib
import os
import re
from typing import Any, Dict, List, Optional

import yaml

TARGET_MAP = {
    "hulc.models.hulc.Hulc": "hulc_amd.hulc.Hulc",
    "hulc.models.gcbc.GCBC": "hulc_amd.hulc.GCBC",
    "torch.optim.Adam": "hulc_amd.hulc.FusedAdam",
    "hulc.models.encoders.language_network.SBert": "hulc_amd.sbert.SBert",
}
MISSING = "???"


class Cfg(dict):
    """dict with attribute access (what the reference code does with DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _load_yaml(path: str):
    with open(path) as f:
        txt = f.read()
    pkg_group = bool(re.match(r"\s*#\s*@package\s+_group_", txt))
    return (yaml.safe_load(txt) or {}), pkg_group


def _set_path(root: dict, path: List[str], value, merge=True):
    d = root
    for k in path[:-1]:
        d = d.setdefault(k, {})
    if not path:
        _merge(root, value)
        return
    if merge and isinstance(value, dict) and isinstance(d.get(path[-1]), dict):
        _merge(d[path[-1]], value)
    else:
        d[path[-1]] = value


def _merge(dst: dict, src: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def _compose_file(conf_dir: str, rel: str, pkg: List[str], choices: Dict[str, str], deleted: set, out: dict):
    """Compose the yaml at conf_dir/rel(.yaml) into `out` under package path `pkg`."""
    path = os.path.join(conf_dir, rel + ".yaml")
    if not os.path.exists(path):
        raise FileNotFoundError(f"config not found: {path}")
    body, _ = _load_yaml(path)
    defaults = body.pop("defaults", None) or []
    group_dir = os.path.dirname(rel)
    self_done = False

    def apply_self():
        _set_path(out, pkg, copy.deepcopy(body))

    for item in defaults:
        if item == "_self_":
            apply_self()
            self_done = True
            continue
        if isinstance(item, str):
            _compose_file(conf_dir, os.path.join(group_dir, item), pkg, choices, deleted, out)
            continue
        (key, option), = item.items()
        key = key.strip()
        if key.startswith("override "):
            key = key[len("override "):].strip()
        if key.startswith("optional "):
            key = key[len("optional "):].strip()
        if key.startswith("hydra/"):
            continue                                # hydra's own logging groups: nothing to compose
        target_pkg = None
        if "@" in key:
            key, target_pkg = key.split("@", 1)
        absolute = key.startswith("/")
        grp = key.lstrip("/")
        grp_rel = grp if absolute else os.path.join(group_dir, grp)
        choice_key = grp_rel.replace(os.sep, "/")
        if choice_key in deleted:
            continue
        option = choices.get(choice_key, option)
        if option is None or option == "null":
            continue
        if target_pkg is not None:
            sub_pkg = (pkg if not absolute else []) + [p for p in target_pkg.split(".") if p and p != "_global_"]
            if absolute:
                sub_pkg = pkg + [p for p in target_pkg.split(".") if p]
        else:
            sub_pkg = ([] if absolute else pkg) + grp.split("/")
            if absolute:
                sub_pkg = grp.split("/")
        _compose_file(conf_dir, os.path.join(grp_rel, str(option)), sub_pkg, choices, deleted, out)
    if not self_done:
        apply_self()


_INTERP = re.compile(r"\$\{([^{}]+)\}")


def _resolve(root: dict):
    now = datetime.datetime.now()

    def lookup(path: str):
        if path.startswith("now:"):
            return now.strftime(path[4:])
        d: Any = root
        for k in path.split("."):
            if isinstance(d, dict) and k in d:
                d = d[k]
            else:
                raise KeyError(f"interpolation key '{path}' not found")
        return res(d)

    def res(v):
        if isinstance(v, str):
            try:
                m = _INTERP.fullmatch(v)
                if m:
                    return copy.deepcopy(lookup(m.group(1)))
                return _INTERP.sub(lambda mm: str(lookup(mm.group(1))), v)
            except KeyError:
                return v        # OmegaConf resolves lazily: a dangling ${...} only fails when accessed
        if isinstance(v, dict):
            for k in list(v.keys()):
                if k == "hydra":
                    continue
                v[k] = res(v[k])
            return v
        if isinstance(v, list):
            return [res(x) for x in v]
        return v

    return res(root)


def _parse_value(s: str):
    try:
        return yaml.safe_load(s)
    except Exception:
        return s


def compose(conf_dir: str, config_name: str = "config", overrides: Optional[List[str]] = None, resolve: bool = True) -> Cfg:
    overrides = overrides or []
    choices: Dict[str, str] = {}
    deleted: set = set()
    assigns: List[tuple] = []
    for ov in overrides:
        if ov.startswith("~"):
            deleted.add(ov[1:].split("=")[0])
            continue
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        grp_dir = os.path.join(conf_dir, k.replace(".", "/"))
        if os.path.isdir(grp_dir) and os.path.exists(os.path.join(grp_dir, v + ".yaml")):
            choices[k.replace(".", "/")] = v
        else:
            assigns.append((k, _parse_value(v)))
    out: dict = {}
    _compose_file(conf_dir, config_name, [], choices, deleted, out)
    for k, v in assigns:
        _set_path(out, k.split("."), v, merge=False)
    out.pop("hydra", None)
    if resolve:
        _resolve(out)
    return _wrap(out)


def missing_keys(cfg, prefix="") -> List[str]:
    out = []
    if isinstance(cfg, dict):
        for k, v in cfg.items():
            out += missing_keys(v, f"{prefix}{k}.")
    elif cfg in ("???", "??"):
        out.append(prefix[:-1])
    return out


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate with `_recursive_: false` semantics (nested configs are passed through as Cfg)."""
    if cfg is None or (isinstance(cfg, dict) and len(cfg) == 0):
        return None
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    target = TARGET_MAP.get(target, target)
    mod, _, name = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return fn(*args, **cfg)
