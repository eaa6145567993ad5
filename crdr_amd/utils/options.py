"""YAML config with `_base_` inheritance and CLI overlay -- the reference's Config surface
(src/utils/options.py:63-130, 200-352) re-implemented without `addict`.

Semantics kept: `_base_` is a path or list of paths relative to the including file; bases must not share
top-level keys; the child is merged over the union of its bases recursively; a dict carrying `_delete_: true`
replaces instead of merging; command-line values override YAML; `exp` = config file stem.
"""
from __future__ import annotations

import argparse
import copy
import os
import os.path as osp
from typing import Any, Dict, List, Optional, Tuple

import yaml

from .path import PathHandler, check_file_exist

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"
RESERVED_KEYS = ("filename", "text")


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in. Missing keys raise
    (KeyError for [], AttributeError for .) exactly like the reference's ConfigDict; `.get` works as for dict."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{k}'") from None

    def __delattr__(self, k):
        del self[k]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def to_dict(self) -> Dict:
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [plain(x) for x in v]
            return v
        return plain(self)


class BaseConfig:
    @staticmethod
    def _file2dict_yaml(filename: str) -> Tuple[Dict, str, List[str]]:
        filename = osp.abspath(osp.expanduser(filename))
        check_file_exist(filename)
        if osp.splitext(filename)[1] != ".yaml":
            raise IOError("Only yaml type are supported now!")
        with open(filename, encoding="utf-8") as f:
            text = f.read()
        cfg = yaml.safe_load(text) or {}
        cfg_text = filename + "\n" + text
        loaded = [filename]
        if BASE_KEY in cfg:
            bases = cfg.pop(BASE_KEY)
            bases = bases if isinstance(bases, list) else [bases]
            merged_base: Dict[str, Any] = {}
            texts = []
            for rel in bases:
                b_cfg, b_text, b_loaded = BaseConfig._file2dict_yaml(osp.join(osp.dirname(filename), rel))
                clash = merged_base.keys() & b_cfg.keys()
                if clash:
                    raise KeyError(f"Duplicate key is not allowed among bases. Duplicate keys: {clash}")
                merged_base.update(b_cfg)
                texts.append(b_text)
                loaded.extend(b_loaded)
            cfg = BaseConfig._merge_a_into_b(cfg, merged_base)
            cfg_text = "\n".join(texts + [cfg_text])
        return cfg, cfg_text, loaded

    @staticmethod
    def _merge_a_into_b(a: Dict, b: Dict) -> Dict:
        out = dict(b)
        for k, v in a.items():
            if isinstance(v, dict) and k in out and not v.pop(DELETE_KEY, False):
                if not isinstance(out[k], dict):
                    raise TypeError(f"{k}={v} in child config cannot inherit from base because {k} is a dict in the "
                                    f"child config but is of type {type(out[k])} in base config. You may set "
                                    f"`{DELETE_KEY}=True` to ignore the base config")
                out[k] = BaseConfig._merge_a_into_b(v, out[k])
            else:
                out[k] = v
        return out

    def __init__(self, cfg_dict: Optional[Dict] = None, cfg_text: Optional[str] = None, filename: Optional[str] = None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, but got {type(cfg_dict)}")
        for key in cfg_dict:
            if key in RESERVED_KEYS:
                raise KeyError(f"{key} is reserved for config file")
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        if cfg_text is None and filename:
            with open(filename) as f:
                cfg_text = f.read()
        object.__setattr__(self, "_text", cfg_text or "")

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    def __repr__(self):
        return f"Config (path: {self.filename}): {dict.__repr__(self._cfg_dict)}"

    def __len__(self):
        return len(self._cfg_dict)

    def __iter__(self):
        return iter(self._cfg_dict)

    def __contains__(self, k):
        return k in self._cfg_dict

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __getstate__(self):
        return (self._cfg_dict, self._filename, self._text)

    def __setstate__(self, state):
        for attr, val in zip(("_cfg_dict", "_filename", "_text"), state):
            object.__setattr__(self, attr, val)

    def __deepcopy__(self, memo):
        return type(self)(copy.deepcopy(self._cfg_dict.to_dict(), memo), cfg_text=self._text or " ", filename=None)

    def dump(self, filename: str) -> None:
        with open(filename, "w") as f:
            yaml.dump(self._cfg_dict.to_dict(), f)


def _drop_none(d: Dict) -> Dict:
    return {k: v for k, v in d.items() if v is not None}


class TrainConfig(BaseConfig):
    @classmethod
    def get_opt(cls, config_dir: str = "./config", argv: Optional[List[str]] = None) -> "TrainConfig":
        args = cls.arg_parse(argv)
        filename = args["config_path"]
        cfg, text, loaded = cls._file2dict_yaml(filename)
        cfg["loaded_yamls"] = loaded
        opt = cls._merge_a_into_b(args, cfg)
        opt["exp"] = osp.basename(filename).split(".")[0]
        opt["path"] = cls.get_path_dict(opt)
        opt["host"] = os.uname()[1]
        opt["is_train"] = True
        return cls(opt, cfg_text=text, filename=filename)

    @staticmethod
    def arg_parse(argv: Optional[List[str]] = None) -> Dict:
        """Command-line arguments take priority over the YAML files (same flags as the reference)."""
        ap = argparse.ArgumentParser()
        ap.add_argument("config_path", type=str)
        ap.add_argument("-d", "--device", type=str, default="cuda:0")
        ap.add_argument("-si", "--start_iter", type=int)
        ap.add_argument("-e", "--eval_step", type=int)
        ap.add_argument("-l", "--log_step", type=int)
        ap.add_argument("-s", "--save_step", type=int)
        ap.add_argument("-b", "--batch_size", type=int)
        ap.add_argument("-ti", "--total_iter", type=int)
        ap.add_argument("-nw", "--num_workers", type=int, default=8)
        ap.add_argument("-wb", "--use_wandb", action="store_true")
        ap.add_argument("-dr", "--dry_run", action="store_true", help="print config and models and exit")
        ap.add_argument("--debug", action="store_true", help='logger level "DEBUG"')
        ap.add_argument("--wandb_dryrun", action="store_true")
        ns = ap.parse_args(argv)
        if ns.use_wandb and ns.debug:
            raise ValueError("--debug and --use_wandb cannot be turned on at the same time.")
        if ns.dry_run:
            ns.device, ns.debug = "cpu", True
        out = _drop_none(vars(ns))
        if "batch_size" in out:
            out["dataset"] = {"batch_size": out.pop("batch_size")}
        return out

    @staticmethod
    def get_path_dict(cfg: Dict) -> Dict:
        root = cfg["ckpt_root"]
        assert os.path.exists(root), f'checkpoint_root directory "{root}" does not exist.'
        return PathHandler(root, cfg["exp"]).get_exp_path_dict()


class TestConfig(BaseConfig):
    __test__ = False  # not a pytest class

    @classmethod
    def get_opt(cls, config_dir: str = "./config", arg_dict: Optional[Dict] = None) -> "TestConfig":
        args = arg_dict or cls.arg_parse()
        filename = args["config_path"]
        cfg, text, loaded = cls._file2dict_yaml(filename)
        cfg["loaded_yamls"] = loaded
        opt = cls._merge_a_into_b(args, cfg)
        opt["exp"] = osp.basename(filename).split(".")[0]
        opt["path"] = cls.get_path_dict(opt)
        opt["host"] = os.uname()[1]
        opt["is_train"] = False
        opt["dataset"]["test_dataset"] = copy.deepcopy(cls.get_test_dataset_config(opt))
        return cls(opt, cfg_text=text, filename=filename)

    @staticmethod
    def arg_parse(argv: Optional[List[str]] = None) -> Dict:
        ap = argparse.ArgumentParser()
        ap.add_argument("config_path", type=str)
        ap.add_argument("iter", type=int)
        ap.add_argument("test_dataset_name", type=str)
        ap.add_argument("-s", "--sample_size", type=int, default=1000000)
        ap.add_argument("-d", "--device", type=str, default="cuda:0")
        flags = ("notsave", "notgtmask", "debug")
        for f in flags:
            ap.add_argument(f"--{f}", action="store_true")
        out = _drop_none(vars(ap.parse_args(argv)))
        for f in flags:
            if not out[f]:
                del out[f]
        return out

    @staticmethod
    def get_path_dict(cfg: Dict) -> Dict:
        ph = PathHandler(cfg["ckpt_root"], cfg["exp"])
        paths = ph.get_exp_path_dict()
        model_path = ph.get_ckpt_path("comp_model", itr=cfg["iter"])
        assert os.path.exists(model_path), f'model_path "{model_path}" does not exist.'
        sample_dir = osp.join(paths["sample_dir"], f"{cfg['exp']}_iter{cfg['iter'] // 1000}K_{cfg['test_dataset_name']}")
        return dict(ckpt_root=cfg["ckpt_root"], model_dir=paths["model_dir"], sample_dir=sample_dir, model_path=model_path)

    @staticmethod
    def get_test_dataset_config(cfg: Dict) -> Dict:
        ds = copy.deepcopy(cfg["dataset"]["eval_dataset"])
        ds["name"] = cfg["test_dataset_name"]
        if "notgtmask" in cfg:
            ds["use_gt_mask"] = not cfg["notgtmask"]
        return ds
