"""Minimal attribute-style config (the reference uses OmegaConf, which is not a dependency here).
Supports what processors/model_manager.py does with its config: attribute and item access, .get, .copy(),
delattr, ** splatting, nested dicts, and loading the YAML contract of configs/vf_model.yaml."""
from __future__ import annotations

import copy

import yaml


class Config(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Config):
                self[key] = Config(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def copy(self):
        return Config(copy.deepcopy(dict(self)))

    def update(self, other=(), **kw):
        for k, v in dict(other, **kw).items():
            setattr(self, k, v)


def load_yaml(path: str) -> Config:
    with open(path) as f:
        return Config(yaml.safe_load(f))
