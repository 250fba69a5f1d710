import ast
import copy

import yaml


class CfgNode(dict):
    """Attribute-access nested dict with the yacs.config.CfgNode methods the reference calls
    (tools/train_net.py:296-299)."""

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("attempted to modify frozen cfg key %s" % name)
        self[name] = CfgNode(value) if isinstance(value, dict) and not isinstance(value, CfgNode) else value

    def _walk(self, fn):
        fn(self)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._walk(fn)

    def freeze(self):
        self._walk(lambda n: object.__setattr__(n, "_frozen", True))

    def defrost(self):
        self._walk(lambda n: object.__setattr__(n, "_frozen", False))

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, "_frozen", self.is_frozen())
        return out

    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return CfgNode(v)
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @staticmethod
    def _like(new, old):
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        return new

    def _merge(self, other):
        for k, v in other.items():
            v = self._decode(v)
            if k in self and isinstance(self[k], CfgNode) and isinstance(v, dict):
                self[k]._merge(v)
            else:
                dict.__setitem__(self, k, self._like(v, self.get(k, v)))

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_list(self, lst):
        if len(lst) % 2:
            raise ValueError("override list must be key value pairs")
        for key, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    dict.__setitem__(node, p, CfgNode())
                node = node[p]
            v = self._decode(v)
            dict.__setitem__(node, parts[-1], self._like(v, node.get(parts[-1], v)))

    def dump(self, **kw):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(plain(self), **kw)
