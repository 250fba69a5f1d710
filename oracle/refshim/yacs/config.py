"""Minimal stand-in for yacs.config.CfgNode (test infrastructure only).

Used ONLY by oracle/refimport.py to import the read-only reference in this
container; yacs is not installed here.  Contains no reference code.
"""
import ast
import copy
import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in init_dict.items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError("immutable CfgNode: %s" % name)
        self[name] = value

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return out

    @staticmethod
    def _coerce(new, old):
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        return new

    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return CfgNode(v)
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    def _merge(self, other):
        for k, v in other.items():
            v = self._decode(v)
            if k in self and isinstance(self[k], CfgNode) and isinstance(v, dict):
                self[k]._merge(v)
            elif k in self:
                dict.__setitem__(self, k, self._coerce(v, self[k]))
            else:
                dict.__setitem__(self, k, v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for key, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            v = self._decode(v)
            dict.__setitem__(node, parts[-1], self._coerce(v, node.get(parts[-1], v)))

    def dump(self, **kw):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self), **kw)
