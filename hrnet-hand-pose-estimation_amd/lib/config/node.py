"""A small yacs-compatible config node (yacs is not installed in the image).

Supports what the hot path's callers use (reference lib/config/default.py:260-270,
lib/models/pose_hrnet.py:279,292): attribute AND item access, `merge_from_file`
(yaml), `merge_from_list` (KEY.SUB value pairs), `defrost` / `freeze`,
`new_allowed` sub-trees (MODEL.EXTRA), python-literal strings such as
`GPUS: (0,1)`, and the same errors: KeyError for an unknown key, ValueError for
a type mismatch, AttributeError on mutation while frozen.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    _RESERVED = ('_frozen', '_new_allowed')

    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        object.__setattr__(self, '_frozen', False)
        object.__setattr__(self, '_new_allowed', new_allowed)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v, new_allowed=new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # -- attribute access ---------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError('Attempted to set {} to {}, but CfgNode is immutable'.format(name, value))
        self[name] = value

    # -- state ---------------------------------------------------------------
    def _set_frozen(self, flag):
        object.__setattr__(self, '_frozen', flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self._frozen

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode(new_allowed=self._new_allowed)
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    # -- merging -------------------------------------------------------------
    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @staticmethod
    def _coerce(new, old, key):
        if old is None or type(new) is type(old):
            return new
        for a, b in ((list, tuple), (tuple, list)):
            if isinstance(new, a) and isinstance(old, b):
                return b(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, CfgNode) and isinstance(new, dict):
            return new
        raise ValueError('Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}'.format(
            type(old), type(new), old, new, key))

    def _merge(self, other, path):
        for k, v in other.items():
            full = '.'.join(path + [k])
            if k not in self:
                if self._new_allowed:
                    self[k] = CfgNode(v, new_allowed=True) if isinstance(v, dict) else self._decode(v)
                    continue
                raise KeyError('Non-existent config key: {}'.format(full))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError('Type mismatch for config key: {}'.format(full))
                self[k]._merge(v, path + [k])
            else:
                self[k] = self._coerce(self._decode(v), self[k], full)

    def merge_from_other_cfg(self, other):
        if self._frozen:
            raise AttributeError('CfgNode is immutable')
        self._merge(other, [])

    def merge_from_file(self, path):
        with open(path, 'r') as f:
            loaded = yaml.safe_load(f) or {}
        self.merge_from_other_cfg(loaded)

    def merge_from_list(self, opts):
        if self._frozen:
            raise AttributeError('CfgNode is immutable')
        opts = list(opts or [])
        if len(opts) % 2:
            raise AssertionError('Override list has odd length: {}; it must be a list of pairs'.format(opts))
        for full, v in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full.split('.')
            for k in keys[:-1]:
                if k not in node:
                    raise KeyError('Non-existent key: {}'.format(full))
                node = node[k]
            leaf = keys[-1]
            if leaf not in node and not node._new_allowed:
                raise KeyError('Non-existent key: {}'.format(full))
            v = self._decode(v)
            node[leaf] = self._coerce(v, node[leaf], full) if leaf in node else v

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(plain(self))
