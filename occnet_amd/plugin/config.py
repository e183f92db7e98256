"""Config loader with the mmcv.Config semantics the occ configs rely on ("Config /
flags"): a config file is executed as Python, its top-level non-dunder names become keys; files in
`_base_` are loaded first and recursively dict-merged (child keys win, base-only keys survive,
lists are replaced wholesale); base-file variables are NOT in the child's scope; `_delete_=True`
in a child dict drops the base dict; `--cfg-options a.b=c` style overrides via merge_from_dict.
(reference usage: tools/train.py:105-107, projects/configs/bevformer/bevformer_base_occ.py:1-4)
"""
import ast
import copy
import os
import types

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (mmcv.ConfigDict / addict behaviour used by the configs)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __deepcopy__(self, memo):
        return ConfigDict({copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()})


def _to_configdict(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _to_configdict(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_to_configdict(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_to_configdict(v) for v in obj)
    return obj


def _merge_a_into_b(a, b):
    """Recursive dict merge: keys of `a` (child) override `b` (base)."""
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and isinstance(b[k], dict) and not v.get(DELETE_KEY, False):
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = copy.deepcopy(v)
    return b


def _exec_file(filename):
    with open(filename, encoding='utf-8') as f:
        src = f.read()
    ast.parse(src, filename)  # syntax errors surface with the file name
    mod = types.ModuleType('_occ_cfg_')
    mod.__file__ = filename
    exec(compile(src, filename, 'exec'), mod.__dict__)
    return {k: v for k, v in mod.__dict__.items()
            if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType))}


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if not filename.endswith('.py'):
        raise IOError('Only py type configs are supported')
    cfg = _exec_file(filename)
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, (list, tuple)) else [base]
        merged = {}
        for b in base:
            d = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = merged.keys() & d.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
            merged.update(d)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = cfg_dict or {}
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        super().__setattr__('_cfg_dict', _to_configdict(cfg_dict))
        super().__setattr__('_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _to_configdict(value)

    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """options: {'a.b.c': value} (the --cfg-options form)."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            parts = full_key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        super().__setattr__('_cfg_dict', _to_configdict(_merge_a_into_b(nested, self._cfg_dict)))


def import_plugin(cfg):
    """The reference's plugin-import convention (tools/train.py:114-135): when `plugin` is set,
    import the module named by `plugin_dir` so its registrations run as an import side effect."""
    import importlib
    if not cfg.get('plugin', False):
        return None
    plugin_dir = cfg.get('plugin_dir', None)
    if plugin_dir:
        parts = os.path.dirname(plugin_dir).split('/')
        return importlib.import_module('.'.join(p for p in parts if p))
    return None
