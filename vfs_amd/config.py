"""Minimal loader for the reference's Python config files (what mmcv.Config.fromfile gives
tools/train.py:70 and tools/test.py:82): the shipped configs are plain dict literals without
`_base_`, so executing the file and wrapping the namespace is sufficient."""
import runpy


class ConfigDict(dict):
    """dict with attribute access (test_cfg.precede_frames, vanilla_tracker.py:133)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @staticmethod
    def fromfile(filename):
        ns = runpy.run_path(filename)
        cfg = Config({k: _wrap(v) for k, v in ns.items() if not k.startswith('__')})
        cfg['filename'] = filename
        return cfg

    def merge_from_dict(self, options):
        """`--options a.b=c` (tools/train.py:49-50)."""
        for full_key, v in options.items():
            d = self
            keys = full_key.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, ConfigDict())
            d[keys[-1]] = _wrap(v)
