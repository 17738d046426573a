"""Registry / scan_modules as basicts/data uses them (registry.py:1-3, __init__.py:3-12)."""
import importlib
import os


class Registry:
    def __init__(self, name):
        self.name, self._obj = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(f):
                self._obj[f.__name__] = f
                return f
            return deco
        self._obj[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._obj[name]


def scan_modules(work_dir, search_path, exclude_files=None):
    """import every module next to `search_path` so that decorators register their functions"""
    d = os.path.dirname(os.path.abspath(search_path))
    pkg = os.path.relpath(d, work_dir).replace(os.sep, ".")
    for f in sorted(os.listdir(d)):
        if f.endswith(".py") and f not in (exclude_files or []):
            importlib.import_module(pkg + "." + f[:-3])
