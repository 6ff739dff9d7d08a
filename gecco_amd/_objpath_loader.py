"""Loads the CPython extension with the object-model loops (csrc/objpath.c), building it on first use when a C
compiler is there.  `module()` returns None when it cannot be had: packing.py / crf.py then run their own (slower)
Python statements of the same loops -- host-side convenience only, the device path has no such fallback."""
import importlib.util
import os

_mod = None
_tried = False


def module():
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if os.environ.get("GECCO_AMD_NO_OBJPATH") == "1":  # tests: force the Python statements
        return None
    try:
        from . import build

        path = build.objpath_path()
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(build.OBJPATH_SRC):
            path = build.build_objpath()
        spec = importlib.util.spec_from_file_location("gecco_amd._objpath", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _mod = mod
    except Exception as exc:  # said once: the object API then runs ~7x slower Python statements of the same loops
        import warnings

        warnings.warn(f"gecco_amd: the C extension for the object API could not be built or loaded ({exc!r}); "
                      "using the Python statements of the same loops", RuntimeWarning, stacklevel=2)
        _mod = None
    return _mod
