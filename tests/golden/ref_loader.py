"""In-memory loader for the Python-2 reference (build container only).

TEST INFRASTRUCTURE.  Used by ``make_golden.py`` (fixture generation) and by
``tests/test_oracle_vs_reference.py`` (skipped when ``/root/reference`` is
absent, i.e. on the GPU box).  Nothing here is imported by the product.

The reference (``/root/reference/*.py``) is Python 2 (print statements,
``xrange``, ``np.int``, ``from IPython import display``).  Following SURVEY.md
section 8(c) the four files on the hot path are read as text, translated with
``lib2to3`` *in memory* and exec'd into fresh module objects.  No reference
source is written anywhere; nothing is fetched.
"""
import os
import sys
import types
import warnings

REFERENCE_DIR = os.environ.get("UAVENV_REFERENCE_DIR", "/root/reference")
_MODULES = ("sinr_visualisation", "channel", "ue_mobility", "mobile_env")
_loaded = None


def reference_available():
    return all(os.path.isfile(os.path.join(REFERENCE_DIR, m + ".py")) for m in _MODULES)


def load_reference():
    """Return dict name -> module for the reference's hot-path files."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference sources not present at %s" % REFERENCE_DIR)
    import numpy as np

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor

        fixers = refactor.get_fixers_from_package("lib2to3.fixes")
        tool = refactor.RefactoringTool(fixers)

    # stubs for what the image lacks / numpy removed
    for name in ("IPython", "IPython.display"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["IPython"].display = sys.modules["IPython.display"]
    if not hasattr(np, "int"):
        np.int = int  # ue_mobility.py:423 uses np.int (removed in numpy >= 1.24)
    import matplotlib

    matplotlib.use("Agg")

    mods = {}
    for name in _MODULES:
        path = os.path.join(REFERENCE_DIR, name + ".py")
        with open(path, "r") as f:
            src = f.read()
        if not src.endswith("\n"):
            src += "\n"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tree = tool.refactor_string(src, name)
        mod = types.ModuleType(name)
        mod.__file__ = path
        sys.modules[name] = mod
        exec(compile(str(tree), path, "exec"), mod.__dict__)
        mods[name] = mod
    _loaded = mods
    return mods
