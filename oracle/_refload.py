"""Load the real reference implementation (build container only).

``/root/reference/ces/calibrate.py`` mixes tab- and space-indented lines and
raises ``TabError`` under Python 3 (an ordinary Python error).
The loader reads the file as text, rewrites *leading whitespace only* (each
group of four leading spaces becomes one tab) in memory, and executes the
result in a private module.  No token other than indentation changes and
nothing is written to disk; the reference source never enters this repo.

This module is imported only by ``oracle/make_golden.py``.  It needs
``/root/reference`` and therefore never runs on the GPU box.
"""
import os
import re
import types

REFERENCE_ROOT = os.environ.get("CES_REFERENCE_ROOT", "/root/reference")


def _retab(text):
    fixed = []
    for line in text.split("\n"):
        lead = re.match(r"[ \t]*", line).group(0)
        fixed.append(lead.replace("    ", "\t") + line[len(lead):])
    return "\n".join(fixed)


def load_reference_calibrate():
    """Return a module object holding the reference's ``enka`` / ``sampling``."""
    path = os.path.join(REFERENCE_ROOT, "ces", "calibrate.py")
    with open(path) as fh:
        text = fh.read()
    mod = types.ModuleType("reference_ces_calibrate")
    mod.__file__ = path
    exec(compile(_retab(text), path, "exec"), mod.__dict__)
    return mod


def load_reference_utils():
    """``ces/utils.py`` imports unmodified."""
    import importlib.util
    path = os.path.join(REFERENCE_ROOT, "ces", "utils.py")
    spec = importlib.util.spec_from_file_location("reference_ces_utils", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
