"""Load the hot-path classes of the reference (sunshinnnn/DSMnet) for pinning.

TEST INFRASTRUCTURE ONLY, and usable only where ``/root/reference`` exists
(the build container).  The GPU box never has the reference, so nothing on the
``-m gpu`` / ``smoke()`` / ``bench.py`` paths may call into this module; it is
used by ``tests/golden/make_goldens.py`` (which writes the committed fixtures)
and by the CPU-only oracle-vs-reference tests, which skip when the tree is
absent.

The reference is Python 2 / PyTorch 0.3 source.  It is read *as text from where
it lies* and executed by this container's torch with the minimal shims recorded
in SURVEY.md section 8c; nothing is copied into the repository.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DSMNET_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "util_conv.py"))


def _exec_text(name, path, stop_before=None):
    """Execute a reference file's text as module ``name`` (optionally truncated)."""
    with open(path, "r") as fh:
        lines = fh.readlines()
    if stop_before is not None:
        lines = lines[:stop_before]
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile("".join(lines), path, "exec"), mod.__dict__)
    return mod


_loaded = {}


def load():
    """Return a dict of reference modules: util_conv, util_fun, submodule,
    stackhourglass, gcnet, dispnetcorr, iresnet."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only
    mdir = os.path.join(REFERENCE_ROOT, "models")
    pdir = os.path.join(mdir, "psmnet")
    # shim 1: util_conv.py has Py2 ``print`` statements in a dead test() from
    # line 275 on; only the text above it parses under Py3.
    with open(os.path.join(mdir, "util_conv.py")) as fh:
        src = fh.readlines()
    cut = next(i for i, l in enumerate(src) if l.startswith("def test("))
    _loaded["util_conv"] = _exec_text("util_conv", os.path.join(mdir, "util_conv.py"), cut)
    # shim 2: implicit relative imports -> put the model dirs on sys.path
    for d in (mdir, pdir):
        if d not in sys.path:
            sys.path.insert(0, d)
    # shim 5: iresnet imports ``util.imwrap`` but the package is ``utils/``
    util_pkg = types.ModuleType("util")
    util_pkg.__path__ = [os.path.join(REFERENCE_ROOT, "utils")]
    sys.modules.setdefault("util", util_pkg)
    import importlib
    for name in ("util_fun", "submodule", "stackhourglass", "gcnet", "dispnetcorr", "iresnet"):
        try:
            _loaded[name] = importlib.import_module(name)
        except Exception as exc:  # keep going: op-level pinning needs util_conv only
            _loaded[name] = exc
    return _loaded


def fix_gcnet(model):
    """Shims 3 and 4 for ``gcnet``: Py2 int division and BatchNorm2d on 5-D."""
    import torch.nn as nn
    model.D = int(model.D)
    for lname in ("l33", "l34", "l35", "l36"):
        seq = getattr(model.layer3d, lname)
        old = seq[1]
        new = nn.BatchNorm3d(old.num_features)
        new.load_state_dict(old.state_dict())
        seq[1] = new
    return model
