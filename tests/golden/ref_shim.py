"""Loader shim that imports the reference's ``trainers`` package from /root/reference
IN THIS CONTAINER ONLY, to generate golden vectors (tests/golden/make_golden.py).

Nothing from the reference is copied: the shim only *points at* /root/reference/src and
patches the interpreter so the py2-era sources import under Python 3 / torch 2.x on CPU:
  * tab -> 8-space expansion at load time (mixed indentation, lsps_trainer.py:58);
  * empty stub modules for cv2 / tensorboardX / torchvision (imports only);
  * identity ``.cuda()`` on tensors and modules (the reference hard-calls .cuda(gpu));
  * ``Tensor.get_device`` tolerant on CPU.
It refuses to run when /root/reference is absent (e.g. on the GPU box).
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

REF_SRC = os.environ.get("LSPS_REFERENCE_SRC", "/root/reference/src")


class _TabExpandLoader(importlib.machinery.SourceFileLoader):
    def get_data(self, path):
        data = super().get_data(path)
        if path.endswith(".py"):
            data = data.decode("utf-8").expandtabs(8).encode("utf-8")
        return data


class _Finder(importlib.machinery.PathFinder):
    @classmethod
    def find_spec(cls, fullname, path=None, target=None):
        spec = importlib.machinery.PathFinder.find_spec(fullname, path, target)
        if spec is None or not spec.origin or not spec.origin.startswith(REF_SRC):
            return None
        if spec.origin.endswith(".py"):
            spec.loader = _TabExpandLoader(spec.loader.name, spec.loader.path)
        return spec


def load_reference_trainers():
    """Returns the reference ``trainers`` package (module object)."""
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference sources not present at %s" % REF_SRC)
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True
    for name in ("cv2", "tensorboardX", "torchvision"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    # identity .cuda(): the reference calls .cuda(gpu) unconditionally
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: 0
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    if not any(isinstance(f, type) and f is _Finder for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder)
    for m in [k for k in sys.modules if k == "trainers" or k.startswith("trainers.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    import trainers  # noqa: E402  (the reference package)
    return trainers


class _DataLoader(_TabExpandLoader):
    """As above, plus the single python-2 print statement of utils/handdetector.py (line 214) made a call —
    in memory, at load time; nothing of the reference is written anywhere."""

    def get_data(self, path):
        data = super().get_data(path)
        if path.endswith("handdetector.py"):
            import re
            data = re.sub(rb'^(\s*)print (".*")\s*$', rb'\1print(\2)', data, flags=re.M)
        return data


def load_reference_data(cv2_standin):
    """Returns (dataset_hand2 module, HandDetector class, NYUImporter class) of the REAL reference, imported with
    stubs for what the image lacks: `cv2` = `cv2_standin` (see oracle/data_ref.py's header for what that does and
    does not pin), `progressbar`, `cPickle`; `xrange` -> range."""
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference sources not present at %s" % REF_SRC)
    import builtins
    import pickle
    sys.dont_write_bytecode = True
    builtins.xrange = range
    sys.modules["cv2"] = cv2_standin
    sys.modules.setdefault("progressbar", types.ModuleType("progressbar"))
    sys.modules.setdefault("cPickle", pickle)
    sys.modules.setdefault("_pickle", pickle)
    import matplotlib
    matplotlib.use("Agg")

    class _F(importlib.machinery.PathFinder):
        @classmethod
        def find_spec(cls, fullname, path=None, target=None):
            spec = importlib.machinery.PathFinder.find_spec(fullname, path, target)
            if spec is None or not spec.origin or not spec.origin.startswith(REF_SRC):
                return None
            if spec.origin.endswith(".py"):
                spec.loader = _DataLoader(spec.loader.name, spec.loader.path)
            return spec

    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    sys.meta_path.insert(0, _F)
    for m in [k for k in sys.modules if k.split(".")[0] in ("data", "utils")]:
        del sys.modules[m]
    import data  # noqa: E402,F401  (package first: utils.handdetector <-> data.importers import each other)
    from data import dataset_hand2  # noqa: E402
    from data.importers import NYUImporter  # noqa: E402
    from utils.handdetector import HandDetector  # noqa: E402
    return dataset_hand2, HandDetector, NYUImporter
