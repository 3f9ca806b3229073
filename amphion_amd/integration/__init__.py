"""Run the reference's own entry points on the MI355X kernels without editing them.

    PYTHONPATH=<repo>/amphion_amd/integration:<repo>:<Amphion checkout> \
        python <Amphion>/bins/vocoder/inference.py ...

``sitecustomize.py`` in this directory calls ``install()``, which registers a one-shot
``sys.meta_path`` finder for ``models.vocoders.vocoder_inference``: right after the reference
module body has executed, its ``_vocoders`` / ``_vocoder_forward_funcs`` / ``_vocoder_infer_funcs``
entries for ``hifigan`` and ``bigvgan`` are replaced (the dicts are read at call time,
vocoder_inference.py:245,346,507).  ``models`` is a regular package, so it cannot be shadowed --
it is patched.
"""
import importlib.abc
import importlib.util
import sys

TARGET = "models.vocoders.vocoder_inference"


class _PatchLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        from amphion_amd.models.vocoders.vocoder_inference import install_into_reference

        install_into_reference(module)
        module.__amphion_amd_patched__ = True


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname != TARGET:
            return None
        sys.meta_path.remove(self)  # one shot; avoids recursion in find_spec below
        spec = importlib.util.find_spec(fullname)
        if spec is None or spec.loader is None:
            return None
        spec.loader = _PatchLoader(spec.loader)
        return spec


def install():
    if TARGET in sys.modules:
        from amphion_amd.models.vocoders.vocoder_inference import install_into_reference

        install_into_reference(sys.modules[TARGET])
        return
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
