"""Run the reference's own entry points on the MI355X kernels without editing them.

    PYTHONPATH=<repo>/amphion_amd/integration:<repo>:<Amphion checkout> \
        python <Amphion>/bins/vocoder/inference.py ...

``sitecustomize.py`` in this directory calls ``install()``, which registers a ``sys.meta_path`` finder for
``models.vocoders.vocoder_inference`` and for ``models.codec.codec_inference`` (which carries its own copy of the
same three registries, codec_inference.py:39-75): right after a reference module body has executed, its
``_vocoders`` / ``_vocoder_forward_funcs`` / ``_vocoder_infer_funcs`` entries for the GAN generators built here are
replaced (the dicts are read at call time, vocoder_inference.py:245,346,507).  ``models`` is a regular package, so
it cannot be shadowed -- it is patched.
"""
import importlib.abc
import importlib.util
import sys

TARGETS = ("models.vocoders.vocoder_inference", "models.codec.codec_inference")
TARGET = TARGETS[0]
# generator modules whose CLASSES other models import directly instead of going through the registries:
# models/tts/vits/vits.py:20 (HiFiGAN_vits as Generator), models/tts/jets/jets.py:23,454-458 (HiFiGAN with
# n_mel = attention_dim as the JETS waveform decoder).  Their class attributes are replaced right after the module
# body ran, so a later `from models.vocoders.gan.generator.hifigan import HiFiGAN` binds the MI355X class
# (inference only: these classes have no backward).
CLASS_TARGETS = {
    "models.vocoders.gan.generator.hifigan": ("HiFiGAN", "HiFiGAN_vits"),
    "models.vocoders.gan.generator.bigvgan": ("BigVGAN",),
    "models.vocoders.gan.generator.melgan": ("MelGAN",),
    "models.vocoders.gan.generator.nsfhifigan": ("NSFHiFiGAN",),
    "models.vocoders.gan.generator.apnet": ("APNet",),
}


def _patch_classes(module, fullname):
    import importlib

    ours = importlib.import_module("amphion_amd." + fullname)
    for name in CLASS_TARGETS[fullname]:
        if hasattr(module, name) and hasattr(ours, name):
            setattr(module, "_reference_" + name, getattr(module, name))     # the original stays reachable
            setattr(module, name, getattr(ours, name))
    module.__amphion_amd_patched__ = True


class _PatchLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        if module.__name__ in CLASS_TARGETS:
            _patch_classes(module, module.__name__)
            return
        from amphion_amd.models.vocoders.vocoder_inference import install_into_reference

        install_into_reference(module)
        module.__amphion_amd_patched__ = True
        for other in TARGETS:       # a target imported while this finder was busy resolving another one
            m = sys.modules.get(other)
            if m is not None and hasattr(m, "_vocoders") and not getattr(m, "__amphion_amd_patched__", False):
                install_into_reference(m)
                m.__amphion_amd_patched__ = True


class _Finder(importlib.abc.MetaPathFinder):
    def __init__(self):
        self._pending = set(TARGETS) | set(CLASS_TARGETS)
        self._busy = False

    def find_spec(self, fullname, path, target=None):
        if self._busy or fullname not in self._pending:
            return None
        self._pending.discard(fullname)       # one shot per target
        self._busy = True                     # find_spec below walks sys.meta_path again
        try:
            spec = importlib.util.find_spec(fullname)
        finally:
            self._busy = False
        if not self._pending:
            sys.meta_path.remove(self)
        if spec is None or spec.loader is None:
            return None
        spec.loader = _PatchLoader(spec.loader)
        return spec


def install():
    from_loaded = [t for t in TARGETS if t in sys.modules]
    if from_loaded:
        from amphion_amd.models.vocoders.vocoder_inference import install_into_reference

        for t in from_loaded:
            install_into_reference(sys.modules[t])
            sys.modules[t].__amphion_amd_patched__ = True
    cls_loaded = [t for t in CLASS_TARGETS if t in sys.modules]
    for t in cls_loaded:
        if not getattr(sys.modules[t], "__amphion_amd_patched__", False):
            _patch_classes(sys.modules[t], t)
    if (len(from_loaded) < len(TARGETS) or len(cls_loaded) < len(CLASS_TARGETS)) and not any(isinstance(f, _Finder) for f in sys.meta_path):
        f = _Finder()
        f._pending -= set(from_loaded) | set(cls_loaded)
        sys.meta_path.insert(0, f)
