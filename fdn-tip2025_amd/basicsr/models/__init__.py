"""`basicsr.models` of the drop-in: only `archs/` lives here.  The `__path__` is extended over a reference checkout later on
`sys.path` (see `basicsr/__init__.py`), so `basicsr.models.<anything else>` still imports from there - but the checkout's own
`basicsr/models/__init__.py` (which imports every `*_model.py` of the training stack, reference basicsr/models/__init__.py:8-18)
is deliberately NOT executed: inference (inference_fdn_lolblur.py) never needs it."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)


def __getattr__(name):
    """`from basicsr.models import create_model` (reference test.py / train.py; basicsr/models/__init__.py:8-40 of the checkout): the checkout's model
    registry is executed ON DEMAND, the first time one of its names is asked for - never by the inference driver.  It imports every `*_model.py` of the
    training stack, so it needs the checkout's own dependencies (cv2, lmdb, ...): a failure there is reported as what it is."""
    if name.startswith("__"):
        raise AttributeError(name)
    import importlib.util
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    reg = sys.modules.get(__name__ + "._reference_registry")
    if reg is None:
        for p in list(__path__):
            init = os.path.join(p, "__init__.py")
            if os.path.abspath(p) == here or not os.path.isfile(init):
                continue
            spec = importlib.util.spec_from_file_location(__name__ + "._reference_registry", init)
            reg = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(reg)
            except Exception as e:
                raise ImportError(f"basicsr.models.{name}: the MI355X drop-in provides only `basicsr.models.archs`; the model registry of the reference "
                                  f"checkout ({init}) was run on demand and failed: {e!r}") from e
            sys.modules[__name__ + "._reference_registry"] = reg
            break
    if reg is not None and hasattr(reg, name):
        return getattr(reg, name)
    raise AttributeError(f"module 'basicsr.models' (MI355X drop-in: only `archs/` lives here) has no attribute {name!r}"
                         + ("" if reg is not None else "; no reference checkout with a basicsr/models/__init__.py follows this package on sys.path"))
