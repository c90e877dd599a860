"""`basicsr.models` of the drop-in: only `archs/` lives here.  The `__path__` is extended over a reference checkout later on
`sys.path` (see `basicsr/__init__.py`), so `basicsr.models.<anything else>` still imports from there - but the checkout's own
`basicsr/models/__init__.py` (which imports every `*_model.py` of the training stack, reference basicsr/models/__init__.py:8-18)
is deliberately NOT executed: inference (inference_fdn_lolblur.py) never needs it."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
