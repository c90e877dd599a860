"""Drop-in `basicsr` namespace for the MI355X-native FDN inference path.

Put `fdn-tip2025_amd/` FIRST on PYTHONPATH and a reference checkout after it: the driver's
`from basicsr.models.archs.FDN_arch import *` / `LPNet_arch import *` (inference_fdn_lolblur.py:3-5) resolve to the
HIP-backed modules in this package, and every other `basicsr.*` module the driver needs (`basicsr.utils`,
inference_fdn_lolblur.py:1,6; `basicsr.metrics`, ...) resolves from the checkout: this package, `basicsr.models` and
`basicsr.models.archs` extend their `__path__` over the later `basicsr` directories on `sys.path`
(the reference's `basicsr/` has no `__init__.py` of its own - a namespace portion - which `pkgutil.extend_path` picks up too).
A module that exists in both places is taken from HERE (this package's directories stay first in each `__path__`).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
