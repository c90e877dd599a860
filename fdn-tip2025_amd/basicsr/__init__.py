"""Drop-in `basicsr` namespace for the MI355X-native FDN inference path.

Put `fdn-tip2025_amd/` first on PYTHONPATH and the reference driver's
`from basicsr.models.archs.FDN_arch import *` / `LPNet_arch import *`
(inference_fdn_lolblur.py:3-5) resolves to the HIP-backed modules in this package.
"""
