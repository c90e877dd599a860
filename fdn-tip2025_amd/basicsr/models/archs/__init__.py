"""Architecture registry mirror (reference: basicsr/models/archs/__init__.py:10-46):
modules named *_arch.py in this directory are importable by class name."""
import importlib
import os
from pkgutil import extend_path

# other *_arch.py modules (mar_arch, ...) still import from a reference checkout later on sys.path; FDN_arch, LPNet_arch and
# fdnlol24_arch are found here first
__path__ = extend_path(__path__, __name__)

_arch_dir = os.path.dirname(os.path.abspath(__file__))
_arch_files = sorted(f[:-3] for f in os.listdir(_arch_dir) if f.endswith("_arch.py"))


def dynamic_instantiation(modules, cls_type, opt):
    for m in modules:
        cls_ = getattr(m, cls_type, None)
        if cls_ is not None:
            return cls_(**opt)
    raise ValueError(f"{cls_type} is not found.")


def define_network(opt):
    opt = dict(opt)
    network_type = opt.pop("type")
    mods = [importlib.import_module(f"basicsr.models.archs.{n}") for n in _arch_files]
    return dynamic_instantiation(mods, network_type, opt)
