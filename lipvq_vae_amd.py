"""Import alias: ``import lipvq_vae_amd`` loads the package directory ``lipvq-vae_amd/``
(a hyphen is not legal in a Python module name)."""
import importlib.util as _u
import sys as _s
from pathlib import Path as _P

_dir = _P(__file__).resolve().parent / "lipvq-vae_amd"
_spec = _u.spec_from_file_location("lipvq_vae_amd", _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = _u.module_from_spec(_spec)
_s.modules["lipvq_vae_amd"] = _mod
_spec.loader.exec_module(_mod)
