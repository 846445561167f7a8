"""Import shim: the package directory is named `duckdb-arrow_amd` (hyphen, as the build contract asks), which
Python cannot import by name.  `import duckdb_arrow_amd` loads it from that directory and aliases itself to it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "duckdb-arrow_amd")
_spec = importlib.util.spec_from_file_location("duckdb_arrow_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["duckdb_arrow_amd"] = _mod
_spec.loader.exec_module(_mod)
