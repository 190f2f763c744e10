"""flid_amd -- MI355X (gfx950) native engine for FLiD's temporal-GNN backbone hot path.

Host side: thin Python over `csrc/libflid_tg.so` (C ABI declared in include/flid_tg.h).  The package mirrors the
reference's own module names for this path (`models.TGAT`, `models.MemoryModel`, `models.DyGFormer`, `models.modules`,
`utils.utils`) so the reference's trainers can import it unchanged -- see INTEGRATION.md / `flid_amd.install()`.
"""
import sys

__version__ = "0.1.0"


def install(prefix_models: str = "models", prefix_utils: str = "utils"):
    """Alias this package's mirrors under the reference's import names (`from models.TGAT import TGAT`, ...)."""
    import importlib
    import types

    for pkg, subs in ((prefix_models, ("TGAT", "MemoryModel", "DyGFormer", "modules")), (prefix_utils, ("utils",))):
        src_pkg = "flid_amd.models" if pkg == prefix_models else "flid_amd.utils"
        holder = sys.modules.get(pkg)
        if holder is None:
            holder = types.ModuleType(pkg)
            holder.__path__ = []
            sys.modules[pkg] = holder
        for s in subs:
            mod = importlib.import_module(f"{src_pkg}.{s}")
            sys.modules[f"{pkg}.{s}"] = mod
            setattr(holder, s, mod)
