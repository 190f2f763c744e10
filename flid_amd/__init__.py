"""flid_amd -- MI355X (gfx950) native engine for FLiD's temporal-GNN backbone hot path.

Host side: thin Python over `csrc/libflid_tg.so` (C ABI declared in include/flid_tg.h).  The package mirrors the
reference's own module names for this path (`models.TGAT`, `models.MemoryModel`, `models.DyGFormer`, `models.TCL`,
`models.GraphMixer`, `models.modules`, `utils.utils`) so the reference's trainers can import it unchanged -- see INTEGRATION.md / `flid_amd.install()`.
"""
import sys

__version__ = "0.1.0"


def install(prefix_models: str = "models", prefix_utils: str = "utils"):
    """Alias this package's mirrors under the reference's import names (`from models.TGAT import TGAT`, `from utils.utils import
    get_neighbor_sampler`, ...) so that the reference's trainers run on the HIP engine unchanged.

    Only the backbone modules are replaced: models.TGAT, models.MemoryModel, models.DyGFormer, models.TCL, models.GraphMixer,
    models.modules, utils.utils (PTCL/EM_init.py:1-9 imports exactly these model classes).
    If a package named `models` / `utils` is importable (the reference checkout on sys.path) it is KEPT as the holder, so that
    everything else the trainers import keeps resolving to the reference's own files -- `utils.metrics`, `utils.EarlyStopping`,
    `utils.load_configs`, `models.EdgeBank` -- and those, importing `models.modules` / `utils.utils` in turn, get the mirrors.  Without such a package an
    empty holder is created.  Call before the first `import models...` of the host program."""
    import importlib
    import types

    for pkg, subs in ((prefix_models, ("TGAT", "MemoryModel", "DyGFormer", "TCL", "GraphMixer", "modules")), (prefix_utils, ("utils",))):
        src_pkg = "flid_amd.models" if pkg == prefix_models else "flid_amd.utils"
        holder = sys.modules.get(pkg)
        if holder is None or holder.__name__.startswith("flid_amd"):
            try:
                holder = importlib.import_module(pkg)              # the host program's own package, if there is one
                if getattr(holder, "__name__", "").startswith("flid_amd"):
                    raise ImportError
            except Exception:
                holder = types.ModuleType(pkg)
                holder.__path__ = []
            sys.modules[pkg] = holder
        for s in subs:
            mod = importlib.import_module(f"{src_pkg}.{s}")
            sys.modules[f"{pkg}.{s}"] = mod
            setattr(holder, s, mod)
