"""CPU: the C-ABI library loads and exports every symbol include/flid_tg.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from flid_amd import _lib


def _declared():
    text = open(os.path.join(REPO, "include", "flid_tg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"libflid_tg.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_binding_loads_and_reports_version():
    assert _lib.lib().tg_version() >= 1


def test_argument_errors_are_reported_without_a_gpu():
    l = _lib.lib()
    rc = l.tg_gemm_f32(0, 0, -1, 1, 1, 1.0, None, 1, None, 1, None, 1, None, 0, 0, None)
    assert rc == -1 and b"negative" in l.tg_last_error()
    with pytest.raises(_lib.TgError):
        _lib.check(rc, "tg_gemm_f32")


def test_product_never_imports_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "flid_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                if re.search(r"^\s*(from|import)\s+oracle|oracle/", open(os.path.join(root, f)).read(), flags=re.M):
                    bad.append(f)
    assert not bad, f"product files reference the oracle: {bad}"


def test_missing_library_and_cpu_device_fail_loudly(tmp_path):
    """no CPU fallback: a missing libflid_tg.so raises at first use, and the model classes refuse a CPU device"""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['FLID_TG_LIB'] = %r\n"
            "from flid_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept Exception as e:\n    print('RAISED', type(e).__name__)\n" % (REPO, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED" in out.stdout, out.stdout + out.stderr
    import numpy as np
    from flid_amd.models.TGAT import TGAT
    with pytest.raises(RuntimeError):
        TGAT(np.zeros((3, 4), np.float32), np.zeros((3, 4), np.float32), None, 4, 1, 2, 0.0, device="cpu")


def test_install_aliases_the_reference_import_names():
    """flid_amd.install(): `from models.TGAT import TGAT`, `from utils.utils import get_neighbor_sampler` ... resolve to the mirrors
    (run in a fresh interpreter so that this process's module table stays clean)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import flid_amd; flid_amd.install()\n"
            "from models.TGAT import TGAT\nfrom models.MemoryModel import MemoryModel, compute_src_dst_node_time_shifts\n"
            "from models.DyGFormer import DyGFormer\nfrom models.modules import TimeEncoder, MergeLayer, MultiHeadAttention, MLPClassifier\n"
            "from utils.utils import NeighborSampler, get_neighbor_sampler\n"
            "import flid_amd.models.TGAT as T\nassert TGAT is T.TGAT\n"
            "m = MemoryModel.__init__.__code__.co_varnames\n"
            "assert 'src_node_mean_time_shift' in m and 'model_name' in m\n"
            "print('ALIASED')\n" % REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "ALIASED" in out.stdout, out.stdout + out.stderr
