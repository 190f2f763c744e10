"""CPU: the drop-in boundary -- after flid_amd.install() every name the reference's trainers import from the hot-path modules
resolves (PTCL/EM_init.py:1-9, PTCL/E_step.py:16-31, PTCL/trainer.py:12, PTCL/EM_warmup.py:9-15, models/TCL.py:5-6), the rest of a host
`models` / `utils` package stays reachable, and the restated host glue behaves like the reference's (NegativeEdgeSampler pinned
by a golden the reference produced)."""
import importlib
import os
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture()
def clean_modules():
    saved = {k: v for k, v in sys.modules.items() if k in ("models", "utils") or k.startswith(("models.", "utils."))}
    for k in saved:
        del sys.modules[k]
    path = list(sys.path)
    yield
    for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith(("models.", "utils."))]:
        del sys.modules[k]
    sys.modules.update(saved)
    sys.path[:] = path


def test_install_resolves_every_name_the_trainers_import(clean_modules):
    import flid_amd
    flid_amd.install()
    # PTCL/EM_init.py:1-7 (the hot-path names), PTCL/E_step.py:17-25, PTCL/trainer.py:12, PTCL/EM_warmup.py:9-15, train.py:13
    from models.TGAT import TGAT                                                                      # noqa: F401
    from models.MemoryModel import MemoryModel, compute_src_dst_node_time_shifts                     # noqa: F401
    from models.DyGFormer import DyGFormer                                                            # noqa: F401
    from models.TCL import TCL                                                                        # noqa: F401  (PTCL/EM_init.py:3)
    from models.GraphMixer import GraphMixer                                                          # noqa: F401  (PTCL/EM_init.py:4)
    from models.modules import MergeLayer, MLPClassifier, MLPClassifier_BN                            # noqa: F401
    from models.modules import TimeEncoder, TransformerEncoder, MultiHeadAttention                    # noqa: F401  (models/TCL.py:5)
    from utils.utils import convert_to_gpu                                                            # noqa: F401
    from utils.utils import NegativeEdgeSampler, NeighborSampler                                      # noqa: F401
    from utils.utils import set_random_seed, convert_to_gpu, get_parameter_sizes, create_optimizer    # noqa: F401,F811
    from utils.utils import get_neighbor_sampler, NegativeEdgeSampler                                 # noqa: F401,F811
    import flid_amd.models.TGAT as mine
    assert TGAT is mine.TGAT
    assert TCL.__module__ == "flid_amd.models.TCL" and GraphMixer.__module__ == "flid_amd.models.GraphMixer"
    head = MLPClassifier_BN(input_dim=172, dropout=0.1)
    assert sorted(head.state_dict()) == sorted(["fc1.weight", "fc1.bias", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
                                                "bn1.num_batches_tracked", "fc2.weight", "fc2.bias", "bn2.weight", "bn2.bias",
                                                "bn2.running_mean", "bn2.running_var", "bn2.num_batches_tracked", "fc3.weight", "fc3.bias"])
    enc = TransformerEncoder(attention_dim=8, num_heads=2, dropout=0.0)
    assert {"multi_head_attention.in_proj_weight", "linear_layers.0.weight", "norm_layers.1.bias"} <= set(enc.state_dict())
    assert get_parameter_sizes(head) == sum(p.numel() for p in head.parameters())
    opt = create_optimizer(head, "Adam", 1e-4, 0.0)
    assert isinstance(opt, torch.optim.Adam)
    with pytest.raises(ValueError, match="Wrong value for optimizer"):
        create_optimizer(head, "Adagrad", 1e-4)
    a, b = convert_to_gpu(head, enc, device="cpu")
    assert a is head and b is enc and convert_to_gpu(head, device="cpu") is head
    set_random_seed(5)
    x = np.random.rand()
    set_random_seed(5)
    assert np.random.rand() == x


def test_install_keeps_the_host_packages_other_modules(clean_modules, tmp_path):
    """a host checkout with its own models/ and utils/ packages: install() overrides the backbone modules and leaves the rest
    (models.EdgeBank here, utils.metrics) importable -- and THEIR imports of models.modules / utils.utils get the mirrors"""
    (tmp_path / "models").mkdir()
    (tmp_path / "utils").mkdir()
    (tmp_path / "models" / "__init__.py").write_text("")
    (tmp_path / "utils" / "__init__.py").write_text("")
    (tmp_path / "models" / "EdgeBank.py").write_text(textwrap.dedent("""
        from models.modules import TimeEncoder, TransformerEncoder
        from utils.utils import NeighborSampler
        class EdgeBank:
            parts = (TimeEncoder, TransformerEncoder, NeighborSampler)
    """))
    for name in ("TGAT", "TCL", "GraphMixer"):
        (tmp_path / "models" / f"{name}.py").write_text(f"raise ImportError('the host {name} must have been replaced')\n")
    (tmp_path / "utils" / "metrics.py").write_text("def get_link_prediction_metrics():\n    return 'host'\n")
    sys.path.insert(0, str(tmp_path))
    import flid_amd
    flid_amd.install()
    from models.EdgeBank import EdgeBank
    from models.TGAT import TGAT
    from models.TCL import TCL
    from utils.metrics import get_link_prediction_metrics
    import flid_amd.models.modules as mm
    import flid_amd.utils.utils as uu
    assert EdgeBank.parts == (mm.TimeEncoder, mm.TransformerEncoder, uu.NeighborSampler)
    assert TGAT.__module__ == "flid_amd.models.TGAT" and TCL.__module__ == "flid_amd.models.TCL" and get_link_prediction_metrics() == "host"
    assert os.path.dirname(importlib.import_module("models").__file__) == str(tmp_path / "models")


@pytest.mark.parametrize("strategy", ["random", "historical", "inductive"])
def test_negative_edge_sampler_draws_like_the_reference(strategy):
    from flid_amd.utils.utils import NegativeEdgeSampler
    g = load_golden("neg_sampler")
    src, dst, t = g["src"], g["dst"], g["t"]
    ns = NegativeEdgeSampler(src, dst, interact_times=t, last_observed_time=float(t[len(src) // 2]), negative_sample_strategy=strategy, seed=3)
    for call, (lo, hi) in enumerate(((60, 80), (80, 100), (100, 120))):
        a, b = ns.sample(size=hi - lo, batch_src_node_ids=src[lo:hi], batch_dst_node_ids=dst[lo:hi],
                         current_batch_start_time=float(t[lo]), current_batch_end_time=float(t[hi - 1]))
        assert np.array_equal(a, g[f"{strategy}{call}_s"]) and np.array_equal(b, g[f"{strategy}{call}_d"]), call
    ns.reset_random_state()
    a, b = ns.sample(size=20, batch_src_node_ids=src[60:80], batch_dst_node_ids=dst[60:80], current_batch_start_time=float(t[60]),
                     current_batch_end_time=float(t[79]))
    assert np.array_equal(a, g[f"{strategy}R_s"]) and np.array_equal(b, g[f"{strategy}R_d"])
    with pytest.raises(ValueError, match="Not implemented error for negative_sample_strategy"):
        NegativeEdgeSampler(src, dst, interact_times=t, negative_sample_strategy="nope").sample(3)


def test_time_shifts_match_the_reference():
    """compute_src_dst_node_time_shifts (models/MemoryModel.py:718-751; JODIE's constants, computed and ignored for TGN)"""
    from flid_amd.models.MemoryModel import compute_src_dst_node_time_shifts
    g = load_golden("time_shifts")
    for tag in ("a", "b"):
        got = np.array(compute_src_dst_node_time_shifts(g[f"src_{tag}"], g[f"dst_{tag}"], g[f"t_{tag}"]))
        np.testing.assert_allclose(got, g[f"shifts_{tag}"], rtol=1e-12)
