"""Host-side mirror of the reference's sampler interface (utils/utils.py:71-302) over the device graph.

`get_neighbor_sampler(data, ...)` / `NeighborSampler` keep the reference's names, arguments, return types (numpy int64 /
int64 / float32 arrays) and error behaviour, so the reference's trainers and its out-of-scope backbones (TCL, GraphMixer)
keep working against it.  The in-scope backbones do not go through these numpy methods: they read `sampler.graph`
(a flid_amd.graph.TemporalGraph) and sample on the device."""
from typing import Optional

import numpy as np
import torch

from ..graph import TemporalGraph


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("flid_amd needs a ROCm device: the sampler runs in libflid_tg.so on the GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class NeighborSampler:
    def __init__(self, graph: TemporalGraph, sample_neighbor_strategy: str = "uniform", time_scaling_factor: float = 0.0,
                 seed: Optional[int] = None):
        self.graph = graph
        self.sample_neighbor_strategy = sample_neighbor_strategy
        self.time_scaling_factor = time_scaling_factor
        self.seed = seed
        self._probs = None
        if self.seed is not None:
            self.random_state = np.random.RandomState(self.seed)          # utils/utils.py:109-110

    # -- reference API ------------------------------------------------------------------------------------------
    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed)             # utils/utils.py:275-280

    def find_neighbors_before(self, node_id: int, interact_time: float, return_sampled_probabilities: bool = False):
        rp, nb, ei, tt = self.graph.host_csr()
        lo, hi = rp[node_id], rp[node_id + 1]                              # IndexError beyond max id, as the reference
        i = int(np.searchsorted(tt[lo:hi], interact_time))
        pr = self._node_probs(node_id)[:i] if return_sampled_probabilities else None
        return nb[lo:lo + i].astype(np.int64), ei[lo:lo + i].astype(np.int64), tt[lo:lo + i], pr

    def get_historical_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray, num_neighbors: int = 20):
        assert num_neighbors > 0, 'Number of sampled neighbors for each node should be greater than 0!'
        node_ids = np.asarray(node_ids)
        if self.sample_neighbor_strategy == "recent":
            if len(node_ids) and (node_ids.max() >= self.graph.num_rows or node_ids.min() < 0):
                raise IndexError("list index out of range")
            dev = _device()
            ids = torch.from_numpy(np.ascontiguousarray(node_ids, dtype=np.int32)).to(dev)
            tt = np.asarray(node_interact_times)
            times = torch.from_numpy(np.ascontiguousarray(tt, dtype=np.float32 if tt.dtype == np.float32 else np.float64)).to(dev)
            nbr, eid, t32, _ = self.graph.sample_recent(ids, times, num_neighbors, want_dt=False)
            return (nbr.cpu().numpy().astype(np.longlong), eid.cpu().numpy().astype(np.longlong), t32.cpu().numpy())
        if self.sample_neighbor_strategy in ("uniform", "time_interval_aware"):
            return self._random_on_host(node_ids, node_interact_times, num_neighbors)
        raise ValueError(f'Not implemented error for sample_neighbor_strategy {self.sample_neighbor_strategy}!')

    def get_all_first_hop_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray):
        a, b, c = [], [], []
        for v, when in zip(node_ids, node_interact_times):
            x, y, z, _ = self.find_neighbors_before(int(v), when)
            a.append(x); b.append(y); c.append(z)
        return a, b, c

    def get_multi_hop_neighbors(self, num_hops: int, node_ids: np.ndarray, node_interact_times: np.ndarray, num_neighbors: int = 20):
        assert num_hops > 0, 'Number of sampled hops should be greater than 0!'
        a, b, c = self.get_historical_neighbors(node_ids, node_interact_times, num_neighbors)
        la, lb, lc = [a], [b], [c]
        for _ in range(1, num_hops):
            a, b, c = self.get_historical_neighbors(la[-1].flatten(), lc[-1].flatten(), num_neighbors)
            la.append(a.reshape(len(node_ids), -1)); lb.append(b.reshape(len(node_ids), -1)); lc.append(c.reshape(len(node_ids), -1))
        return la, lb, lc

    # -- random strategies: bit-exact only by consuming numpy's RandomState stream in node order (utils.py:176-199) ----
    def _node_probs(self, node_id):
        rp, _, _, tt = self.graph.host_csr()
        t = tt[rp[node_id]:rp[node_id + 1]]
        if len(t) == 0:
            return np.array([])
        with np.errstate(divide="ignore", invalid="ignore"):
            ex = np.exp(self.time_scaling_factor * (t - np.max(t)))
            pr = ex / np.cumsum(ex)
        pr[np.isnan(pr)] = -1e10
        return pr

    def _random_on_host(self, node_ids, times, k):
        rp, nb, ei, tt = self.graph.host_csr()
        n = len(node_ids)
        on = np.zeros((n, k)).astype(np.longlong)
        oe = np.zeros((n, k)).astype(np.longlong)
        ot = np.zeros((n, k)).astype(np.float32)
        weighted = self.sample_neighbor_strategy == "time_interval_aware"
        for i, (v, when) in enumerate(zip(node_ids, times)):
            lo, hi = rp[v], rp[v + 1]
            cnt = int(np.searchsorted(tt[lo:hi], when))
            if cnt == 0:
                continue
            p = None
            if weighted:
                p = torch.softmax(torch.from_numpy(self._node_probs(v)[:cnt]).float(), dim=0).numpy()
            rng = np.random if self.seed is None else self.random_state
            pick = rng.choice(a=cnt, size=k, p=p)
            on[i], oe[i], ot[i] = nb[lo + pick], ei[lo + pick], tt[lo + pick]
            pos = ot[i].argsort()
            on[i], oe[i], ot[i] = on[i][pos], oe[i][pos], ot[i][pos]
        return on, oe, ot


def get_neighbor_sampler(data, sample_neighbor_strategy: str = 'uniform', time_scaling_factor: float = 0.0, seed: int = None):
    """mirror of utils/utils.py:283-302.  `data` needs src_node_ids, dst_node_ids, edge_ids, node_interact_times."""
    _device()
    g = TemporalGraph(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    return NeighborSampler(g, sample_neighbor_strategy, time_scaling_factor, seed)
