"""Host-side mirror of the reference's sampler interface (utils/utils.py:71-302) over the device graph.

`get_neighbor_sampler(data, ...)` / `NeighborSampler` keep the reference's names, arguments, return types (numpy int64 /
int64 / float32 arrays) and error behaviour, so the reference's trainers and its out-of-scope backbones (TCL, GraphMixer)
keep working against it.  The in-scope backbones do not go through these numpy methods: they read `sampler.graph`
(a flid_amd.graph.TemporalGraph) and sample on the device."""
import random
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from ..graph import TemporalGraph


# ---- host glue the reference's trainers import from utils.utils (PTCL/trainer.py:12, PTCL/E_step.py:24-25, train.py:13) ----------
def set_random_seed(seed: int = 0):
    """seed python / numpy / torch (reference utils/utils.py:9-21)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def convert_to_gpu(*data, device: str):
    """`.to(device)` on every argument; one argument comes back bare, several as a tuple (reference utils/utils.py:24-38)"""
    moved = tuple(item.to(device) for item in data)
    return moved if len(moved) > 1 else moved[0]


def get_parameter_sizes(model: nn.Module):
    """number of trainable scalars (reference utils/utils.py:41-47)"""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def create_optimizer(model: nn.Module, optimizer_name: str, learning_rate: float, weight_decay: float = 0.0):
    """Adam / SGD / RMSprop over model.parameters() (reference utils/utils.py:50-68)"""
    table = {"Adam": torch.optim.Adam, "SGD": torch.optim.SGD, "RMSprop": torch.optim.RMSprop}
    if optimizer_name not in table:
        raise ValueError(f"Wrong value for optimizer {optimizer_name}!")
    return table[optimizer_name](params=model.parameters(), lr=learning_rate, weight_decay=weight_decay)


class NegativeEdgeSampler(object):
    """Negative destination sampling for the link-prediction warm-up (reference utils/utils.py:305-495): strategies "random"
    (the trainers' choice, PTCL/EM_warmup.py:78-83), "historical" and "inductive".  Same constructor, attributes, method names and
    numpy RandomState consumption order as the reference, so a seeded run draws the same negatives."""

    def __init__(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, interact_times: np.ndarray = None, last_observed_time: float = None,
                 negative_sample_strategy: str = 'random', seed: int = None):
        self.seed = seed
        self.negative_sample_strategy = negative_sample_strategy
        self.src_node_ids, self.dst_node_ids, self.interact_times = src_node_ids, dst_node_ids, interact_times
        self.unique_src_node_ids = np.unique(src_node_ids)
        self.unique_dst_node_ids = np.unique(dst_node_ids)
        self.unique_interact_times = np.unique(interact_times)
        self.earliest_time = min(self.unique_interact_times)
        self.last_observed_time = last_observed_time
        if negative_sample_strategy != 'random':
            self.possible_edges = set((s, d) for s in self.unique_src_node_ids for d in self.unique_dst_node_ids)
        if negative_sample_strategy == 'inductive':
            self.observed_edges = self.get_unique_edges_between_start_end_time(self.earliest_time, self.last_observed_time)
        if seed is not None:
            self.random_state = np.random.RandomState(seed)

    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed)

    def get_unique_edges_between_start_end_time(self, start_time: float, end_time: float):
        inside = np.logical_and(self.interact_times >= start_time, self.interact_times <= end_time)
        return set(zip(self.src_node_ids[inside], self.dst_node_ids[inside]))

    def sample(self, size: int, batch_src_node_ids: np.ndarray = None, batch_dst_node_ids: np.ndarray = None,
               current_batch_start_time: float = 0.0, current_batch_end_time: float = 0.0):
        if self.negative_sample_strategy == 'random':
            return self.random_sample(size=size)
        if self.negative_sample_strategy == 'historical':
            return self.historical_sample(size, batch_src_node_ids, batch_dst_node_ids, current_batch_start_time, current_batch_end_time)
        if self.negative_sample_strategy == 'inductive':
            return self.inductive_sample(size, batch_src_node_ids, batch_dst_node_ids, current_batch_start_time, current_batch_end_time)
        raise ValueError(f'Not implemented error for negative_sample_strategy {self.negative_sample_strategy}!')

    def random_sample(self, size: int):
        rng = np.random if self.seed is None else self.random_state
        si = rng.randint(0, len(self.unique_src_node_ids), size)          # source indices are drawn first, then destinations
        di = rng.randint(0, len(self.unique_dst_node_ids), size)
        return self.unique_src_node_ids[si], self.unique_dst_node_ids[di]

    def random_sample_with_collision_check(self, size: int, batch_src_node_ids: np.ndarray, batch_dst_node_ids: np.ndarray):
        assert batch_src_node_ids is not None and batch_dst_node_ids is not None
        free = list(self.possible_edges - set(zip(batch_src_node_ids, batch_dst_node_ids)))
        assert len(free) > 0
        pick = self.random_state.choice(len(free), size=size, replace=len(free) < size)
        return np.array([free[i][0] for i in pick]), np.array([free[i][1] for i in pick])

    def _from_pool(self, pool, size, batch_src_node_ids, batch_dst_node_ids):
        """`size` negatives from the edge set `pool`, topped up with collision-checked random edges when the pool is too small
        (the shared tail of the historical and inductive strategies)"""
        ps = np.array([e[0] for e in pool])
        pd = np.array([e[1] for e in pool])
        if size > len(pool):
            rs, rd = self.random_sample_with_collision_check(size - len(pool), batch_src_node_ids, batch_dst_node_ids)
            ns, nd = np.concatenate([rs, ps]), np.concatenate([rd, pd])
        else:
            pick = self.random_state.choice(len(pool), size=size, replace=False)
            ns, nd = ps[pick], pd[pick]
        return ns.astype(np.longlong), nd.astype(np.longlong)       # (an empty operand makes concatenate return floats)

    def historical_sample(self, size: int, batch_src_node_ids: np.ndarray, batch_dst_node_ids: np.ndarray,
                          current_batch_start_time: float, current_batch_end_time: float):
        assert self.seed is not None
        past = self.get_unique_edges_between_start_end_time(self.earliest_time, current_batch_start_time)
        now = self.get_unique_edges_between_start_end_time(current_batch_start_time, current_batch_end_time)
        return self._from_pool(past - now, size, batch_src_node_ids, batch_dst_node_ids)

    def inductive_sample(self, size: int, batch_src_node_ids: np.ndarray, batch_dst_node_ids: np.ndarray,
                         current_batch_start_time: float, current_batch_end_time: float):
        assert self.seed is not None
        past = self.get_unique_edges_between_start_end_time(self.earliest_time, current_batch_start_time)
        now = self.get_unique_edges_between_start_end_time(current_batch_start_time, current_batch_end_time)
        return self._from_pool(past - self.observed_edges - now, size, batch_src_node_ids, batch_dst_node_ids)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("flid_amd needs a ROCm device: the sampler runs in libflid_tg.so on the GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class NeighborSampler:
    def __init__(self, graph: TemporalGraph, sample_neighbor_strategy: str = "uniform", time_scaling_factor: float = 0.0,
                 seed: Optional[int] = None, device_random: bool = False):
        """device_random (not in the reference, opt-in): the `uniform` / `time_interval_aware` strategies draw on the DEVICE from a
        counter-based generator (tg_sample_random) instead of numpy's RandomState on the host -- the same distributions, not the same
        stream: no bit-exactness with the reference, no per-node host loop (the host path costs ~20 us per query)."""
        self.graph = graph
        self.sample_neighbor_strategy = sample_neighbor_strategy
        self.time_scaling_factor = time_scaling_factor
        self.seed = seed
        self._probs = None
        self.device_random = bool(device_random) and sample_neighbor_strategy in ("uniform", "time_interval_aware")
        self._draws = 0                                                   # calls since the last reset: part of the device generator's key
        if self.device_random and sample_neighbor_strategy == "time_interval_aware":
            graph.set_time_weights(time_scaling_factor)
        if self.seed is not None:
            self.random_state = np.random.RandomState(self.seed)          # utils/utils.py:109-110

    # -- reference API ------------------------------------------------------------------------------------------
    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed)             # utils/utils.py:275-280
        self._draws = 0

    def sample_on_device(self, ids: torch.Tensor, times: torch.Tensor, num_neighbors: int):
        """device_random mode: (nbr i32, eid i32, t f32, dt f32), each (n, k), for device ids int32 / times float64 | float32; every call
        advances the generator's key, reset_random_state() rewinds it"""
        key = (0x5EED if self.seed is None else int(self.seed)) * 0x9E3779B97F4A7C15 + self._draws * 0xD1B54A32D192ED03
        self._draws += 1
        return self.graph.sample_random(ids, times, num_neighbors, key, weighted=self.sample_neighbor_strategy == "time_interval_aware")

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
            if self.device_random and num_neighbors <= 128:          # (tg_sample_random's slot limit; GraphMixer's time_gap draw stays on the host)
                dev = _device()
                ids = torch.from_numpy(np.ascontiguousarray(node_ids, dtype=np.int32)).to(dev)
                tt = np.asarray(node_interact_times)
                times = torch.from_numpy(np.ascontiguousarray(tt, dtype=np.float32 if tt.dtype == np.float32 else np.float64)).to(dev)
                nbr, eid, t32, _ = self.sample_on_device(ids, times, num_neighbors)
                return (nbr.cpu().numpy().astype(np.longlong), eid.cpu().numpy().astype(np.longlong), t32.cpu().numpy())
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


def get_neighbor_sampler(data, sample_neighbor_strategy: str = 'uniform', time_scaling_factor: float = 0.0, seed: int = None,
                         device_random: bool = False):
    """mirror of utils/utils.py:283-302.  `data` needs src_node_ids, dst_node_ids, edge_ids, node_interact_times.
    device_random: see NeighborSampler (opt-in, not in the reference)."""
    _device()
    g = TemporalGraph(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    return NeighborSampler(g, sample_neighbor_strategy, time_scaling_factor, seed, device_random)
