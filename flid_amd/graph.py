"""TemporalGraph: device-resident time-sorted adjacency (tg_graph) + typed lookups.

replaces the storage half of utils/utils.py:71-110 (python lists of per-node numpy arrays)."""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from ._lib import check, lib
from .ops import _p, _stream


class TemporalGraph:
    def __init__(self, src: np.ndarray, dst: np.ndarray, eid: np.ndarray, t: np.ndarray, num_rows: Optional[int] = None):
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        eid = np.ascontiguousarray(eid, dtype=np.int64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        if not (len(src) == len(dst) == len(eid) == len(t)):
            raise ValueError("src/dst/eid/t must have equal length")
        if num_rows is None:
            num_rows = int(max(src.max(), dst.max())) + 1 if len(src) else 1        # utils/utils.py:293-297
        self.num_rows = int(num_rows)
        self.num_edges = len(src)
        h = C.c_void_p()
        check(lib().tg_graph_create(src.ctypes.data, dst.ctypes.data, eid.ctypes.data, t.ctypes.data, len(src),
                                    self.num_rows, C.byref(h)), "tg_graph_create")
        self._h = h
        self._host = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().tg_graph_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h

    def host_csr(self):
        """(row_ptr i64, nbr i32, eid i32, t f64) copied back once; used by the numpy-facing sampler strategies."""
        if self._host is None:
            n = lib().tg_graph_num_entries(self._h)
            rp = np.empty(self.num_rows + 1, dtype=np.int64)
            nb = np.empty(n, dtype=np.int32)
            ei = np.empty(n, dtype=np.int32)
            tt = np.empty(n, dtype=np.float64)
            check(lib().tg_graph_export(self._h, rp.ctypes.data, nb.ctypes.data, ei.ctypes.data, tt.ctypes.data), "tg_graph_export")
            self._host = (rp, nb, ei, tt)
        return self._host

    def count_before_host(self, ids: np.ndarray, times: np.ndarray) -> np.ndarray:
        """history length of every (node, time) query, computed on the host from the exported CSR (no device work, no sync)"""
        rp, _, _, tt = self.host_csr()
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        times = np.ascontiguousarray(times, dtype=np.float64)
        out = np.empty(len(ids), dtype=np.int64)
        check(lib().tg_host_count_before(rp.ctypes.data, tt.ctypes.data, self.num_rows, ids.ctypes.data, times.ctypes.data, len(ids),
                                         out.ctypes.data), "tg_host_count_before")
        return out

    # ---- device lookups ---------------------------------------------------------------------------
    def sample_recent(self, ids: torch.Tensor, times: torch.Tensor, k: int, out=None, want_dt=True):
        """ids int32 (n,), times float64 or float32 (n,) on the device.  Returns (nbr i32, eid i32, t f32, dt f32|None),
        each (n, k).  `out` may carry four preallocated row-slices to fill in place."""
        n = ids.numel()
        dev = ids.device
        if out is None:
            out = (torch.empty((n, k), dtype=torch.int32, device=dev), torch.empty((n, k), dtype=torch.int32, device=dev),
                   torch.empty((n, k), dtype=torch.float32, device=dev),
                   torch.empty((n, k), dtype=torch.float32, device=dev) if want_dt else None)
        t64 = times if times.dtype == torch.float64 else None
        t32 = times if times.dtype == torch.float32 else None
        if t64 is None and t32 is None:
            raise ValueError("times must be float64 or float32")
        assert ids.dtype == torch.int32 and ids.is_contiguous() and times.is_contiguous()
        check(lib().tg_sample_recent(self._h, _p(ids), _p(t64), _p(t32), n, int(k), _p(out[0]), _p(out[1]), _p(out[2]),
                                     _p(out[3]), C.c_void_p(0), _stream()), "tg_sample_recent")
        return out

    def set_time_weights(self, time_scaling_factor: float):
        """running sums of the time-interval-aware sampling weights (device random mode of the sampler)"""
        check(lib().tg_graph_set_time_weights(self._h, float(time_scaling_factor)), "tg_graph_set_time_weights")
        self._time_weights = float(time_scaling_factor)

    def sample_random(self, ids: torch.Tensor, times: torch.Tensor, k: int, seed: int, weighted: bool = False):
        """k random historical neighbors per query drawn ON THE DEVICE (counter-based generator; not numpy's stream): same outputs as
        sample_recent"""
        n, dev = ids.numel(), ids.device
        out = (torch.empty((n, k), dtype=torch.int32, device=dev), torch.empty((n, k), dtype=torch.int32, device=dev),
               torch.empty((n, k), dtype=torch.float32, device=dev), torch.empty((n, k), dtype=torch.float32, device=dev))
        t64 = times if times.dtype == torch.float64 else None
        t32 = times if times.dtype == torch.float32 else None
        assert ids.dtype == torch.int32 and ids.is_contiguous() and times.is_contiguous() and (t64 is not None or t32 is not None)
        check(lib().tg_sample_random(self._h, _p(ids), _p(t64), _p(t32), n, int(k), int(bool(weighted)), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                     _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), None, _stream()), "tg_sample_random")
        return out

    def first_hop_window(self, ids: torch.Tensor, times: torch.Tensor, max_len: int, width: int):
        n = ids.numel()
        dev = ids.device
        nbr = torch.empty((n, width), dtype=torch.int32, device=dev)
        eid = torch.empty((n, width), dtype=torch.int32, device=dev)
        tt = torch.empty((n, width), dtype=torch.float32, device=dev)
        ln = torch.empty(n, dtype=torch.int32, device=dev)
        check(lib().tg_first_hop_window(self._h, _p(ids), _p(times), n, int(max_len), int(width), _p(nbr), _p(eid), _p(tt),
                                        _p(ln), _stream()), "tg_first_hop_window")
        return nbr, eid, tt, ln

    def recent_window_mean(self, ids: torch.Tensor, times: torch.Tensor, window: int, table: torch.Tensor):
        """GraphMixer's node encoder (models/GraphMixer.py:125-150) without the (n, window, D) intermediate: (n, D)"""
        n = ids.numel()
        out = torch.empty((n, table.shape[1]), dtype=torch.float32, device=ids.device)
        check(lib().tg_recent_window_mean(self._h, _p(ids), _p(times), n, int(window), _p(table), table.stride(0), table.shape[1],
                                          _p(out), out.stride(0), _stream()), "tg_recent_window_mean")
        return out

    def _dedupe_ws_for(self, n, dev):
        """hash-set workspace of the CALLING STREAM: the prefetch path runs on a side stream while a plain call (a validation forward,
        a negative-sample forward) may use the same sampler on the main stream -- one shared workspace let the two corrupt each
        other's row maps, and a regrow could free it under the other stream's kernels"""
        cap = int(lib().tg_dedupe_capacity(n))
        if not isinstance(getattr(self, "_dedupe_ws", None), dict):
            self._dedupe_ws = {}
        key = (str(dev), int(_stream() or 0))
        ws = self._dedupe_ws.get(key)
        if ws is None or ws[0].numel() < cap or ws[2].numel() < n:
            # (values: capacity + 1024 ints -- the numbering passes keep their per-tile totals behind the table, include/flid_tg.h)
            ws = (torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap + 1024, dtype=torch.int32, device=dev),
                  torch.empty(max(n, 1), dtype=torch.int32, device=dev))
            self._dedupe_ws[key] = ws
        return cap, ws

    def dedupe_pairs_async(self, ids: torch.Tensor, t32: torch.Tensor, row_offset: int, out_ids: torch.Tensor, out_t: torch.Tensor,
                           row: torch.Tensor):
        """distinct (id, float32 time) pairs of a sampled level, written into caller buffers (each at least len(ids) long):
        out_ids / out_t = the pairs in the order of their first occurrence in `ids`, row[i] = row_offset + index of slot i's pair.  Returns the DEVICE tensor
        (count, index of the padding pair or -1): nothing is read back here.  The hash-set workspace is per graph object AND per
        stream (calls on two streams do not share it)."""
        n, dev = ids.numel(), ids.device
        cap, ws = self._dedupe_ws_for(n, dev)
        cp = torch.empty(4, dtype=torch.int32, device=dev)        # (count, padding row, scratch, scratch)
        check(lib().tg_dedupe_pairs(_p(ids), _p(t32), n, cap, _p(ws[0]), _p(ws[1]), _p(ws[2]), int(row_offset), _p(out_ids), _p(out_t),
                                    _p(row), _p(cp), _stream()), "tg_dedupe_pairs")
        return cp[:2]

    def dedupe_pairs(self, ids: torch.Tensor, t32: torch.Tensor, row_offset: int):
        """as dedupe_pairs_async with fresh buffers and ONE 8-byte readback: (uniq_ids i32, uniq_t f32, row_of_slot i32, pad_row or -1)"""
        n, dev = ids.numel(), ids.device
        out_ids = torch.empty(n, dtype=torch.int32, device=dev)
        out_t = torch.empty(n, dtype=torch.float32, device=dev)
        row = torch.empty(n, dtype=torch.int32, device=dev)
        count, pad = self.dedupe_pairs_async(ids, t32, row_offset, out_ids, out_t, row).cpu().tolist()
        return out_ids[:count], out_t[:count], row, (pad + row_offset if pad >= 0 else -1)
