"""Decode steps of a whole model replayed from ONE hipGraph (round 4).

A decode step of the reference is ~20 launches and ~40 lines of Python per layer (models/llama_kivi.py:314-399); here it is one
launch per layer, but at small batch the GPU still finishes a layer faster than the host can enqueue the next (0.4-0.5 ms of
enqueue per 32-layer step).  The launches of a step depend on six lengths that change every step, so they cannot simply be
captured -- unless the kernels read the lengths from device memory: `kivi_mf_decode_layer_dyn` (include/kivi_hip.h, kivi_mf_step)
sizes the launch geometry for the step's whole geometry class (same super-block counts, `kivi_mf_step_key`) and the kernels take
Tq / Tv / residual and window lengths from a 32-byte device struct.  All layers of a model share the same lengths, so one struct
serves the model: per step the host uploads six numbers (a one-thread kernel), replays the graph, advances its own copy of the
lengths (`kivi_mf_step_advance`) and -- every residual_length steps -- launches the K flushes (`kivi_kt_pack`), outside the graph.
The graph is re-captured when the geometry class changes (every ~512 steps per side) or a cache had to grow.

Replayed and eager steps agree bit for bit: both follow the launch plan of the step's whole geometry class (kivi_mf_launch_plan --
sized for the class's longest row, ceil(Tq / 512) * 512 + residual_length keys; since round 6 an eager step is planned for that
bound too, and the one-launch caps are whole super-blocks + a full residual, so the bands of round 5 where an eager step ran a block
per row and the replayed one sliced the row are gone: tests/test_graph_gpu.py covers Tq just under 8192 / 9216).
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Callable, List, Optional

import torch

from . import _lib
from .cache_mf import KiviLayerCacheMF


class MfStepDriver:
    """The shared lengths of the layer caches of one model (all KiviLayerCacheMF, all at the same length)."""

    def __init__(self, caches: List[KiviLayerCacheMF]):
        assert caches and all(isinstance(c, KiviLayerCacheMF) and c.ring for c in caches), "matrix-pipe caches only"
        c0 = caches[0]
        # the caches are held WEAKLY: a driver (and the graph captured over it) kept on a model must not keep a finished request's KV
        # cache alive (advisor r5); whoever steps the driver owns the caches
        self._refs = [weakref.ref(c) for c in caches]
        self.lib = _lib.load()
        self.resync()
        self.dev = torch.zeros(4, dtype=torch.int64, device=c0.kt.device)       # kivi_mf_step in device memory (32 bytes)
        self._ptrs = None

    @property
    def caches(self) -> List[KiviLayerCacheMF]:
        cs = [r() for r in self._refs]
        assert all(c is not None for c in cs), "the caches this driver was built for are gone"
        return cs

    def serves(self, caches) -> bool:
        """True when this driver was built for exactly these (still living) cache objects."""
        return len(caches) == len(self._refs) and all(r() is c for r, c in zip(self._refs, caches))

    def resync(self) -> None:
        """Take the lengths from the caches again (they may have been advanced by eager steps or a new prompt since the last call)."""
        self.host = self.caches[0].host_step()
        for c in self.caches[1:]:
            h = c.host_step()
            assert (h.Tq, h.Tv, h.k_res_len, h.v_res_len, h.v_win_start) == (self.host.Tq, self.host.Tv, self.host.k_res_len, self.host.v_res_len, self.host.v_win_start), \
                "the layers of a model advance together"

    # -- per step, in this order: prepare() [-> capture or replay the launches] -> finish()
    def key(self) -> int:
        c = self.caches[0]
        return int(self.lib.kivi_mf_step_key(ctypes.byref(self.host), c.B, c.nh, c.nh_kv, c.cfg.residual_length, c._flags()))

    def prepare(self) -> bool:
        """Room for one more token in every cache, lengths uploaded.  Returns True when captured launches are stale (a cache was
        reallocated or the geometry class changed since the last call)."""
        stale = False
        for c in self.caches:
            c.ensure_room(1)
        # everything a captured launch holds a raw pointer to: the stores, the fp16 residual / window, the per-stream scratch
        ptrs = tuple((c.kt.data_ptr(), c.vt.data_ptr(), c.k_res.data_ptr(), c.v_res.data_ptr()) +
                     tuple(t.data_ptr() for t in c._desc(c.nh, c.kt.device)[4]) for c in self.caches) + (self.key(),)
        if ptrs != self._ptrs:
            stale, self._ptrs = True, ptrs
        _lib.check(self.lib.kivi_mf_step_upload(ctypes.byref(self.host), self.dev.data_ptr(), _lib.stream_ptr(self.dev)), "kivi_mf_step_upload")
        return stale

    def enqueue(self, i: int, q, k, v, out, attention_mask=None):
        """The attend launches of layer i for the prepared step (eagerly, or under stream capture)."""
        return self.caches[i].decode_step_dyn(q, k, v, self.host, self.dev, out, attention_mask)

    def finish(self) -> None:
        """Bookkeeping after the step's launches were enqueued (replayed): lengths advanced, K flushed when the residual is full."""
        c0 = self.caches[0]
        rc = self.lib.kivi_mf_step_advance(ctypes.byref(self.host), c0.cfg.residual_length, c0.v_res.shape[2])
        if rc < 0:
            _lib.check(rc, "kivi_mf_step_advance")
        for c in self.caches:
            c.apply_step(self.host)
        if rc == 1:
            for c in self.caches:
                c.flush_k()
            self.host.Tq += c0.cfg.residual_length
            self.host.k_res_len = 0


class GraphedDecode:
    """`step_fn()` enqueues one whole decode step (dense parts + `driver.enqueue(i, ...)` per layer) on the current stream, reading
    and writing static buffers; `step()` runs it -- the first step of a geometry class eagerly (which also settles everything the
    step allocates lazily), the second one captured into a hipGraph and replayed, the rest replayed."""

    def __init__(self, driver: MfStepDriver, step_fn: Callable[[], None]):
        self.driver, self.step_fn = driver, step_fn
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.warm = False
        self.stream = torch.cuda.Stream(driver.dev.device)      # ONE stream for every capture: the library keeps its scratch per stream
        self.captures = self.replays = self.eager = 0

    def step(self) -> None:
        cur = torch.cuda.current_stream(self.driver.dev.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if self.driver.prepare():                            # room, lengths of THIS step uploaded; stale: reallocation / new class
                self.graph, self.warm = None, False
            if self.graph is None and not self.warm:
                self.step_fn()
                self.warm = True
                self.eager += 1
            else:
                if self.graph is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.stream):   # (records, does not execute)
                        self.step_fn()
                    self.graph = g
                    self.captures += 1
                self.graph.replay()
                self.replays += 1
            self.driver.finish()
        cur.wait_stream(self.stream)
