"""Batched front for the serving caller of the reference (`/root/reference/app.py:8-15`: one
`generate((sr_in, audio), sr_out, timestep)` per HTTP request, one clip at a time).

Requests from any number of caller threads are collected for a few milliseconds, grouped by (input rate,
steps) and pushed through `FlowHighSR.generate_many`: clips of ANY lengths then run as one ragged launch
sequence on the GPU (equal lengths as one batch) while every caller still gets exactly what `generate()` would
have returned for its clip alone (same per-clip noise draw when a seed is given).  Host logic only: no gradio,
no sockets (the reference's UI / network layers are out of scope); `generate()` below has the signature of
the function `app.py` hands to `gr.Interface`.
"""
import queue
import threading
from concurrent.futures import Future

import numpy as np
import torch


class BatchingServer:
    def __init__(self, model, max_batch=32, max_wait_ms=5.0):
        self.model = model
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1e3
        self._q = queue.Queue()
        self._closed = False
        self._worker = threading.Thread(target=self._run, name="flowhigh-batcher", daemon=True)
        self._worker.start()

    # ---- caller side -------------------------------------------------------------------------------
    def submit(self, audio, sr_in, timestep=1, seed=None):
        """Queue one clip (int16 or float, 1-D or [1, T]); returns a Future of a float32 numpy array [T48]."""
        if self._closed:
            raise RuntimeError("server is closed")
        a = np.asarray(audio.detach().cpu() if isinstance(audio, torch.Tensor) else audio)
        if a.ndim == 2:
            a = a.squeeze(0)
        fut = Future()
        self._q.put((a, int(sr_in), int(timestep), seed, fut))
        return fut

    def generate(self, audio, sr_out=48000, timestep=1):
        """Drop-in for app.py's `generate(audio, sr_out, timestep)`: audio = (sr_in, numpy array)."""
        if int(sr_out) != 48000:
            raise NotImplementedError("the mel codec is fixed at 48 kHz")
        sr_in, a = audio
        return 48000, self.submit(a, sr_in, timestep).result()

    def close(self):
        self._closed = True
        self._q.put(None)
        self._worker.join()

    # ---- worker -----------------------------------------------------------------------------------
    def _collect(self):
        """Block for the first request, then keep taking requests for max_wait or until max_batch."""
        first = self._q.get()
        if first is None:
            return None
        batch = [first]
        deadline = threading.Event()
        timer = threading.Timer(self.max_wait, deadline.set)
        timer.start()
        try:
            while len(batch) < self.max_batch and not deadline.is_set():
                try:
                    item = self._q.get(timeout=self.max_wait / 4 or 1e-4)
                except queue.Empty:
                    continue
                if item is None:
                    self._q.put(None)           # let the outer loop see the shutdown marker
                    break
                batch.append(item)
        finally:
            timer.cancel()
        return batch

    def _run(self):
        while True:
            batch = self._collect()
            if batch is None:
                return
            groups = {}
            for item in batch:
                groups.setdefault((item[1], item[2]), []).append(item)      # same input rate and step count
            for (sr_in, steps), items in groups.items():
                try:
                    noise = None
                    if any(it[3] is not None for it in items):
                        noise = []
                        for a, _, _, seed, _ in items:
                            g = torch.Generator().manual_seed(0 if seed is None else int(seed))
                            t48 = -(-a.shape[-1] * 48000 // sr_in)
                            noise.append(self.model._draw_noise(1, t48 // 480, g))
                    outs = self.model.generate_many([it[0] for it in items], sr_in, 48000, steps, noise=noise,
                                                    max_batch=self.max_batch)
                    for it, y in zip(items, outs):
                        it[4].set_result(y.detach().cpu().squeeze(0).numpy())
                except Exception as e:            # noqa: BLE001  (every waiting caller must be released)
                    for it in items:
                        if not it[4].done():
                            it[4].set_exception(e)
