"""Input path of the step loop: host batch -> pinned staging -> asynchronous H2D copy -> index
preparation ON THE DEVICE (``pgca_seq_batch_prepare``), ``depth`` batches ahead of the step that consumes them.

The reference feeds its loop from a ``DataLoader`` with ``pin_memory`` and moves tensors with Accelerate
(reference data/loader.py:533-572, training/trainer.py:464,575); the batch-dict contract (loader.py:252-258,487-497)
is kept - ``prepare`` is ``DPOStep.prepare`` / ``ContrastiveStep.prepare``.  At > 1000 pairs/s per GPU a synchronous
``.to(device)`` + host-side ``nonzero`` per micro-batch would sit on the critical path; here the copy engine and a side
HIP stream do that work under the previous step's kernels, and the only host wait (the number of scored rows, needed to
size the LM-head launch) happens in the feeder thread.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Iterator

import torch

_STOP = object()


def _pin(x):
    if isinstance(x, torch.Tensor) and not x.is_cuda and not x.is_pinned():
        return x.pin_memory()
    return x


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif hasattr(obj, "__dataclass_fields__"):
        for k in obj.__dataclass_fields__:
            yield from _tensors(getattr(obj, k))


class BatchPrefetcher:
    """Iterates ``loader`` and yields ``prepare(batch, device)`` results whose device work was issued on a side stream
    up to ``depth`` batches ahead.  The consumer's current stream waits on the batch's event (no host block)."""

    def __init__(self, loader: Iterable, prepare: Callable, device, depth: int = 2):
        self.loader, self.prepare, self.device, self.depth = loader, prepare, torch.device(device), max(1, int(depth))

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self) -> Iterator:
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        side = torch.cuda.Stream(device=self.device)
        err = []

        def feed():
            try:
                torch.cuda.set_device(self.device)
                for batch in self.loader:
                    host = {k: _pin(v) for k, v in batch.items()} if isinstance(batch, dict) else batch
                    with torch.cuda.stream(side):
                        out = self.prepare(host, self.device)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    q.put((out, ev, host))       # `host` keeps the pinned staging alive until the copy has run
            except BaseException as e:  # noqa: BLE001 - re-raised in the consumer
                err.append(e)
            finally:
                q.put(_STOP)

        th = threading.Thread(target=feed, name="pgca-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is _STOP:
                    break
                out, ev, _host = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in _tensors(out):          # allocated on the side stream, used on this one
                    if t.is_cuda:
                        t.record_stream(cur)
                yield out
        finally:
            while th.is_alive():                 # drain so the feeder can finish if the consumer stopped early
                try:
                    q.get(timeout=0.1)
                except queue.Empty:
                    pass
            th.join()
        if err:
            raise err[0]
