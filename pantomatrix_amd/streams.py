"""Fork/join of independent launch chains onto side HIP streams.

Every GEMM of a 64-clip window is small (a few microseconds of MI355X time) and carries a fixed ramp: the
per-XCD L2s are invalidated at kernel boundaries, so each launch restarts from the Infinity Cache, fills its
LDS ring, and drains through its epilogue.  The EMAGE graph has several independent chains (two WavEncoders, the
motion pre-encoder, the face decoder vs the body stack, three refinement layers, four part decoders + the global
AE), so they are issued on separate streams: one chain's ramp overlaps another chain's steady state.  Under
hipGraph capture the same fork/join events become parallel graph branches.
"""
from __future__ import annotations

import torch

_POOLS = {}


def _side_streams(device, n, parent):
    """Side streams owned by `parent` (keyed by its handle): nested forks issued from different lanes get
    different streams, so two concurrent sub-graphs never queue behind each other by accident."""
    pool = _POOLS.setdefault((device, parent.cuda_stream), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


class Fork:
    """`with` a Fork, lane 0 is the caller's stream and lanes 1..n-1 are side streams that start after everything
    already enqueued on the caller's stream; leaving the block joins them back.  On a CPU device (tests with the
    kernel layer patched out) it degenerates to sequential execution."""

    def __init__(self, device, lanes: int, enabled: bool = True):
        self.cuda = enabled and torch.device(device).type == "cuda" and lanes > 1
        self.lanes = lanes
        if self.cuda:
            self.main = torch.cuda.current_stream(device)
            self.side = _side_streams(torch.device(device), lanes - 1, self.main)

    def __enter__(self):
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(self.main)
            for s in self.side:
                s.wait_event(ev)
        return self

    def __exit__(self, *exc):
        if self.cuda:
            for s in self.side:
                ev = torch.cuda.Event()
                ev.record(s)
                self.main.wait_event(ev)
        return False

    def lane(self, i):
        """Context manager that makes lane i's stream current."""
        if not self.cuda:
            return _Null()
        return torch.cuda.stream(self.main if i == 0 else self.side[i - 1])

    def after(self, waiter: int, signaler: int):
        """Lane `waiter` waits for everything enqueued so far on lane `signaler`."""
        if not self.cuda or waiter == signaler:
            return
        sw = self.main if waiter == 0 else self.side[waiter - 1]
        ss = self.main if signaler == 0 else self.side[signaler - 1]
        ev = torch.cuda.Event()
        ev.record(ss)
        sw.wait_event(ev)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
