"""TEST INFRASTRUCTURE: a control + data plane for several RANKS AS THREADS of one process, so that the multi-rank
exchange of libdemux_hip.so (padded variant slices, reduce-scatter, sliced P-step, all-gather, all-reduce fallback)
runs with world sizes > 1 on a box with ONE GPU: every rank owns a DeviceContext on the same device, and the
collectives are the caller-provided ones of dmx_comm_init_host, done here with a barrier and numpy."""
import threading

import numpy as np


class ThreadWorld:
    def __init__(self, world, timeout=300.):
        self.world, self.timeout = world, timeout
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.collectives = []  # (op, dtype, shape) as rank 0 saw them

    def plane(self, rank):
        return ThreadPlane(self, rank)

    def run(self, target):
        """target(plane) on `world` threads; returns the list of results, re-raises the first failure."""
        results, errors = [None] * self.world, [None] * self.world

        def body(rank):
            try:
                results[rank] = target(self.plane(rank))
            except BaseException as exc:  # noqa: BLE001
                errors[rank] = exc
                self.barrier.abort()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for exc in errors:
            if exc is not None and not isinstance(exc, threading.BrokenBarrierError):
                raise exc
        for exc in errors:
            if exc is not None:
                raise exc
        return results


class ThreadPlane:
    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, value):
        self.shared.slots[self.rank] = value
        self.shared.barrier.wait(self.shared.timeout)
        values = list(self.shared.slots)
        self.shared.barrier.wait(self.shared.timeout)
        return values

    # ---- control plane (demuxalot_amd/distributed.py: SingleProcess) ----
    def broadcast_bytes(self, payload):
        return self._exchange(payload)[0]

    def sum_int64(self, array):
        return np.sum(self._exchange(np.asarray(array, dtype=np.int64)), axis=0)

    def gather_rows(self, rows):
        return np.concatenate(self._exchange(np.ascontiguousarray(rows)), axis=0)

    def gather_to_root(self, rows):
        whole = self.gather_rows(rows)
        return whole if self.rank == 0 else None

    def all_ok(self, ok, message=''):
        reports = self._exchange((bool(ok), message))
        bad = [(r, m) for r, (good, m) in enumerate(reports) if not good]
        return (True, '') if not bad else (False, f'rank {bad[0][0]}: {bad[0][1]}')

    def barrier(self):
        self.shared.barrier.wait(self.shared.timeout)

    # ---- data plane (include/demux_hip.h: dmx_host_collective) ----
    def host_collective(self, op, array):
        if self.rank == 0:
            self.shared.collectives.append((op, array.dtype.name, array.shape))
        theirs = self._exchange(array.copy())
        if op == 'all_reduce':
            total = theirs[0].copy()
            for other in theirs[1:]:
                total += other
            array[...] = total
        elif op == 'reduce_scatter':
            total = theirs[0][self.rank].copy()
            for other in theirs[1:]:
                total += other[self.rank]
            array[self.rank] = total
            for r in range(self.world):  # the other blocks are scratch: make sure nobody relies on them
                if r != self.rank:
                    array[r] = np.nan
        elif op == 'all_gather':
            for r in range(self.world):
                array[r] = theirs[r][r]
        else:
            raise ValueError(op)
