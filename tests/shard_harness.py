"""Drives `world` renderers as the ranks of one sharded job inside ONE process: a thread per rank (ctypes releases the
GIL around the C calls), the exchange transport an in-memory mailbox behind the fr_comm host callback.  Used by the CPU
tests on the host-logic simulator; the multi-process form (gloo / RCCL) lives in tests/dist_worker.py."""
import queue
import threading

import numpy as np

from libfriendship_amd.capi import Renderer


class Mailboxes:
    def __init__(self, world):
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.bytes_sent = [0] * world
        self.messages = [0] * world

    def transport(self, rank):
        def sendrecv(peer, send, recv):
            if send is not None:
                self.bytes_sent[rank] += send.size
                self.messages[rank] += 1
                self.q[(rank, peer)].put(send.tobytes())
            if recv is not None:
                data = self.q[(peer, rank)].get(timeout=120)
                assert len(data) == recv.size, (len(data), recv.size)
                recv[:] = np.frombuffer(data, dtype=np.uint8)
        return sendrecv


class Job:
    """`world` renderers on `lib`, all given the same graph edits and the same calls."""

    def __init__(self, lib, world, mode, engine_mode="auto", gather=False, serial_exchange=False):
        self.world = world
        self.boxes = Mailboxes(world)
        self.ranks = [Renderer(lib, mode=engine_mode) for _ in range(world)]
        for r, ren in enumerate(self.ranks):
            ren.set_shard(r, world, mode, gather=gather, sendrecv=self.boxes.transport(r), serial_exchange=serial_exchange)

    def each(self, fn):
        """fn(rank, renderer) on every rank concurrently; returns the results, re-raises the first failure."""
        res, err = [None] * self.world, [None] * self.world

        def run(r):
            try:
                res[r] = fn(r, self.ranks[r])
            except BaseException as e:  # noqa: BLE001
                err[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        for e in err:
            if e is not None:
                raise e
        return res

    def fill(self, n_slots, start, end, rows, sentinel=np.float32(-12345.0)):
        """Every rank renders the call into a sentinel-filled buffer; returns the per-rank buffers."""
        def one(_r, ren):
            out = np.full((n_slots, end - start), sentinel, dtype=np.float32)
            return ren.fill_buffer(n_slots, start, end, rows, out=out)
        return self.each(one)

    def assemble(self, bufs, n_slots, sentinel=np.float32(-12345.0)):
        """Rows from their owners; checks that nobody wrote a row it does not own."""
        got = np.empty_like(bufs[0])
        for r, ren in enumerate(self.ranks):
            lo, hi = ren.shard_rows(n_slots)
            got[lo:hi] = bufs[r][lo:hi]
            others = np.ones(n_slots, bool)
            others[lo:hi] = False
            assert np.all(bufs[r][others] == sentinel), f"rank {r} wrote rows it does not own"
        return got

    def close(self):
        for r in self.ranks:
            r.close()
