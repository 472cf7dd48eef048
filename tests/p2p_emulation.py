"""Test infrastructure: an in-process stand-in for ``torch.distributed`` point-to-point calls with the ORDERING RULES OF
RCCL's un-batched P2P, so that hand-off orders that only RCCL would punish can fail on a box with one GPU (or none).

What RCCL does and Gloo does not (torch ``ProcessGroupNCCL::pointToPoint``): every un-batched ``isend`` / ``irecv`` between
two ranks goes through ONE two-rank communicator per rank pair with one internal stream on either side; operations on that
pair run strictly in the order each side issued them, BOTH directions in the same queue, and tags are ignored.  A transfer
happens when a send at the head of one side's queue meets a receive at the head of the other's.  If the two heads are both
sends or both receives, neither can ever complete: that is the GPU hang ADVICE r04 described for ``FrameEmitter``'s
forwards on the pair (N-2, N-1).  Gloo matches by tag from host threads and never shows it.

``PairFifoTransport`` keeps, per (rank, peer), a FIFO of issued operations; ranks are threads of the test process.  A
mismatch of heads raises ``P2POrderError`` in every waiter (and is kept in ``.crossed``); an operation that is never met
raises ``TimeoutError``.  ``work.wait()`` blocks the calling host thread until the operation has been matched -- RCCL would
let the host run ahead, but the ORDER in which operations enter the pair queues is program order either way -- and, for GPU
tensors, makes the current stream wait for the copy (what an RCCL ``work.wait()`` does).
"""

from __future__ import annotations

import collections
import threading

import torch


class P2POrderError(RuntimeError):
    pass


class _Op:
    __slots__ = ("kind", "tensor", "ready", "done", "matched", "who")

    def __init__(self, kind, tensor, ready, who):
        self.kind, self.tensor, self.ready, self.who = kind, tensor, ready, who
        self.done, self.matched = None, False


class _Work:
    def __init__(self, transport, op):
        self._t, self._op = transport, op

    def wait(self):
        self._t._wait(self._op)
        return True

    def is_completed(self):
        return self._op.matched


class PairFifoTransport:
    def __init__(self, timeout: float = 120.0) -> None:
        self.timeout = timeout
        self._cv = threading.Condition()
        self._q: dict = collections.defaultdict(collections.deque)      # (me, peer) -> ops this rank issued towards peer
        self._me = threading.local()
        self.crossed: str | None = None
        self.log: list = []                                             # (rank, kind, peer) in global issue order
        self._copy_stream = None

    # ------------------------------------------------------------------ who am I
    def bind(self, rank: int) -> None:
        self._me.rank = rank

    @property
    def rank(self) -> int:
        return self._me.rank

    # ------------------------------------------------------------------ torch.distributed look-alikes
    def isend(self, tensor, dst, tag=0, group=None):
        ready = None
        if tensor.is_cuda:
            staged = tensor.clone()                    # on the issuing stream: ordered behind what that stream already holds
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(tensor.device))
        else:
            staged = tensor.clone()
        return self._issue(_Op("s", staged, ready, (self.rank, dst)), dst)

    def irecv(self, tensor, src, tag=0, group=None):
        ready = None
        if tensor.is_cuda:
            ready = torch.cuda.Event()                 # the buffer may still be in use by earlier work of the issuing stream
            ready.record(torch.cuda.current_stream(tensor.device))
        return self._issue(_Op("r", tensor, ready, (self.rank, src)), src)

    def send(self, tensor, dst, tag=0, group=None):
        self.isend(tensor, dst, tag).wait()

    def recv(self, tensor, src=None, tag=0, group=None):
        self.irecv(tensor, src, tag).wait()

    def install(self, monkeypatch, dist_module) -> None:
        for name in ("isend", "irecv", "send", "recv"):
            monkeypatch.setattr(dist_module, name, getattr(self, name))
        monkeypatch.setattr(dist_module, "is_initialized", lambda: False)   # no process group behind this

    # ------------------------------------------------------------------ the pair queues
    def _issue(self, op, peer):
        me = self.rank
        with self._cv:
            if self.crossed:
                raise P2POrderError(self.crossed)
            self.log.append((me, op.kind, peer))
            self._q[(me, peer)].append(op)
            self._progress(me, peer)
        return _Work(self, op)

    def _progress(self, a, b):
        mine, theirs = self._q[(a, b)], self._q[(b, a)]
        while mine and theirs:
            x, y = mine[0], theirs[0]
            if x.kind == y.kind:
                word = "sends" if x.kind == "s" else "receives"
                self.crossed = (f"pair ({min(a, b)}, {max(a, b)}): both sides have {word} at the head of their queues "
                                f"(rank {a}: {[o.kind for o in mine]}, rank {b}: {[o.kind for o in theirs]}) -- on RCCL "
                                f"these two streams wait for each other for ever")
                self._cv.notify_all()
                raise P2POrderError(self.crossed)
            s, r = (x, y) if x.kind == "s" else (y, x)
            if r.tensor.is_cuda:
                if self._copy_stream is None:
                    self._copy_stream = torch.cuda.Stream(device=r.tensor.device)
                cs = self._copy_stream
                cs.wait_event(s.ready)
                cs.wait_event(r.ready)
                with torch.cuda.stream(cs):
                    r.tensor.copy_(s.tensor)
                    done = torch.cuda.Event()
                    done.record(cs)
                s.tensor.record_stream(cs)
                s.done = r.done = done
            else:
                r.tensor.copy_(s.tensor)
            s.matched = r.matched = True
            mine.popleft()
            theirs.popleft()
        self._cv.notify_all()

    def _wait(self, op):
        with self._cv:
            ok = self._cv.wait_for(lambda: op.matched or self.crossed is not None, timeout=self.timeout)
            if self.crossed is not None and not op.matched:
                raise P2POrderError(self.crossed)
            if not ok:
                raise TimeoutError(f"p2p emulation: rank {op.who[0]}'s {'send to' if op.kind == 's' else 'receive from'} "
                                   f"rank {op.who[1]} was never met")
        if op.done is not None:
            torch.cuda.current_stream(op.tensor.device).wait_event(op.done)

    def idle(self) -> bool:
        with self._cv:
            return all(not q for q in self._q.values())

    def directions_per_pair(self) -> dict:
        """{(low, high): set of (src, dst)} over everything issued: which pairs carried traffic in both directions."""
        out: dict = collections.defaultdict(set)
        for rank, kind, peer in self.log:
            src, dst = (rank, peer) if kind == "s" else (peer, rank)
            out[(min(src, dst), max(src, dst))].add((src, dst))
        return dict(out)
