"""Process-group plumbing for one-process-per-GPU runs - plain TCP, no torch.

The product needs three things from a process group, none of them on the data path: sharing the
RCCL unique id, barriers, and max / sum reductions of a float (timing, checks).  The data-path
collective (all-gather of the scored shards) is RCCL inside libugvc_mi355x.so (csrc/comm.hip); on
CPU-only test runs the same reassembly is exercised with `allgather_results_host`.

Environment: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, the names
`python -m torch.distributed.run` sets - so that launcher (or any other that sets them) works
unchanged, but nothing here imports torch.  The launcher's own store listens on MASTER_PORT, so the
group rendezvouses on the first free port ABOVE it: rank 0 binds it, every other rank walks the same
candidate list and keeps the connection whose peer answers the job's hello token.

Topology: a star through rank 0 (world <= 8 on one node; messages are a few bytes).  The one primitive
is an all-gather of byte strings; barrier, broadcast and the reductions are written on top of it.
"""
from __future__ import annotations

import hashlib
import os
import socket
import struct
import time

import numpy as np

from . import schema as S
from . import shard

_PORT_SPAN = 64          # candidate rendezvous ports: MASTER_PORT + 1 .. + _PORT_SPAN
_TIMEOUT_S = 1200.0


def _send(sock: socket.socket, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def _pack_list(items: list) -> bytes:
    return struct.pack("<I", len(items)) + b"".join(struct.pack("<Q", len(x)) + x for x in items)


def _unpack_list(raw: bytes) -> list:
    (k,) = struct.unpack_from("<I", raw, 0)
    off, out = 4, []
    for _ in range(k):
        (n,) = struct.unpack_from("<Q", raw, off)
        out.append(raw[off + 8: off + 8 + n])
        off += 8 + n
    return out


class Group:
    """world == 1: no sockets at all."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))     # ranks on this host (host threads are shared out by it)
        self._peers: list = []          # rank 0: sockets of ranks 1..world-1 (index r - 1)
        self._up: socket.socket | None = None
        self._listener: socket.socket | None = None
        if self.world > 1:
            if not 0 <= self.rank < self.world:
                raise RuntimeError(f"RANK={self.rank} outside WORLD_SIZE={self.world}")
            addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = int(os.environ.get("MASTER_PORT", "29500"))
            run_id = os.environ.get("TORCHELASTIC_RUN_ID", "") + os.environ.get("UGVC_RUN_ID", "")
            token = hashlib.sha256(f"ugvc-rendezvous:{addr}:{port}:{self.world}:{run_id}".encode()).digest()
            if self.rank == 0:
                self._serve(port, token)
            else:
                self._join(addr, port, token)

    # ---- rendezvous
    def _serve(self, port: int, token: bytes):
        last = None
        for cand in range(port + 1, port + 1 + _PORT_SPAN):
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                ls.bind(("", cand))
            except OSError as e:
                last = e
                ls.close()
                continue
            ls.listen(self.world + 8)
            self._listener = ls
            break
        if self._listener is None:
            raise RuntimeError(f"no free rendezvous port in {port + 1}..{port + _PORT_SPAN}: {last}")
        peers = {}
        deadline = time.monotonic() + _TIMEOUT_S
        self._listener.settimeout(5.0)
        while len(peers) < self.world - 1:
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: {len(peers) + 1} of {self.world} ranks after {_TIMEOUT_S:.0f} s")
            try:
                conn, _ = self._listener.accept()
            except socket.timeout:
                continue
            try:
                conn.settimeout(2.0)
                # the hello is exactly 8 + 36 bytes: read that much and no more before the token has been checked (a
                # stranger's length prefix is never trusted, and a silent one holds the accept loop for 2 s, not 10)
                head = _recv_exact(conn, 8 + 36)
                hello = head[8:]
                if struct.unpack("<Q", head[:8])[0] != 36 or hello[:32] != token:
                    conn.close()                       # a stranger on our port
                    continue
                (r,) = struct.unpack("<I", hello[32:])
                if not 1 <= r < self.world or r in peers:
                    conn.close()
                    continue
                _send(conn, token)
                conn.settimeout(_TIMEOUT_S)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[r] = conn
            except (OSError, ConnectionError, struct.error):
                conn.close()
        self._peers = [peers[r] for r in range(1, self.world)]

    def _join(self, addr: str, port: int, token: bytes):
        deadline = time.monotonic() + _TIMEOUT_S
        while True:
            for cand in range(port + 1, port + 1 + _PORT_SPAN):
                try:
                    s = socket.create_connection((addr, cand), timeout=2.0)
                except OSError:
                    continue
                try:
                    s.settimeout(10.0)
                    _send(s, token + struct.pack("<I", self.rank))
                    if _recv(s) == token:
                        s.settimeout(_TIMEOUT_S)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._up = s
                        return
                except (OSError, ConnectionError, struct.error):
                    pass
                s.close()
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank 0 not reachable at {addr}:{port + 1}..{port + _PORT_SPAN}")
            time.sleep(0.2)

    # ---- the primitive: all-gather of byte strings (rank order)
    def allgather_bytes(self, payload: bytes) -> list:
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            items = [payload] + [_recv(p) for p in self._peers]
            raw = _pack_list(items)
            for p in self._peers:
                _send(p, raw)
            return items
        _send(self._up, payload)
        return _unpack_list(_recv(self._up))

    def barrier(self):
        if self.world > 1:
            self.allgather_bytes(b"")

    def broadcast_bytes(self, payload: bytes | None, src: int = 0) -> bytes:
        if self.world == 1:
            return payload
        return self.allgather_bytes(payload if self.rank == src and payload is not None else b"")[src]

    def max_float(self, x: float) -> float:
        if self.world == 1:
            return x
        return max(struct.unpack("<d", b)[0] for b in self.allgather_bytes(struct.pack("<d", float(x))))

    def sum_float(self, x: float) -> float:
        if self.world == 1:
            return x
        total = 0.0
        for b in self.allgather_bytes(struct.pack("<d", float(x))):       # rank order: every rank forms the same sum
            total += struct.unpack("<d", b)[0]
        return total

    def allgather_results_host(self, local: S.FilterResult, n_total: int) -> S.FilterResult:
        """Host-side equivalent of the RCCL all-gather: padded equal-size columns, rank order
        concatenation.  Used by the CPU tests of the sharding logic."""
        if self.world == 1:
            return local
        cap = shard.shard_cap(n_total, self.world)
        b = shard.shard_bounds(n_total, self.world)
        counts = [int(b[r + 1] - b[r]) for r in range(self.world)]
        padded = shard.pad_result(local, cap)
        raw = b"".join(np.ascontiguousarray(c).tobytes() for c in (padded.tree_score, padded.filter, padded.flags))
        per_rank = []
        for blob in self.allgather_bytes(raw):
            ts = np.frombuffer(blob, np.float32, cap, 0)
            fl = np.frombuffer(blob, np.uint8, cap, 4 * cap)
            fg = np.frombuffer(blob, np.uint8, cap, 5 * cap)
            per_rank.append(S.FilterResult(ts, fl, fg))
        return shard.reassemble(per_rank, counts)

    def close(self):
        for p in self._peers:
            try:
                p.close()
            except OSError:
                pass
        self._peers = []
        for s in (self._up, self._listener):
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._up = self._listener = None
