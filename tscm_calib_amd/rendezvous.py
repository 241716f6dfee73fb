"""Side channel of the multi-process (one process per GPU) runs: a TCP star on the loopback / node network.

It carries what RCCL cannot carry before a communicator exists -- the 128-byte ncclUniqueId from rank 0 to the
other ranks -- and the benchmark's barrier and max-over-ranks of the elapsed time.  Standard library only: the
processes of a multi-GPU run must contain exactly one HIP runtime and one RCCL, the ones libtscm_hip.so links
(importing torch for a gloo process group would bring the wheel's own copies into the process).

Rendezvous: the environment of `torchrun` / `python -m torch.distributed.run` (RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT) or of bench.py's own launcher.  torchrun's c10d store owns MASTER_PORT itself, so rank 0 listens on
the first free port of MASTER_PORT+1 .. MASTER_PORT+32 and the other ranks probe the same list; a token derived from
the launch parameters and the launcher's process id keeps strangers (and stale ranks of earlier launches) out.
TSCM_RDZV_PORT pins the port instead; TSCM_RDZV_NONCE replaces the process id (ranks started by different parents).
"""
from __future__ import annotations

import hashlib
import json
import os
import socket
import struct
import time

_MAGIC = b"TSCMRDZV"
_SPAN = 32


def _token(addr: str, port: int, world: int) -> bytes:
    """What the ranks of ONE launch share and a stale rank of an earlier launch does not: the launcher's random id
    (bench.py's own launcher), and the launcher process itself -- under a static torchrun rendezvous
    TORCHELASTIC_RUN_ID is a constant and MASTER_PORT may be re-used, but every rank of a launch is a child of the same
    agent process (single node: the side channel is a loopback star)."""
    own = os.environ.get("TSCM_RDZV_RUN", "")
    # (bench.py's own launcher hands every rank a random id: the parent's process id adds nothing there, and would keep ranks apart
    # that a wrapper script starts one by one)
    nonce = os.environ.get("TSCM_RDZV_NONCE", "" if own else str(os.getppid()))
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + own + "|" + nonce
    return hashlib.sha256(f"{addr}|{port}|{world}|{run}".encode()).digest()[:16]


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("!I", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("side channel closed by the peer")
        buf += chunk
    return bytes(buf)


def _recv(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("!I", _recv_exact(sock, 4))
    if n > (1 << 24):
        raise ConnectionError("oversized side-channel message")
    return _recv_exact(sock, n)


class SideChannel:
    """rank 0 = hub.  Collectives are blocking and must be called by every rank in the same order."""

    def __init__(self, rank: int, world: int, addr: str | None = None, port: int | None = None, timeout: float = 180.0):
        self.rank, self.world = rank, world
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else os.environ.get("MASTER_PORT", "29533"))
        pinned = os.environ.get("TSCM_RDZV_PORT")
        ports = [int(pinned)] if pinned else [base + 1 + k for k in range(_SPAN)]
        tok = _token(addr, base, world)
        self._peers: list[socket.socket] = []      # hub: socket of rank r at index r - 1
        self._hub: socket.socket | None = None
        self._srv: socket.socket | None = None
        deadline = time.time() + timeout
        if rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", p))
                    srv.listen(world)
                    self.port = p
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError(f"side channel: no free port in {ports[0]}..{ports[-1]} on {addr}")
            self._srv = srv
            slots: dict[int, socket.socket] = {}
            while len(slots) < world - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"side channel: {world - 1 - len(slots)} rank(s) did not connect within {timeout:.0f} s")
                try:
                    c.settimeout(10.0)
                    hello = _recv(c)
                    ok = len(hello) == 8 + 16 + 8 and hello[:8] == _MAGIC and hello[8:24] == tok
                    r, w = struct.unpack("!ii", hello[24:]) if ok else (-1, -1)
                    if not ok or w != world or not (0 < r < world) or r in slots:
                        c.close()
                        continue
                    _send(c, _MAGIC + tok)
                    c.settimeout(None)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    slots[r] = c
                except (OSError, ConnectionError, struct.error):
                    c.close()
            self._peers = [slots[r] for r in range(1, world)]
        else:
            last = None
            while self._hub is None:
                for p in ports:
                    try:
                        c = socket.create_connection((addr, p), timeout=2.0)
                        c.settimeout(10.0)
                        _send(c, _MAGIC + tok + struct.pack("!ii", rank, world))
                        if _recv(c) == _MAGIC + tok:
                            c.settimeout(None)
                            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._hub, self.port = c, p
                            break
                        c.close()
                    except (OSError, ConnectionError, struct.error) as e:
                        last = e
                if self._hub is None:
                    if time.time() > deadline:
                        raise TimeoutError(f"side channel: rank {rank} found no hub on {addr}:{ports[0]}..{ports[-1]} ({last})")
                    time.sleep(0.05)

    # ---------------------------------------------------------------- collectives
    def gather(self, obj):
        """JSON-serialisable `obj` of every rank -> list on rank 0 (None elsewhere)."""
        if self.rank == 0:
            return [obj] + [json.loads(_recv(c).decode()) for c in self._peers]
        _send(self._hub, json.dumps(obj).encode())
        return None

    def bcast(self, payload: bytes | None) -> bytes:
        """Bytes of rank 0 to everyone."""
        if self.rank == 0:
            for c in self._peers:
                _send(c, payload)
            return payload
        return _recv(self._hub)

    def allgather_bytes(self, payload: bytes) -> list[bytes]:
        """Bytes of every rank to everyone, in rank order (the IPC back-end's memory handles)."""
        vals = self.gather(payload.hex())
        out = self.bcast(json.dumps(vals).encode() if self.rank == 0 else None)
        return [bytes.fromhex(x) for x in json.loads(out.decode())]

    def barrier(self) -> None:
        self.gather(0)
        self.bcast(b"go")

    def allreduce_max(self, x: float) -> float:
        vals = self.gather(float(x))
        out = self.bcast(struct.pack("!d", max(vals)) if self.rank == 0 else None)
        return struct.unpack("!d", out)[0]

    def close(self) -> None:
        for c in self._peers + ([self._hub] if self._hub else []) + ([self._srv] if self._srv else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._hub, self._srv = [], None, None
