"""Torch-free control plane for barcode-sharded runs: a few small host-side exchanges over TCP sockets.

The data plane of a multi-GPU run lives inside libdemux_hip.so (RCCL reduce-scatter / all-gather on the context's
stream).  What the ranks still have to tell each other on the host is tiny - the RCCL unique id, the per-variant
molecule counts of the regularised prior (demuxalot/demux.py:381-384), timings, and on request result rows - and
must not drag a second HIP runtime into the process (importing torch maps torch's own libamdhip64 / librccl next to
the ROCm ones libdemux_hip.so is linked against).  So: one process per GPU, started by any launcher that exports
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (`python -m torch.distributed.run` does), and this module.

Topology: a star.  Rank 0 listens, every other rank holds one connection to it; every operation is
"gather at rank 0, combine, scatter".  All payloads are raw buffers with a fixed little header - no pickles.

Rendezvous: under torchrun MASTER_PORT belongs to the launcher's own store, so rank 0 binds an ephemeral port
and publishes it (with a random token) in a file named after (MASTER_ADDR, MASTER_PORT, TORCHELASTIC_RUN_ID) in the
temporary directory - one node, as the benchmark contract has it; the others poll the file and present the token,
so that a stale file of an earlier run is recognised and waited out.  With an explicit `port` the file is skipped and
the token is derived from DEMUXALOT_AMD_PLANE_TOKEN / the port.  Message lengths come from the peer and are bounded
before anything is allocated; the HELLO of an unauthenticated peer is read with a 12-byte limit.
"""
import os
import socket
import struct
import tempfile
import time

import numpy as np

_HDR = struct.Struct('<4sIq')  # magic, opcode, payload bytes
_MAGIC = b'DMXP'
OP_HELLO, OP_BCAST, OP_SUM, OP_GATHER, OP_BARRIER, OP_MAX, OP_COLL, OP_STATUS = range(8)


def _send(sock, op, payload=b''):
    sock.sendall(_HDR.pack(_MAGIC, op, len(payload)))
    if len(payload):
        sock.sendall(payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('control plane: peer closed the connection')
        got += k
    return buf


_MAX_PAYLOAD = 1 << 36  # 64 GiB: no legitimate message (result rows, staged exchange buffers) comes near


def _recv(sock, expect_op, limit=_MAX_PAYLOAD):
    magic, op, n = _HDR.unpack(bytes(_recv_exact(sock, _HDR.size)))
    if magic != _MAGIC or op != expect_op:
        raise ConnectionError(f'control plane: out-of-step message (op {op}, expected {expect_op}): the ranks did not call '
                              f'the same operations in the same order')
    if not 0 <= n <= limit:  # the length comes from the peer: never allocate what it says unchecked
        raise ConnectionError(f'control plane: message of {n} bytes (op {op}) exceeds the {limit} allowed here')
    return _recv_exact(sock, n)


def _explicit_port_token(port):
    """Token of a rendezvous on an explicit port: DEMUXALOT_AMD_PLANE_TOKEN (shared by the ranks' environments) when set,
    else derived from the port - enough to tell the ranks of one test apart from a stray connection, not a secret."""
    import hashlib
    secret = os.environ.get('DEMUXALOT_AMD_PLANE_TOKEN', '')
    return hashlib.sha256(f'demuxalot_amd plane {secret} {int(port)}'.encode()).digest()[:8]


class SocketControlPlane:
    """The control-plane interface of demuxalot_amd.distributed (rank, world, broadcast_bytes, sum_int64, gather_rows,
    barrier) plus max_float64 / all_ok / gather_to_root, over plain sockets.

        plane = SocketControlPlane.from_env()         # RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT
        plane = SocketControlPlane(rank, world, '127.0.0.1', port=29512)   # tests
    """

    def __init__(self, rank, world, addr='127.0.0.1', port=None, rendezvous_key=None, timeout=180.0,
                 host_collectives=False):
        self.rank, self.world = int(rank), int(world)
        self._peers = []   # rank 0: sockets of ranks 1 .. world-1 (index rank-1)
        self._root = None  # other ranks: socket to rank 0
        self._timeout = timeout
        if host_collectives:
            self.host_collective = self._host_collective
        if self.world == 1:
            return
        key = rendezvous_key or f'{addr}_{os.environ.get("MASTER_PORT", "0")}_{os.environ.get("TORCHELASTIC_RUN_ID", "none")}'
        self._file = os.path.join(tempfile.gettempdir(), 'demuxalot_amd_plane_' + ''.join(ch if ch.isalnum() else '_' for ch in key))
        if self.rank == 0:
            self._listen(addr, port)
        else:
            self._connect(addr, port)

    @classmethod
    def from_env(cls, **kwargs):
        return cls(int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
                   os.environ.get('MASTER_ADDR', '127.0.0.1'), **kwargs)

    # ---- rendezvous ---------------------------------------------------------------------------------
    def _listen(self, addr, port):
        if port is None and os.path.exists(self._file):
            os.unlink(self._file)
        server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        server.bind((addr, 0 if port is None else int(port)))
        server.listen(self.world)
        # The token every peer must present: random and published in the rendezvous file; with an explicit port (tests) the
        # ranks derive it from DEMUXALOT_AMD_PLANE_TOKEN, or - absent that - from the port itself (same-host test harnesses)
        token = os.urandom(8) if port is None else _explicit_port_token(port)
        if port is None:
            tmp = self._file + f'.{os.getpid()}.{token.hex()}'
            # O_EXCL | O_NOFOLLOW, owner-only: a file or symlink planted under the predictable name is not written through
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
            with os.fdopen(fd, 'wb') as f:
                f.write(struct.pack('<I', server.getsockname()[1]) + token)
            os.replace(tmp, self._file)
        server.settimeout(self._timeout)
        peers = {}
        while len(peers) < self.world - 1:
            conn, _ = server.accept()
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.settimeout(self._timeout)
            try:
                hello = bytes(_recv(conn, OP_HELLO, limit=12))  # rank + token: nothing larger is read from a stranger
            except (ConnectionError, socket.timeout, OSError):
                conn.close()
                continue
            if len(hello) != 12:
                conn.close()
                continue
            rank, = struct.unpack('<I', hello[:4])
            if hello[4:12] != token or not 0 < rank < self.world or rank in peers:
                conn.close()  # a stranger, or a rank of an earlier run that read a stale file
                continue
            _send(conn, OP_HELLO, struct.pack('<I', self.world))
            peers[rank] = conn
        server.close()
        if port is None:
            try:
                os.unlink(self._file)
            except OSError:
                pass
        self._peers = [peers[r] for r in range(1, self.world)]

    def _connect(self, addr, port):
        deadline = time.monotonic() + self._timeout
        while True:
            try:
                if port is None:
                    with open(self._file, 'rb') as f:
                        blob = f.read()
                    if len(blob) != 12:
                        raise FileNotFoundError
                    target, token = struct.unpack('<I', blob[:4])[0], blob[4:]
                else:
                    target, token = int(port), _explicit_port_token(port)
                sock = socket.create_connection((addr, target), timeout=5.0)
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                sock.settimeout(self._timeout)
                _send(sock, OP_HELLO, struct.pack('<I', self.rank) + token)
                world, = struct.unpack('<I', bytes(_recv(sock, OP_HELLO)))
                assert world == self.world, f'rank 0 runs a world of {world}, this rank of {self.world}'
                self._root = sock
                return
            except (OSError, ConnectionError):
                if time.monotonic() > deadline:
                    raise TimeoutError(f'control plane: rank {self.rank} could not reach rank 0 at {addr} '
                                       f'({"port file " + self._file if port is None else "port " + str(port)})')
                time.sleep(0.05)

    def close(self):
        for s in self._peers + ([self._root] if self._root is not None else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._root = [], None

    # ---- primitives -----------------------------------------------------------------------------------
    def _gather_at_root(self, op, payload):
        """rank 0: list of every rank's payload (own first); others: None after sending."""
        if self.rank == 0:
            return [bytes(payload)] + [bytes(_recv(s, op)) for s in self._peers]
        _send(self._root, op, payload)
        return None

    def _scatter_from_root(self, op, payload):
        if self.rank == 0:
            for s in self._peers:
                _send(s, op, payload)
            return payload
        return bytes(_recv(self._root, op))

    # ---- the interface of demuxalot_amd.distributed -----------------------------------------------------
    def broadcast_bytes(self, payload):
        """`payload` (bytes) of rank 0 on every rank."""
        if self.world == 1:
            return payload
        return self._scatter_from_root(OP_BCAST, payload if self.rank == 0 else b'')

    def sum_int64(self, array):
        a = np.ascontiguousarray(array, dtype=np.int64)
        if self.world == 1:
            return a
        parts = self._gather_at_root(OP_SUM, a.tobytes())
        total = None
        if self.rank == 0:
            total = np.sum([np.frombuffer(p, dtype=np.int64) for p in parts], axis=0).astype(np.int64)
        out = self._scatter_from_root(OP_SUM, total.tobytes() if self.rank == 0 else b'')
        return np.frombuffer(out, dtype=np.int64).reshape(a.shape).copy()

    def max_float64(self, value):
        if self.world == 1:
            return float(value)
        parts = self._gather_at_root(OP_MAX, struct.pack('<d', float(value)))
        best = struct.pack('<d', max(struct.unpack('<d', p)[0] for p in parts)) if self.rank == 0 else b''
        return struct.unpack('<d', self._scatter_from_root(OP_MAX, best))[0]

    def all_ok(self, ok, message=''):
        """True on every rank iff `ok` on every rank; otherwise every rank gets the first failing rank's message.
        Returns (all_ok, message)."""
        if self.world == 1:
            return bool(ok), message
        parts = self._gather_at_root(OP_STATUS, (b'\1' if ok else b'\0') + message.encode()[:2000])
        verdict = b''
        if self.rank == 0:
            bad = [(r, p[1:].decode(errors='replace')) for r, p in enumerate(parts) if p[:1] != b'\1']
            verdict = b'\1' if not bad else b'\0' + f'rank {bad[0][0]}: {bad[0][1]}'.encode()
        verdict = self._scatter_from_root(OP_STATUS, verdict)
        return verdict[:1] == b'\1', verdict[1:].decode(errors='replace')

    def barrier(self):
        if self.world == 1:
            return
        self._gather_at_root(OP_BARRIER, b'')
        self._scatter_from_root(OP_BARRIER, b'')

    def gather_to_root(self, rows):
        """Rank 0: the ranks' arrays concatenated along axis 0 in rank order (= barcode order); others: None.
        Raw buffers: every rank sends its bytes once, to rank 0 only."""
        rows = np.ascontiguousarray(rows)
        if self.world == 1:
            return rows
        header = struct.pack('<q', rows.shape[0])
        parts = self._gather_at_root(OP_GATHER, header + rows.tobytes())
        if self.rank != 0:
            return None
        tail = rows.shape[1:]
        blocks = [np.frombuffer(p, dtype=rows.dtype, offset=8).reshape((struct.unpack('<q', p[:8])[0],) + tail) for p in parts]
        return np.concatenate(blocks, axis=0)

    def gather_rows(self, rows):
        """The ranks' arrays concatenated along axis 0 on EVERY rank (small arrays only: assignments, column sums;
        result matrices go to rank 0 with gather_to_root or stay on the devices)."""
        rows = np.ascontiguousarray(rows)
        if self.world == 1:
            return rows
        whole = self.gather_to_root(rows)
        if self.rank == 0:
            payload = struct.pack('<q', whole.shape[0]) + whole.tobytes()
        out = self._scatter_from_root(OP_GATHER, payload if self.rank == 0 else b'')
        n, = struct.unpack('<q', out[:8])
        return np.frombuffer(out, dtype=rows.dtype, offset=8).reshape((n,) + rows.shape[1:]).copy()

    # ---- optional: the library's per-iteration exchange over this plane (hosts without a usable RCCL fabric, tests) ----
    def _host_collective(self, op, array):
        """`array`: the library's pinned staging buffer as demuxalot_amd.device hands it over - [count] for all_reduce,
        [world, count] otherwise (include/demux_hip.h: dmx_host_collective).  Sums are formed rank by rank in rank order."""
        def summed(parts):
            total = np.frombuffer(parts[0], dtype=array.dtype).copy()
            for p in parts[1:]:
                total += np.frombuffer(p, dtype=array.dtype)
            return total.tobytes()
        if op == 'all_reduce':
            parts = self._gather_at_root(OP_COLL, array.tobytes())
            out = self._scatter_from_root(OP_COLL, summed(parts) if self.rank == 0 else b'')
            array[...] = np.frombuffer(out, dtype=array.dtype).reshape(array.shape)
        elif op == 'reduce_scatter':  # row `rank` of the sum is what the caller reads
            parts = self._gather_at_root(OP_COLL, array.tobytes())
            out = self._scatter_from_root(OP_COLL, summed(parts) if self.rank == 0 else b'')
            array[self.rank] = np.frombuffer(out, dtype=array.dtype).reshape(array.shape)[self.rank]
        elif op == 'all_gather':  # row `rank` of everybody
            parts = self._gather_at_root(OP_COLL, array[self.rank].tobytes())
            out = self._scatter_from_root(OP_COLL, b''.join(parts) if self.rank == 0 else b'')
            array[...] = np.frombuffer(out, dtype=array.dtype).reshape(array.shape)
        else:
            raise ValueError(op)
