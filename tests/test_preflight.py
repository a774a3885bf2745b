"""The RCCL preflight guard of bench.py: a failing (here: no GPU / no peers) or hanging child must come back as (False, reason)
within the timeout instead of blocking the caller."""
import socket
import time

from femus_amd import rccl_preflight


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_preflight_reports_failure_without_hanging():
    t0 = time.time()
    # rank 1 of 2 with nobody listening: the child waits for its peer; the guard must cut it off at the timeout
    ok, msg = rccl_preflight.run(1, 2, "127.0.0.1", _free_port(), 0, timeout=20.0)
    assert ok is False and isinstance(msg, str) and msg
    assert time.time() - t0 < 60.0
