import base64
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unb64(s):
    return base64.b64decode(s) if s is not None else None


@pytest.fixture(scope="session")
def oracle():
    from oracle.checker import ORACLE_SO, Oracle

    if not os.path.exists(ORACLE_SO):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtamp_oracle.so"])
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.checker import Ref

    if not Ref.available():
        pytest.skip("oracle/_ref/libtamp_ref.so not built (reference sources absent)")
    return Ref()


def workload_rows(name):
    """'synth_text:4096' -> generator function and stream length (see tests/golden/make_golden.py)."""
    import numpy as np

    from tamp_amd import workloads as wl

    kind, n = name.split(":")
    n = int(n)
    mask = None
    if "&" in kind:
        kind, m = kind.split("&")
        mask = int(m)
    if kind == "telemetry_bad":
        def gen(count):
            bad = wl.telemetry(4, n).copy()
            bad[1, 40] = 0xC3
            bad[2, 0] = 0x80
            bad[3, 255] = 0xFF
            return bad[:count]
        return gen
    fn = getattr(wl, kind)

    def gen(count):
        rows = fn(count, n)
        return rows & np.uint8(mask) if mask is not None else rows

    return gen
