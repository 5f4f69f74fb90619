#!/usr/bin/env python3
"""tests/golden/fuzz_regressions.json: inputs on which a randomised GPU run once disagreed with the checker.

    python tests/golden/make_fuzz_regressions.py [name window literal extended lazy input.npy|input.bin]

Without arguments the expected bytes of every case already in the file are recomputed with the REFERENCE C library
(oracle/_ref, built in place from /root/reference; build container only).  With arguments one case is added first:
the failing stream that tools/fuzz_gpu.py leaves in gpurun_out/fuzz_fail_input.npy.  Cases are data: the input bytes,
the configuration, and what the reference produces for them.
"""
import base64
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.checker import Ref  # noqa: E402

PATH = os.path.join(HERE, "fuzz_regressions.json")


def main():
    cases = json.load(open(PATH))["cases"] if os.path.exists(PATH) else []
    if len(sys.argv) > 1:
        name, window, literal, extended, lazy, src = sys.argv[1:7]
        data = np.load(src).tobytes() if src.endswith(".npy") else open(src, "rb").read()
        cases = [c for c in cases if c["name"] != name]
        cases.append({"name": name, "conf": {"window": int(window), "literal": int(literal), "extended": bool(int(extended)),
                                            "lazy_matching": bool(int(lazy))}, "input": base64.b64encode(data).decode()})
    ref = Ref()
    for c in cases:
        data = base64.b64decode(c["input"])
        off, ln = np.zeros(1, np.uint64), np.array([len(data)], np.uint32)
        kw = dict(c["conf"])
        kw["lazy"] = kw.pop("lazy_matching")
        r = ref.compress_batch(np.frombuffer(data, np.uint8), off, ln, **kw)
        c["status"] = int(r.status[0])
        c["expected"] = base64.b64encode(r.stream(0)).decode()
    json.dump({"generator": "tests/golden/make_fuzz_regressions.py", "checker": "oracle/_ref (reference C)", "cases": cases},
              open(PATH, "w"), indent=0)
    print(f"{len(cases)} case(s) -> {PATH}")


if __name__ == "__main__":
    main()
