#!/usr/bin/env python3
"""tests/golden/decoder_resume_fuzz.json: decoder-object call scripts on which a randomised GPU run
(tools/fuzz_resume_gpu.py) once disagreed with the checker, in the record format of decoder_resume.json.

    python tests/golden/make_decoder_resume_fuzz.py [name case.pkl]

Without arguments the calls of every case already in the file are recorded again from ONE reference TampDecompressor
object (oracle/_ref, built in place from /root/reference; build container only).  With arguments one case is added
first: a pickle {blob, script, conf, wb, dic} of the failing object (the fuzzer's RNG replayed on the CPU up to the
round and object its MISMATCH line names).  Cases are data: stream bytes, call script, what the reference returned.
"""
import base64
import json
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.checker import Ref  # noqa: E402

PATH = os.path.join(HERE, "decoder_resume_fuzz.json")
b64 = lambda b: base64.b64encode(bytes(b)).decode()  # noqa: E731


def main():
    recs = json.load(open(PATH)) if os.path.exists(PATH) else []
    if len(sys.argv) > 2:
        c = pickle.load(open(sys.argv[2], "rb"))
        recs = [r for r in recs if r["name"] != sys.argv[1]]
        recs.append(dict(name=sys.argv[1], source="tools/fuzz_resume_gpu.py", data=b64(c["blob"]),
                         conf=list(c["conf"]) if c["conf"] is not None else None, window_bits=c["wb"],
                         dictionary=b64(c["dic"]) if c["dic"] is not None else None, script=[list(s) for s in c["script"]]))
    ref = Ref()
    for r in recs:
        dic = base64.b64decode(r["dictionary"]) if r["dictionary"] else None
        conf = tuple(r["conf"]) if r["conf"] is not None else None
        r0, calls = ref.decode_script(base64.b64decode(r["data"]), [tuple(s) for s in r["script"]], conf=conf,
                                      window_bits=r["window_bits"], dictionary=dic)
        r["init"], r["calls"] = r0, [[st, b64(out), k] for st, out, k in calls]
    json.dump(recs, open(PATH, "w"), indent=0)
    print(f"{len(recs)} case(s) -> {PATH}")


if __name__ == "__main__":
    main()
