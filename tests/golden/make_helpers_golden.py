"""Golden values of the reference's pure-Python helpers (tamp/__init__.py:18-70), recorded by executing the reference's
own definitions in the build container:  python tests/golden/make_helpers_golden.py  -> tests/golden/helpers.json"""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/tamp/__init__.py").read()
ns = {}
exec(compile(src[: src.index("try:", src.index("def compute_min_pattern_size"))], "reference", "exec"), ns)
out = {
    "bit_size": {str(v): ns["bit_size"](v) for v in [0, 1, 2, 3, 255, 256, 65535, 65536, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**40]},
    "min_pattern": {f"{w},{l}": ns["compute_min_pattern_size"](w, l) for w in range(8, 16) for l in range(5, 9)},
    "seeded": [
        {"size": size, "seed": seed, "literal": lit,
         "sha256": hashlib.sha256(bytes(ns["initialize_dictionary"](size, seed=seed, literal=lit))).hexdigest(),
         "head": bytes(ns["initialize_dictionary"](size, seed=seed, literal=lit))[:16].hex()}
        for size in (8, 13, 256, 1024, 32768) for seed in (1, 12345, 0xDEADBEEF, 3758097560) for lit in (5, 6, 7, 8)
    ],
}
json.dump(out, open(os.path.join(HERE, "helpers.json"), "w"), indent=1)
print("written", len(out["seeded"]))
