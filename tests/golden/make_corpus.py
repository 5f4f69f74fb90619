"""Freeze the real-text corpora used by the parity tests and the bench's `real_text` leg.

Run once in the build container:  python tests/golden/make_corpus.py
Writes  tests/golden/corpus_prose.txt.xz   (licence texts + Debian copyright files found on this image)
        tests/golden/corpus_python.txt.xz  (CPython 3.10 standard-library sources found on this image)
        tests/golden/corpus_markup.txt.xz  (round 4: HTML of the GDB manual that ROCm ships -- running prose inside markup,
                                            the closest thing on this image to the metric's own corpus, enwik8 = wiki XML)
        tests/golden/corpus_manifest.json  (sizes, SHA-256 of the raw bytes, the file list)

The fixtures are DATA for this repo's tests (no file of the reference repository goes in): public licence texts
(/usr/share/common-licenses, verbatim redistribution permitted), the copyright files Debian ships next to every
package, PSF-licensed standard-library modules, and the GFDL-licensed GDB manual (texinfo HTML under /opt/rocm/share/html/rocgdb).
`python tests/golden/make_corpus.py markup` rebuilds one corpus and leaves the others' fixtures and manifest entries alone.  They replace the globbed stand-ins of round 2, which moved
whenever DESIGN.md was edited and could be missing on another box.
"""
import glob
import hashlib
import json
import lzma
import os

HERE = os.path.dirname(os.path.abspath(__file__))
TARGET = 3 << 20  # 3 MiB raw each (768 chunks of 4 KiB)


def collect(patterns, limit):
    buf, files = bytearray(), []
    for pat in patterns:
        for f in sorted(glob.glob(pat)):
            if os.path.islink(f) or not os.path.isfile(f):
                continue
            try:
                data = open(f, "rb").read()
            except OSError:
                continue
            if not data or b"\0" in data:
                continue
            take = data[: limit - len(buf)]
            buf += take
            files.append({"path": f, "bytes": len(take)})
            if len(buf) >= limit:
                return bytes(buf), files
    return bytes(buf), files


def main():
    import sys

    mpath = os.path.join(HERE, "corpus_manifest.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    for name, patterns in (
        ("prose", ["/usr/share/common-licenses/*", "/usr/share/doc/*/copyright"]),
        ("python", ["/usr/lib/python3.10/*.py", "/usr/lib/python3.10/*/*.py"]),
        ("markup", ["/opt/rocm/share/html/rocgdb/*.html"]),
    ):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        raw, files = collect(patterns, TARGET)
        assert len(raw) >= 2 << 20, (name, len(raw))
        path = os.path.join(HERE, f"corpus_{name}.txt.xz")
        with open(path, "wb") as fh:
            fh.write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        manifest[name] = {
            "file": os.path.basename(path),
            "raw_bytes": len(raw),
            "sha256": hashlib.sha256(raw).hexdigest(),
            "xz_bytes": os.path.getsize(path),
            "sources": files,
        }
        print(name, len(raw), "->", os.path.getsize(path))
    with open(os.path.join(HERE, "corpus_manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)


if __name__ == "__main__":
    main()
