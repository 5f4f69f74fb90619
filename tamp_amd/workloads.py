"""Deterministic synthetic inputs for BASELINE.json's configs (host-side data only, no codec logic).

Thin ctypes front-end over ``libtamp_workloads.so`` (tamp_amd/csrc/workloads.c); see SURVEY.md
section 8(d) for the shapes.  Every generator returns a C-contiguous ``uint8`` array of shape
``(n_streams, stream_len)``: stream ``i`` (global index ``first_index + i``) is row ``i``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtamp_workloads.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise FileNotFoundError(f"{_SO} missing: run __graft_entry__.build() or `make -C tamp_amd/csrc`")
        _lib = C.CDLL(_SO)
        for name in ("wl_synth_text", "wl_telemetry", "wl_lcg_runs", "wl_stress"):
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_int]
            getattr(_lib, name).restype = None
        _lib.wl_telemetry_dictionary.argtypes = [C.c_void_p]
    return _lib


def _gen(name: str, n_streams: int, stream_len: int, first_index: int, threads: int | None) -> np.ndarray:
    out = np.empty((n_streams, stream_len), dtype=np.uint8)
    if n_streams and stream_len:
        getattr(_load(), name)(out.ctypes.data_as(C.c_void_p), n_streams, stream_len, first_index,
                               threads or min(32, os.cpu_count() or 1))
    return out


def synth_text(n_streams: int, stream_len: int = 4096, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """Config 2: Zipf(1.0) word sampler over a 512-word vocabulary, one xorshift32 stream per row."""
    return _gen("wl_synth_text", n_streams, stream_len, first_index, threads)


def telemetry(n_streams: int, stream_len: int = 256, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """Config 5: 7-bit JSON-ish telemetry messages, space padded."""
    return _gen("wl_telemetry", n_streams, stream_len, first_index, threads)


def lcg_runs(n_streams: int, stream_len: int = 512, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """Run-heavy two-letter data (the reference's LCG fuzz corpus recurrence)."""
    return _gen("wl_lcg_runs", n_streams, stream_len, first_index, threads)


def stress(n_streams: int, stream_len: int = 8192, first_index: int = 0, threads: int | None = None) -> np.ndarray:
    """xorshift stress shapes: incompressible / long runs / long repeats (row index mod 3)."""
    return _gen("wl_stress", n_streams, stream_len, first_index, threads)


def telemetry_dictionary(seeded_default_256: bytes) -> bytes:
    """Config 5's shared custom dictionary: the seeded 256-byte default with the field skeleton at its tail."""
    buf = np.frombuffer(bytes(seeded_default_256), dtype=np.uint8).copy()
    assert buf.size == 256
    _load().wl_telemetry_dictionary(buf.ctypes.data_as(C.c_void_p))
    return buf.tobytes()


def csr_for_fixed(n_streams: int, stream_len: int):
    """(in_off uint64[n], in_len uint32[n]) for equally sized, densely packed streams."""
    in_off = np.arange(n_streams, dtype=np.uint64) * np.uint64(stream_len)
    in_len = np.full(n_streams, stream_len, dtype=np.uint32)
    return in_off, in_len


# ---- real inputs: files cut into independent streams (host-side data preparation only) ----------------------

#: Real-text stand-ins that exist in this image (and on the GPU box) while the metric's own corpus, enwik8, does not
#: (no network): documentation / licence prose and Python sources.  name -> glob patterns, read in sorted order.
REAL_TEXT_SOURCES = {
    "prose": ["/opt/skills/guides/*.md", "{repo}/*.md", "/usr/share/common-licenses/*", "/usr/share/doc/*/copyright"],
    "python": ["/usr/lib/python3.10/*.py", "/usr/lib/python3.10/*/*.py"],
    "markup": ["/opt/rocm/share/html/rocgdb/*.html"],
}

#: SHA-256 of the reference C's output for the whole enwik8 file as ONE stream, window=10 literal=8, no lazy matching
#: (/root/reference/tests/test_dataset_regression.py:38-43): (v1 format, extended format); sizes from README.md:266.
ENWIK8_PINS = {
    "len": 100_000_000,
    "v1_sha256": "02e05af059a0040d641988075cf1dfc479a084f9a34b5c8a348354211c5fa038",
    "extended_sha256": "d9d804c91b4dc5e81856db074760037040421cfa84e1ea211e16dde8c295ce6d",
    "v1_size": 51_635_633,
    "extended_size": 51_016_917,
    "first_100k_v1_size": 50_841,  # README.md:336
}


def gather_files(patterns, max_bytes: int) -> bytes:
    """Concatenation of the files matching ``patterns`` (sorted per pattern), cut at ``max_bytes``."""
    import glob

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    buf = bytearray()
    for pat in patterns:
        for f in sorted(glob.glob(pat.format(repo=repo))):
            try:
                with open(f, "rb") as fh:
                    buf += fh.read()
            except OSError:
                continue
            if len(buf) >= max_bytes:
                return bytes(buf[:max_bytes])
    return bytes(buf)


#: frozen real-text corpora (tests/golden/make_corpus.py): 3 MiB each, xz-compressed in the tree
FROZEN_CORPORA = {"prose": "corpus_prose.txt.xz", "python": "corpus_python.txt.xz", "markup": "corpus_markup.txt.xz"}
_corpus_cache = {}


def frozen_corpus(name: str) -> bytes:
    """The committed fixture for ``name`` (raw bytes; checked against the manifest's SHA-256), or b"" if absent."""
    if name in _corpus_cache:
        return _corpus_cache[name]
    import hashlib
    import json
    import lzma

    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    path = os.path.join(gold, FROZEN_CORPORA.get(name, ""))
    raw = b""
    if os.path.isfile(path):
        raw = lzma.decompress(open(path, "rb").read())
        with open(os.path.join(gold, "corpus_manifest.json")) as fh:
            want = json.load(fh)[name]
        if hashlib.sha256(raw).hexdigest() != want["sha256"] or len(raw) != want["raw_bytes"]:
            raise RuntimeError(f"frozen corpus {name!r} does not match its manifest")
    _corpus_cache[name] = raw
    return raw


def real_text(name: str, max_bytes: int = 64 << 20, frozen_only: bool = False) -> bytes:
    """Real text for throughput / parity runs: the frozen 3 MiB fixture of ``name`` (tests/golden/corpus_<name>.txt.xz, so
    that numbers do not move with the machine or with this repo's own markdown), cut at ``max_bytes``.  The fixture IS the
    corpus: asking for more than it holds returns all of it -- callers that need more streams tile it (``tile_rows``).
    Only when the fixture is absent (an installed package without the repo's tests/ directory next to it) and
    ``frozen_only`` is false are the files of REAL_TEXT_SOURCES globbed instead, which is machine dependent."""
    raw = frozen_corpus(name)
    if raw or frozen_only:
        return raw[:max_bytes]
    return gather_files(REAL_TEXT_SOURCES[name], max_bytes)


#: where bench.py looks for the metric's own corpus (BASELINE.json: enwik8) when no --corpus / $TAMP_CORPUS names one
CORPUS_PROBE_DIRS = (".", "~", "/data", "/tmp", "~/datasets", "./datasets")
#: the URL the reference's own Makefile downloads it from (/root/reference/Makefile:159-166)
ENWIK8_URL = "https://mattmahoney.net/dc/enwik8.zip"


def probe_corpus(env=None, dirs=None, fetch=False, fetch_to="/tmp", timeout_s: float = 15.0, want_len: int = ENWIK8_PINS["len"]):
    """Look for enwik8 (plain file or ``enwik8.zip``) without being told where: ``$TAMP_CORPUS``, then CORPUS_PROBE_DIRS,
    then ONE guarded download attempt of ENWIK8_URL (``timeout_s`` seconds in all; any failure is silent -- the build
    and GPU boxes have no network).  A candidate counts only if it holds exactly ``want_len`` bytes.  A zip is unpacked
    next to ``fetch_to``.  -> (path or None, record); the record (what was tried, what was found) goes into bench.py's
    JSON line as ``config.corpus_probe`` whether or not anything turned up."""
    import time
    import zipfile

    env = os.environ if env is None else env
    rec = {"tried": [], "found": None, "fetch": "not attempted"}

    def good(path):
        try:
            return os.path.isfile(path) and os.path.getsize(path) == want_len
        except OSError:
            return False

    def unzip(zpath):
        try:
            with zipfile.ZipFile(zpath) as z:
                for info in z.infolist():
                    if info.file_size == want_len:
                        os.makedirs(fetch_to, exist_ok=True)
                        out = os.path.join(fetch_to, "enwik8")
                        with z.open(info) as src, open(out, "wb") as dst:
                            while True:
                                chunk = src.read(1 << 22)
                                if not chunk:
                                    break
                                dst.write(chunk)
                        return out if good(out) else None
        except (OSError, zipfile.BadZipFile, RuntimeError):
            pass
        return None

    cands = []
    if env.get("TAMP_CORPUS"):
        cands.append(env["TAMP_CORPUS"])
    for d in (CORPUS_PROBE_DIRS if dirs is None else dirs):
        d = os.path.expanduser(d)
        cands += [os.path.join(d, "enwik8"), os.path.join(d, "enwik8.zip")]
    for c in cands:
        rec["tried"].append(c)
        if c.endswith(".zip"):
            if os.path.isfile(c):
                out = unzip(c)
                if out:
                    rec["found"] = out
                    return out, rec
        elif good(c):
            rec["found"] = c
            return c, rec
    if fetch:
        t0 = time.monotonic()
        zpath = os.path.join(fetch_to, "enwik8.zip")
        try:
            import urllib.request

            os.makedirs(fetch_to, exist_ok=True)
            with urllib.request.urlopen(ENWIK8_URL, timeout=min(5.0, timeout_s)) as resp, open(zpath + ".part", "wb") as fh:
                while True:
                    if time.monotonic() - t0 > timeout_s:
                        raise TimeoutError(f"over {timeout_s:.0f} s")
                    chunk = resp.read(1 << 20)
                    if not chunk:
                        break
                    fh.write(chunk)
            os.replace(zpath + ".part", zpath)
            out = unzip(zpath)
            rec["fetch"] = "ok" if out else "downloaded, but no member of %d bytes" % want_len
            if out:
                rec["found"] = out
                return out, rec
        except Exception as e:  # noqa: BLE001 -- no network is the normal case here
            rec["fetch"] = "failed after %.1f s: %s" % (time.monotonic() - t0, repr(e)[:80])
            try:
                os.remove(zpath + ".part")
            except OSError:
                pass
        rec["tried"].append(ENWIK8_URL)
    return None, rec


def split_fixed(blob, chunk: int = 4096, keep_tail: bool = True):
    """A byte string as independent streams of ``chunk`` bytes: (flat uint8, in_off uint64[n], in_len uint32[n]).

    BASELINE configs[2] ("enwik8 split into independent 4 KiB streams"): the last, shorter piece is KEPT as a short
    stream (100,000,000 B -> 24,414 streams of 4096 B + one of 576 B) so that every input byte is compressed;
    ``keep_tail=False`` drops it.
    """
    flat = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob.reshape(-1)
    total = int(flat.size)
    n_full = total // chunk
    tail = total - n_full * chunk
    n = n_full + (1 if (tail and keep_tail) else 0)
    in_off = np.arange(n, dtype=np.uint64) * np.uint64(chunk)
    in_len = np.full(n, chunk, dtype=np.uint32)
    if tail and keep_tail:
        in_len[-1] = tail
    else:
        flat = flat[: n_full * chunk]
    return flat, in_off, in_len


#: streams of BASELINE configs[2] (enwik8 cut into 4 KiB pieces: 24,414 full ones; the 576-byte tail aside)
CONFIGS2_STREAMS = 24_414


def standin_rows(n_streams: int = CONFIGS2_STREAMS, chunk: int = 4096, seed: int = 2) -> np.ndarray:
    """Stand-in for BASELINE configs[2] while enwik8 is out of reach: every distinct 4 KiB chunk of the three frozen
    corpora (prose, markup, Python sources), tiled to ``n_streams`` rows and shuffled with a fixed seed -- a real-text
    batch of the metric's own shape (24,414 streams) whose costs are as uneven as real text makes them."""
    parts = []
    for name in ("prose", "markup", "python"):
        flat = np.frombuffer(real_text(name, frozen_only=True), dtype=np.uint8)
        k = flat.size // chunk
        if k:
            parts.append(flat[: k * chunk].reshape(k, chunk))
    if not parts:
        raise ValueError("the frozen corpora (tests/golden/corpus_*.txt.xz) are not next to this package")
    base = np.concatenate(parts)
    rng = np.random.default_rng(seed)
    reps = (n_streams + len(base) - 1) // len(base)
    idx = np.concatenate([rng.permutation(len(base)) for _ in range(reps)])[:n_streams]  # every chunk equally often
    return np.ascontiguousarray(base[idx])


def tile_rows(blob, n_streams: int, chunk: int = 4096) -> np.ndarray:
    """``n_streams`` rows of ``chunk`` bytes cut from ``blob``, repeating it when it is shorter (throughput runs)."""
    flat = np.frombuffer(bytes(blob), dtype=np.uint8)
    k = flat.size // chunk
    if k == 0:
        raise ValueError("corpus shorter than one chunk")
    rows = flat[: k * chunk].reshape(k, chunk)
    reps = (n_streams + k - 1) // k
    return np.ascontiguousarray(np.tile(rows, (reps, 1))[:n_streams])
