"""Command line of the package: ``python -m tamp_amd compress|decompress|build-dictionary`` with the options of the reference's CLI.

Mirrors ``tamp/cli/main.py:115-232`` (``tamp compress`` / ``tamp decompress``: ``--input/-i``, ``--output/-o``,
``--window/-w``, ``--literal/-l``, ``--dictionary/-d``, ``--lazy-matching``, ``--extended`` / ``--no-extended``; stdin /
stdout when no path is given; "No data provided." on empty input) and its dictionary rule (``main.py:90-105``): a
dictionary file of exactly ``1 << window`` bytes is used as it is, a shorter one is raw effective bytes -- the seeded
default fills the buffer and the file's contents are copied to its END -- a longer one is an error.  The codec work
runs on the GPU through the same ``tamp_amd.compress`` / ``tamp_amd.decompress`` as everything else; there is no
``--implementation`` choice to make.  ``build-dictionary`` (``tamp/cli/build_dictionary.py:706-927``: a custom dictionary
from a corpus of messages) is ``tamp_amd/build_dictionary.py``: same options and output contract, every whole-corpus
evaluation one batch launch on the GPU.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path
from typing import Optional


def load_dictionary(path, window: int, literal: int, extended: bool) -> bytearray:
    """The reference CLI's dictionary rule (tamp/cli/main.py:90-105): a file of exactly the window size IS the window; a
    shorter one is placed at the END of a seeded window (v1 streams seed with the literal-8 alphabet); a longer one is
    refused."""
    import tamp_amd

    payload = Path(path).read_bytes()
    capacity = 1 << window
    if len(payload) > capacity:
        raise ValueError(f"Dictionary file ({len(payload)} bytes) is larger than window size ({capacity} bytes).")
    if len(payload) == capacity:
        return bytearray(payload)
    seeded = tamp_amd.initialize_dictionary(capacity, literal=literal if extended else 8)
    seeded[capacity - len(payload):] = payload
    return seeded


def _slurp(path: Optional[str]) -> bytes:
    """Whole input from a file, or from stdin when no path was given; empty input is an error like in the reference."""
    blob = Path(path).read_bytes() if path is not None else sys.stdin.buffer.read()
    if len(blob) == 0:
        raise ValueError("No data provided.")
    return blob


def _deliver(path: Optional[str], blob: bytes) -> None:
    if path is not None:
        Path(path).write_bytes(blob)
        return
    sys.stdout.buffer.write(blob)


def _bits(lo: int, hi: int):
    def parse(text: str) -> int:
        v = int(text)
        if not lo <= v <= hi:
            raise argparse.ArgumentTypeError(f"must be in [{lo}, {hi}]")
        return v

    return parse


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="tamp_amd", description="Compress/Decompress data in Tamp format (on an MI355X).")
    sub = ap.add_subparsers(dest="command", required=True)
    for name, help_ in (("compress", "Compress an input file or stream."), ("decompress", "Decompress an input file or stream.")):
        p = sub.add_parser(name, help=help_)
        p.add_argument("input_pos", nargs="?", default=None, metavar="INPUT", help="input file (default: stdin)")
        p.add_argument("output_pos", nargs="?", default=None, metavar="OUTPUT", help="output file (default: stdout)")
        p.add_argument("--input", "-i", default=None)
        p.add_argument("--output", "-o", default=None)
        p.add_argument("--window", "-w", type=_bits(8, 15), default=10, help="bits of the dictionary window")
        p.add_argument("--literal", "-l", type=_bits(5, 8), default=8, help="bits of a literal")
        p.add_argument("--dictionary", "-d", default=None, help="custom initialization dictionary (binary file)")
        p.add_argument("--extended", action=argparse.BooleanOptionalAction, default=True)
        if name == "compress":
            p.add_argument("--lazy-matching", action=argparse.BooleanOptionalAction, default=False)
    b = sub.add_parser("build-dictionary", help="Build a custom dictionary from a corpus of messages.")
    b.add_argument("input", metavar="INPUT", help="directory of sample files, or one file cut at --delimiter")
    b.add_argument("--output", "-o", default="dictionary.bin", help="binary dictionary file (effective bytes only)")
    b.add_argument("--window", "-w", type=_bits(8, 15), default=10)
    b.add_argument("--literal", "-l", type=_bits(5, 8), default=8)
    b.add_argument("--extended", action=argparse.BooleanOptionalAction, default=True)
    b.add_argument("--delimiter", default="\n", help="splits a single input file into samples (default: newline)")
    b.add_argument("--trim-threshold", "-t", type=_bits(2, 1 << 15), default=None,
                   help="shortest substring worth an entry (default: a few values are tried, the best one kept)")
    b.add_argument("--target-fill", "-f", type=float, default=None, help="fraction of the window to fill (default: the knee)")
    b.add_argument("--quiet", "-q", action="store_true")
    return ap


def main(argv=None) -> int:
    import tamp_amd

    args = build_parser().parse_args(argv)
    if args.command == "build-dictionary":
        from tamp_amd import build_dictionary as bd

        if args.target_fill is not None and not 0.0 < args.target_fill <= 1.0:
            print("tamp_amd: --target-fill must be in (0, 1]", file=sys.stderr)
            return 2
        try:
            bd.build_dictionary_cli(args.input, args.output, window=args.window, literal=args.literal, extended=args.extended,
                                    delimiter=args.delimiter, trim_threshold=args.trim_threshold,
                                    target_fill=args.target_fill, quiet=args.quiet)
        except ValueError as e:
            print(f"tamp_amd: {type(e).__name__}: {e}", file=sys.stderr)
            return 1
        return 0
    src = args.input if args.input is not None else args.input_pos
    dst = args.output if args.output is not None else args.output_pos
    try:
        data = _slurp(src)
        kwargs = {}
        if args.dictionary is not None:
            kwargs["dictionary"] = load_dictionary(args.dictionary, args.window, args.literal, args.extended)
        if args.command == "compress":
            out = tamp_amd.compress(data, window=args.window, literal=args.literal, lazy_matching=args.lazy_matching,
                                    extended=args.extended, **kwargs)
        else:
            out = tamp_amd.decompress(data, **kwargs)
    except (ValueError, IndexError, tamp_amd.ExcessBitsError) as e:
        print(f"tamp_amd: {type(e).__name__}: {e}", file=sys.stderr)
        return 1
    _deliver(dst, bytes(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
