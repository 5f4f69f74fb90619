"""Command line of the package: ``python -m tamp_amd compress|decompress`` with the options of the reference's CLI.

Mirrors ``tamp/cli/main.py:115-232`` (``tamp compress`` / ``tamp decompress``: ``--input/-i``, ``--output/-o``,
``--window/-w``, ``--literal/-l``, ``--dictionary/-d``, ``--lazy-matching``, ``--extended`` / ``--no-extended``; stdin /
stdout when no path is given; "No data provided." on empty input) and its dictionary rule (``main.py:90-105``): a
dictionary file of exactly ``1 << window`` bytes is used as it is, a shorter one is raw effective bytes -- the seeded
default fills the buffer and the file's contents are copied to its END -- a longer one is an error.  The codec work
runs on the GPU through the same ``tamp_amd.compress`` / ``tamp_amd.decompress`` as everything else; there is no
``--implementation`` choice to make.  ``build-dictionary`` (a corpus-driven offline tool, ``tamp/cli/build_dictionary.py``)
is not part of the codec path and is not provided.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path
from typing import Optional


def load_dictionary(path, window: int, literal: int, extended: bool) -> bytearray:
    """tamp/cli/main.py:90-105."""
    import tamp_amd

    raw = Path(path).read_bytes()
    window_size = 1 << window
    if len(raw) == window_size:
        return bytearray(raw)
    if len(raw) > window_size:
        raise ValueError(f"Dictionary file ({len(raw)} bytes) is larger than window size ({window_size} bytes).")
    dictionary = tamp_amd.initialize_dictionary(window_size, literal=literal if extended else 8)
    if raw:
        dictionary[-len(raw):] = raw
    return dictionary


def _read(path: Optional[str]) -> bytes:
    data = sys.stdin.buffer.read() if path is None else Path(path).read_bytes()
    if not data:
        raise ValueError("No data provided.")
    return data


def _write(path: Optional[str], data: bytes) -> None:
    if path is None:
        sys.stdout.buffer.write(data)
    else:
        Path(path).write_bytes(data)


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
    return ap


def main(argv=None) -> int:
    import tamp_amd

    args = build_parser().parse_args(argv)
    src = args.input if args.input is not None else args.input_pos
    dst = args.output if args.output is not None else args.output_pos
    try:
        data = _read(src)
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
    _write(dst, bytes(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
