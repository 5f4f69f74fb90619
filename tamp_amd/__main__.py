"""``python -m tamp_amd compress|decompress ...`` (tamp_amd/cli.py)."""
import sys

from .cli import main

sys.exit(main())
