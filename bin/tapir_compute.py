#!/usr/bin/env python3
"""Drop-in for the reference's bin/tapir_compute.py, backed by the MI355X engine (tapir_amd.cli)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tapir_amd.cli import main  # noqa: E402

if __name__ == '__main__':
    main()
