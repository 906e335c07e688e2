"""sys.path set-up shared by tests/, bench.py and __graft_entry__.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "torchrec-oldfork_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
