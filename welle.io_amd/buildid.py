"""Identity of the build a counter profile belongs to.  tools/collect_profiles.py writes these hashes into every profiles/*.json it
produces; bench.py recomputes them for the library it has loaded and reports a profile's figures only while they match (a kernel
change without a new tools/make_profiles.sh run then shows up as `stale_profile`, not as silently outdated traffic numbers)."""
import glob
import hashlib
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)


def source_sha256():
    """SHA-256 over the kernel and host sources of the library (csrc/*.hip, *.h, *.cpp, the Makefile and include/dabphy*.h), in name order"""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.hip")) + glob.glob(os.path.join(PKG_DIR, "csrc", "*.h")) +
                   glob.glob(os.path.join(PKG_DIR, "csrc", "*.cpp")) + [os.path.join(PKG_DIR, "csrc", "Makefile"), os.path.join(ROOT, "include", "dabphy.h"), os.path.join(ROOT, "include", "dabphy_test.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def file_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()
