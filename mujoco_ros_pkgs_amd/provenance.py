"""Fingerprint of the kernel sources: a profile (rocprofv3 counters) only describes the kernels it was collected on.
``tools/summarise_profiles.py`` stores the fingerprint next to the counters, ``bench.py`` refuses counters whose
fingerprint differs from the sources it runs ("a kernel change without a re-profile silently reports stale
counters")."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_sha():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.h")) + [os.path.join(_CSRC, "Makefile")])
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
