"""Which kernel sources a measurement belongs to: a hash over the device code of the engine (every kernel header, the
instantiation units, storage / geometry / tuning headers -- not the host engine), written beside committed counter figures
(profiles/pmc_traffic.json) and compared by bench.py, which labels `roofline.traffic` stale when the kernels have changed since."""
from __future__ import annotations

import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_PATTERNS = ("wafer_stencil*.h", "wafer_tu_*.hip", "wafer_tu_*.inc", "wafer_storage.h", "wafer_geom.h", "wafer_rowwalk.h",
             "wafer_tuning.h", "wafer_launch.h")


def kernel_sources() -> list:
    files = set()
    for pat in _PATTERNS:
        files.update(glob.glob(os.path.join(_CSRC, pat)))
    return sorted(files)


def kernel_sources_sha16() -> str:
    h = hashlib.sha256()
    for f in kernel_sources():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
