"""Build libdfd_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

`python -m deepfakedetection_amd.build` or `build()`; hipcc cross-compiles for
gfx950 without a GPU.  The shared object lands next to this file so that it travels
with the source snapshot to the GPU box.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = CSRC / "build"
LIB_PATH = PKG_DIR / "libdfd_hip.so"
SOURCES = ("dfd_rowpass.hip", "dfd_dwfwd.hip", "dfd_dwbwd.hip", "dfd_dwbwdf.hip", "dfd_dwconv.hip", "dfd_dwmm.hip", "dfd_pwconv.hip", "dfd_pwntw.hip", "dfd_pwntd.hip", "dfd_pwtnw.hip", "dfd_misc.hip", "dfd_vit.hip",
           "dfd_mx.hip", "dfd_attn.hip", "dfd_coord.hip", "dfd_resize.hip", "dfd_augment.hip", "dfd_conv3.hip", "dfd_stem.hip", "dfd_gemm.hip")
ARCH = "gfx950"
FLAGS = ("-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", f"--offload-arch={ARCH}")


def _hipcc() -> str:
    found = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(found).exists():
        raise RuntimeError("hipcc not found: libdfd_hip.so cannot be built")
    return found


def _deps_mtime() -> float:
    headers = list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "dfd_hip.h"]
    return max(p.stat().st_mtime for p in headers)


def source_digest() -> str:
    """sha256 over the kernel sources and the ABI header (names + contents, sorted): what a counter file under profiles/
    records so that bench.py can tell whether it still describes the kernels it is running."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "dfd_hip.h"]):
        h.update(path.name.encode())
        h.update(path.read_bytes())
    return h.hexdigest()


def _compile(src: Path, obj: Path, hipcc: str) -> None:
    cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{proc.stderr[-4000:]}")


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 and link libdfd_hip.so; returns its path."""
    hipcc = _hipcc()
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    hdr_time = _deps_mtime()
    jobs = []
    objs = []
    for name in SOURCES:
        src = CSRC / name
        obj = OBJ_DIR / (src.stem + ".o")
        objs.append(obj)
        stale = force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_time)
        if stale:
            jobs.append((src, obj))
    if jobs:
        if verbose:
            print(f"[dfd build] compiling {[s.name for s, _ in jobs]}", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(lambda so: _compile(so[0], so[1], hipcc), jobs))
    # relink when any object is newer than the library too (an object compiled by hand, e.g. with -Rpass-analysis, is not a "job")
    if jobs or not LIB_PATH.exists() or max(o.stat().st_mtime for o in objs) > LIB_PATH.stat().st_mtime:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB_PATH)]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stderr[-4000:]}")
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path, os.path.getsize(path))
