"""Builds libwafer_hip.so (the C-ABI engine) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwafer_hip.so")
# One translation unit per kernel family (wafer_launch.h): they compile in parallel and an edit to one kernel
# rebuilds one unit.  The engine units (wafer_engine*.hip, shared declarations in wafer_engine.h) hold the host logic and the
# small elementwise / set-up kernels.
SOURCES = ["wafer_engine.hip", "wafer_engine_schedules.hip", "wafer_engine_comm.hip", "wafer_engine_solve.hip", "wafer_tu_lds.hip", "wafer_tu_excited_r1.hip", "wafer_tu_excited_r2.hip", "wafer_tu_excited_r3.hip",
           "wafer_tu_fused2.hip", "wafer_tu_fused2w.hip", "wafer_tu_fused3.hip", "wafer_tu_fused3_wide.hip", "wafer_tu_x2.hip", "wafer_mailbox.hip"]
HEADERS = ["wafer_engine.h", "wafer_storage.h", "wafer_geom.h", "wafer_tuning.h", "wafer_launch.h", "wafer_stencil_fused2w.hip.h", "wafer_stencil.hip.h", "wafer_stencil_lds.hip.h",
           "wafer_stencil_fused2.hip.h", "wafer_stencil_fused3.hip.h", "wafer_stencil_fused3_iter.inc.h", "wafer_stencil_x2.hip.h", "wafer_stencil_x2_iter.inc.h", "wafer_elementwise.hip.h", "wafer_rowwalk.h", "wafer_setup.hip.h",
           "wafer_tu_excited.inc"]
# -ffp-contract=off: the stencil update must round exactly like the reference's
# (rustc never fuses mul+add); see DESIGN.md "Parity contract".
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC"]
OBJDIR = os.path.join(HERE, "build")


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine cannot be built")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = {os.path.join(CSRC, f) for f in SOURCES + HEADERS}
    for src in SOURCES:   # whatever the units really include (include/wafer_mailbox.h among them)
        deps.update(_unit_deps(src))
    return any(os.path.getmtime(d) > t for d in deps)


CLI = os.path.join(HERE, "wafer-hip")


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """The host driver (wafer.yaml -> table + observables/wavefunction files), plain g++."""
    src = os.path.join(CSRC, "wafer_cli.cpp")
    deps = [src, os.path.join(CSRC, "wafer_files.h"), LIB]
    if not force and os.path.exists(CLI) and os.path.getmtime(CLI) > max(os.path.getmtime(d) for d in deps):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", src, "-o", CLI, "-L", HERE, "-lwafer_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return CLI


SLABS = os.path.join(HERE, "wafer-hip-slabs")


def build_rccl_host(force: bool = False, verbose: bool = False) -> str:
    """The native multi-GPU host (hooks served by RCCL's C API directly; wafer_rccl_host.cpp)."""
    src = os.path.join(CSRC, "wafer_rccl_host.cpp")
    deps = [src, os.path.join(CSRC, "wafer_rccl_hooks.h"), LIB]
    if not force and os.path.exists(SLABS) and os.path.getmtime(SLABS) > max(os.path.getmtime(d) for d in deps):
        return SLABS
    cmd = [hipcc(), "-O2", "-std=c++17", "-pthread", src, "-o", SLABS, "-L", HERE, "-lwafer_hip", "-lrccl",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SLABS


RCCL_LIB = os.path.join(HERE, "libwafer_rccl.so")


def build_rccl_lib(force: bool = False, verbose: bool = False) -> str:
    """libwafer_rccl.so: the RCCL hooks for hosts that are not C++ (wafer_rccl_lib.cpp)."""
    srcs = [os.path.join(CSRC, "wafer_rccl_lib.cpp"), os.path.join(CSRC, "wafer_rccl_hooks.h"),
            os.path.join(os.path.dirname(HERE), "include", "wafer_rccl.h"), os.path.join(os.path.dirname(HERE), "include", "wafer_mailbox.h")]
    if not force and os.path.exists(RCCL_LIB) and os.path.getmtime(RCCL_LIB) > max([os.path.getmtime(x) for x in srcs] + [os.path.getmtime(LIB)]):
        return RCCL_LIB
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", srcs[0], "-o", RCCL_LIB, "-L", HERE, "-lwafer_hip", "-lrccl",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return RCCL_LIB


def _unit_deps(src: str) -> list:
    """headers a unit includes (transitively; every #include "..." resolved relative to the including file, as the
    preprocessor does): a stale check without running the preprocessor"""
    seen, todo = set(), [os.path.join(CSRC, src)]
    while todo:
        f = todo.pop()
        try:
            text = open(f).read()
        except OSError:
            continue
        for line in text.splitlines():
            line = line.strip()
            if line.startswith('#include "'):
                path = os.path.normpath(os.path.join(os.path.dirname(f), line.split('"')[1]))
                if path not in seen and os.path.exists(path):
                    seen.add(path)
                    todo.append(path)
    return sorted(seen | {os.path.join(os.path.dirname(HERE), "include", "wafer_hip.h")})


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        # WAFER_IEEE_DIV=1 at build time keeps hipcc's IEEE division sequence for the divisions by
        # loop-invariant denominators (wafer_div_invariant, wafer_stencil.hip.h) -- for A/B runs
        extra = ["-DWAFER_IEEE_DIV"] if os.environ.get("WAFER_IEEE_DIV", "") not in ("", "0") else []
        os.makedirs(OBJDIR, exist_ok=True)
        stamp = os.path.join(OBJDIR, "flags.txt")
        flags_now = " ".join(FLAGS + extra)
        if not os.path.exists(stamp) or open(stamp).read() != flags_now:
            force = True
        jobs = []
        for src in SOURCES:
            obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
            deps = [os.path.join(CSRC, src)] + _unit_deps(src)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps):
                cmd = [hipcc(), *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((src, subprocess.Popen(cmd, cwd=CSRC)))
        failed = [src for src, p in jobs if p.wait() != 0]
        if failed:
            raise RuntimeError("hipcc failed on " + ", ".join(failed))
        with open(stamp, "w") as f:
            f.write(flags_now)
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC",
               *[os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES], "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    build_cli(force=force, verbose=verbose)
    build_rccl_host(force=force, verbose=verbose)
    build_rccl_lib(force=force, verbose=verbose)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
