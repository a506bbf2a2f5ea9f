"""Host-side launch logic that needs no GPU: compiled with g++ from the headers the engine uses."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wafer_amd", "csrc")

HARNESS = r"""
#include "wafer_tuning.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
    // per_layer nplanes slots fill  ->  zchunk
    for (int i = 1; i + 3 < argc; i += 4)
        printf("%d\n", wafer_pick_zchunk(atoll(argv[i]), atoi(argv[i + 1]), atoll(argv[i + 2]), atoi(argv[i + 3])));
    return 0;
}
"""


@pytest.fixture(scope="module")
def pick(tmp_path_factory):
    d = tmp_path_factory.mktemp("host")
    src, exe = d / "pick.cpp", d / "pick"
    src.write_text(HARNESS)
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-fsanitize=address,undefined", "-I", CSRC, str(src), "-o", str(exe)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]

    def call(*cases):
        args = [str(x) for c in cases for x in c]
        out = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr[-2000:]
        return [int(x) for x in out.stdout.split()]
    return call


def makespan(per_layer, nplanes, slots, fill, zc):
    nch = -(-nplanes // zc)
    rounds = -(-per_layer * nch // slots)
    return rounds * (zc + fill)


def test_zchunk_choices_of_the_known_grids(pick):
    """what the three-step kernel (fill 6, 256 CUs) chooses: unchanged where rounds 1-2 measured it, the fix at 384^3"""
    got = pick((128, 512, 256, 6), (32, 256, 256, 6), (72, 384, 256, 6), (512, 128, 256, 6), (512, 1024, 256, 6))
    assert got == [256, 32, 55, 128, 1024]


@pytest.mark.parametrize("slots,fill", [(256, 6), (512, 3), (256, 5), (304, 6)])
def test_zchunk_minimises_the_makespan(pick, slots, fill):
    """exhaustively against a Python statement of the same cost, on awkward tile counts and plane counts"""
    cases = [(pl, n, slots, fill) for pl in (1, 3, 7, 32, 60, 72, 100, 128, 200, 512, 2048) for n in (1, 2, 5, 31, 64, 100, 128, 384, 1000)]
    got = pick(*cases)
    for (pl, n, s, f), zc in zip(cases, got):
        assert 1 <= zc <= n
        best = min(makespan(pl, n, s, f, -(-n // nch)) for nch in range(1, min(n, 64) + 1))
        assert makespan(pl, n, s, f, zc) == best, (pl, n, zc)


def test_zchunk_degenerate_arguments(pick):
    zc = pick((0, 10, 256, 6), (5, 1, 256, 6), (5, 10, 0, 6))
    assert all(1 <= z <= n for z, n in zip(zc, (10, 1, 10)))


# ---- workgroup schedules of the three-step kernel (wafer_stencil_fused3.hip.h, host code) ---------------------------------
SCHED = r"""
#include "wafer_stencil_fused3.hip.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
    std::vector<WaferF3Block> t;
    const int kind = atoi(argv[1]), ntx = atoi(argv[2]), nty = atoi(argv[3]), lo = atoi(argv[4]), hi = atoi(argv[5]);
    if (kind == 0) wafer_f3_schedule_plain(t, ntx, nty, lo, hi, atoi(argv[6]), atoi(argv[7]) != 0);
    else if (kind == 1) wafer_f3_schedule_mixed(t, ntx, nty, lo, hi, atoi(argv[6]));
    else {
        const bool nw[2] = {atoi(argv[8]) != 0, atoi(argv[9]) != 0};
        wafer_f3_schedule_halves(t, ntx, nty, lo, hi, atoi(argv[6]), atoi(argv[7]), nw, atoi(argv[10]), atoi(argv[11]), atoi(argv[12]), true, 0,
                                 atoi(argv[13]));
    }
    for (const auto &b : t) printf("%d %d %d %d %d %d %d %d\n", b.tile, b.zs, b.ze, b.down, b.wait_late, b.wait_it, b.bump, b.wt);
    return 0;
}
"""


@pytest.fixture(scope="module")
def sched(tmp_path_factory):
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("sched")
    src, exe = d / "sched.hip", d / "sched"
    src.write_text(SCHED)
    r = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", CSRC, str(src), "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]

    def call(*args):
        out = subprocess.run([str(exe), *[str(a) for a in args]], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr[-2000:]
        keys = ("tile", "zs", "ze", "down", "wait_late", "wait_it", "bump", "wt")
        return [dict(zip(keys, (int(x) for x in line.split()))) for line in out.stdout.splitlines()]
    return call


def covered_once(blocks, ntiles, lo, hi):
    for t in range(ntiles):
        planes = sorted(p for b in blocks if b["tile"] == t for p in range(b["zs"], b["ze"]))
        assert planes == list(range(lo, hi)), f"tile {t}: planes {planes[:5]}.. of [{lo}, {hi})"


@pytest.mark.parametrize("ntx,nty,lo,hi,zc,swz", [(4, 32, 0, 512, 256, 1), (3, 24, 3, 387, 55, 1), (1, 1, 0, 7, 3, 0), (2, 5, 3, 20, 100, 1),
                                                  (8, 64, 3, 131, 128, 1)])
def test_plain_schedule_covers_every_plane_of_every_tile_once(sched, ntx, nty, lo, hi, zc, swz):
    b = sched(0, ntx, nty, lo, hi, zc, swz)
    covered_once(b, ntx * nty, lo, hi)
    assert all(x["down"] == 0 and x["bump"] == -1 and x["wait_late"] == -1 for x in b)
    if swz:   # XCD-contiguous: the tiles dispatch slots b, b+8, b+16, ... (one XCD) work on are consecutive
        ids = [x["tile"] + (x["zs"] - lo) // zc * ntx * nty for x in b]
        assert sorted(ids) == list(range(len(b)))
        per_xcd = ids[0::8]
        assert per_xcd == list(range(per_xcd[0], per_xcd[0] + len(per_xcd)))


def test_mixed_schedule_long_columns_then_short_pieces(sched):
    b = sched(1, 8, 64, 6, 125, 4)
    covered_once(b, 512, 6, 125)
    n_long = sum(1 for x in b if (x["zs"], x["ze"]) == (6, 125))
    assert n_long == 512 - 32 and all((x["zs"], x["ze"]) == (6, 125) for x in b[:n_long])


@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("need", [(1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("ntx,nty,lo,nzl,nshort,nsub,layout", [(8, 64, 3, 128, 32, 4, 0), (8, 64, 3, 128, 32, 4, 1), (8, 64, 3, 128, 32, 4, 2),
                                                              (2, 3, 3, 16, 1, 2, 0), (1, 2, 3, 7, 0, 4, 0), (3, 5, 3, 37, 2, 3, 1),
                                                              (2, 2, 3, 5, 1, 2, 0)])
def test_halves_schedule_invariants(sched, ntx, nty, lo, nzl, nshort, nsub, layout, first, need):
    """what the engine's single-launch pass relies on (wafer_engine.hip launch_halves_pass): both halves cover their planes
    once; half A marches down to the lower boundary, half B up to the upper one; exactly one piece per tile and half stores
    the boundary, counts itself done and -- where that side has a neighbour -- waits for the ghost flag at the iteration
    whose prefetch first touches a ghost plane; the half named `first` is dispatched first"""
    hi, mid, depth, ntiles = lo + nzl, lo + nzl // 2, 3, ntx * nty
    b = sched(2, ntx, nty, lo, hi, mid, first, need[0], need[1], nshort, nsub, depth, layout)
    covered_once(b, ntiles, lo, hi)
    thin = mid - lo < depth or hi - mid < depth
    for x in b:
        half = 0 if x["down"] else 1
        assert (lo <= x["zs"] < x["ze"] <= mid) if half == 0 else (mid <= x["zs"] < x["ze"] <= hi)
        at_boundary = x["zs"] == lo if half == 0 else x["ze"] == hi
        assert (x["bump"] == half) == at_boundary and (x["bump"] in (-1, half))
        if at_boundary:
            n = x["ze"] - x["zs"]
            assert x["wt"] == (n if (thin or depth > n) else depth)
            if need[half]:
                assert x["wait_late"] == half
                # marching z1 = ze + 1 - it (down) / zs - 2 + it (up), the prefetch reads plane z -+ 2: first ghost plane at
                assert x["wait_it"] == (x["ze"] - lo if half == 0 else hi - x["zs"])
                assert 0 <= x["wait_it"] < (x["ze"] - x["zs"]) + 4
            else:
                assert x["wait_late"] == -1
        else:
            assert x["wait_late"] == -1 and x["wt"] == 0
    for half in (0, 1):
        assert sum(1 for x in b if x["bump"] == half) == ntiles
    halves_in_order = [0 if x["down"] else 1 for x in b]
    assert halves_in_order[0] == first and halves_in_order == sorted(halves_in_order, reverse=bool(first))


def test_committed_counter_figures_name_the_kernel_sources_they_were_measured_on():
    """profiles/pmc_traffic.json carries the hash of the kernel sources it was measured on (tools/pmc_summary.py,
    wafer_amd/provenance.py); bench.py's roofline.traffic is labelled stale when the kernels have changed since"""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from wafer_amd import provenance
    doc = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert len(doc["kernel_sources_sha16"]) == 16
    files = [os.path.basename(f) for f in provenance.kernel_sources()]
    assert "wafer_stencil_fused3.hip.h" in files and "wafer_tu_fused3.hip" in files and "wafer_engine.hip" not in files
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    traffic, key, stale = bench.pmc_traffic("wafer_k_step3_fused<double, double, true, 0, true, 1>")
    assert traffic > 3e9 and key.startswith("void wafer_k_step3_fused<double, double, true, 0, true, 1>(")
    assert stale == (doc["kernel_sources_sha16"] != provenance.kernel_sources_sha16())
    assert bench.pmc_traffic("wafer_k_step3_fused") == (None, None, None)          # a family name matches nothing
