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
