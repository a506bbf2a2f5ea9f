"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU builds ("compile-time
-fsanitize=address on the host oracle"; GPU sanitizers are not available on this pool):

  * the oracle's C restatement -- every entry point on small awkward grids (oracle/sanitize_driver.c,
    `make -C oracle asan`);
  * the host driver's parsers and writers (wafer_cli.cpp's YAML-subset reader, wafer_files.h's five array
    formats): the tests of tests/test_files.py and the CPU tests of tests/test_cli.py re-run against a
    binary built with -fsanitize=address,undefined.  A sanitizer report exits with code 99, which none of those
    tests accepts.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:exitcode=99:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:exitcode=99:print_stacktrace=1"}


def test_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_division_plan_under_asan_ubsan(tmp_path):
    """wafer_divplan.h: the enumeration's 128-bit integer arithmetic and shifts, awkward divisors (zero, infinite, NaN, subnormal, the
    ends of the range, negative, powers of two) and 2000 random ones; every checked plan holds on its own candidates and the extra
    Markstein round is right on every candidate whatever the plan found"""
    out = str(tmp_path / "divplan-asan")
    cmd = ["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
           "-std=c++17", os.path.join(ROOT, "tests", "divplan_sanitize_driver.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([out], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=600)
    assert r.returncode == 0 and "DIVPLAN-OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.fixture(scope="module")
def asan_cli(tmp_path_factory):
    from wafer_amd import build
    lib = build.LIB
    if not os.path.exists(lib):
        build.build()
    out = str(tmp_path_factory.mktemp("asan") / "wafer-hip-asan")
    src = os.path.join(build.CSRC, "wafer_cli.cpp")
    cmd = ["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17",
           "-pthread", src, "-o", out, "-L", build.HERE, "-lwafer_hip", f"-Wl,-rpath,{build.HERE}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def test_cli_parsers_and_file_formats_under_asan_ubsan(asan_cli):
    # a sanitizer report must not pass for an expected parse failure (those exit with 1)
    env = dict(os.environ, WAFER_CLI_BIN=asan_cli, **SAN_ENV)
    r = subprocess.run([asan_cli, "--sanitize", " $//Project*\\"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.strip() == "_,36,,47,,47,Project,42,,92,", r.stderr   # output.rs:758-762
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_files.py"), os.path.join(ROOT, "tests", "test_cli.py"),
                        "-m", "not gpu", "-q", "-x", "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "passed" in r.stdout and "failed" not in r.stdout
