"""The host driver `wafer-hip` (wafer_amd/csrc/wafer_cli.cpp): configuration
reading / validation on CPU; the full run (table, summary, observables_N,
wavefunction_N, potential files) against the oracle on the GPU."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.environ.get("WAFER_CLI_BIN") or os.path.join(ROOT, "wafer_amd", "wafer-hip")   # (tests/test_sanitizers.py: the ASan build)
CASE = os.path.join(ROOT, "tests", "golden", "cli_case.yaml")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        from wafer_amd import build
        build.build()
    return CLI


def run(cli, *args, cwd=None):
    return subprocess.run([cli, *args], capture_output=True, text=True, cwd=cwd)


def test_sanitize_string_reference_vector(cli, ref_vectors):
    g = ref_vectors["directory_string"]  # output.rs:758-762
    assert run(cli, "--sanitize", g["input"]).stdout.rstrip("\n") == g["expected"]


def test_config_echo(cli):
    r = run(cli, "-c", CASE, "--check-config")
    assert r.returncode == 0, r.stderr
    c = json.loads(r.stdout)
    assert (c["nx"], c["ny"], c["nz"]) == (24, 20, 28) and c["project_name"] == "cli test #1"
    assert c["dn"] == 0.5 and c["dt"] == 0.04 and c["tolerance"] == 1e-7 and c["max_steps"] == 200000
    assert c["central_difference"] == 1 and c["potential"] == "Harmonic" and c["init_condition"] == "Boolean"
    assert c["screen_update"] == 100 and c["snap_update"] is None and c["file_type"] == "Csv"
    assert c["save_wavefns"] is True and c["wavemax"] == 1


@pytest.mark.parametrize("edit,msg", [
    (("dt: 0.04", "dt: 0.09"), "LargeDt"),                      # config.rs:363
    (("wavenum: 0", "wavenum: 3"), "LargeWavenum"),             # config.rs:366
    (("potential: Harmonic", "potential: Yukawa"), "unknown potential"),
    (("mass: 1.0\n", ""), "missing field `mass`"),
    (("central_difference: ThreePoint", "central_difference: NinePoint"), "central_difference"),
    (("init_symmetry: NotConstrained", "init_symmetry: Sideways"), "init_symmetry"),
])
def test_config_errors(cli, tmp_path, edit, msg):
    text = open(CASE).read()
    assert edit[0] in text
    bad = tmp_path / "bad.yaml"
    bad.write_text(text.replace(edit[0], edit[1]))
    r = run(cli, "-c", str(bad), "--check-config")
    assert r.returncode == 1 and msg in r.stderr


def test_missing_config_file(cli):
    r = run(cli, "-c", "/nonexistent/wafer.yaml", "--check-config")
    assert r.returncode == 1 and "ConfigLoad" in r.stderr


# ---- Rust std::fmt restated for the expected table rows (output.rs:497-521) ----
def rust_exp(v, prec):
    m, e = f"{v:.{prec}e}".split("e")
    return f"{m}e{int(e)}"


def table_row(tau, diff, energy, r_rms):
    spacer = " " * ((100 - 69) // 2)
    last = f"{rust_exp(diff, 5):>15} │" if tau > 0 else f"{'--   ':>15} │"
    return f"{spacer}│{tau:>11.3f} │{rust_exp(energy, 10):>19} │{r_rms:15.5f} │{last}"


@pytest.mark.gpu
def test_full_run_matches_oracle(cli, tmp_path):
    from oracle import wafer_oracle as wo
    r = run(cli, "-c", CASE, "--progress", "--output-dir", str(tmp_path / "out"), "--input-dir", str(tmp_path / "in"))
    assert r.returncode == 0, r.stderr
    out_dirs = os.listdir(tmp_path / "out")
    assert len(out_dirs) == 1 and out_dirs[0].startswith("cli_test_,35,1_")   # sanitize_string
    od = tmp_path / "out" / out_dirs[0]
    assert sorted(os.listdir(od)) == ["cli_case.yaml", "observables_0.csv", "observables_1.csv", "potential.csv",
                                      "wavefunction_0.csv", "wavefunction_1.csv"]

    wo.set_threads(4)   # a 24x20x28 grid: hundreds of OpenMP threads would only spin
    cfg = wo.Config(24, 20, 28, ext=1, potential="Harmonic", dn=0.5, dt=0.04, mass=1.0)
    v = wo.potential_generate(cfg)
    a, b = wo.ab(cfg, v)
    phi = wo.initial_condition(cfg, "Boolean")
    want, conv = wo.solve(cfg, 0, v, a, b, phi, [], 1e-7, 100, max_steps=200000)
    assert conv
    # ground state: every row of the table, character for character except the last digits of the sums
    rows = [l for l in r.stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]
    ground_rows = rows[:len(want)]
    for line, w in zip(ground_rows, want):
        exp = table_row(w["tau"], w["diff"], w["energy"] / w["norm2"], np.sqrt(w["r2"] / w["norm2"]))
        got_cols = [c.strip() for c in line.split("│")[1:5]]
        exp_cols = [c.strip() for c in exp.split("│")[1:5]]
        assert got_cols[0] == exp_cols[0]                                   # tau
        assert float(got_cols[1]) == pytest.approx(float(exp_cols[1]), abs=2e-9)
        assert got_cols[2] == exp_cols[2]                                   # r_rms to 5 decimals
        assert len(line) == len(exp)                                        # same column geometry
    assert "Ground state caclulation" in r.stdout and "1st excited state caclulation" in r.stdout
    e0 = want[-1]["energy"] / want[-1]["norm2"]
    m = re.search(r"══▶ Ground state energy = ([0-9.eE+-]+)", r.stdout)
    assert float(m.group(1)) == pytest.approx(e0, abs=2e-9)
    # observables_0.csv: header + one record (output.rs:32-45, 619-632)
    lines = open(od / "observables_0.csv").read().splitlines()
    assert lines[0] == "state,energy,binding_energy,r,l_r"
    rec = lines[1].split(",")
    assert rec[0] == "0" and float(rec[1]) == pytest.approx(e0, abs=2e-9)
    assert float(rec[4]) == pytest.approx(24 / float(rec[3]), rel=1e-12)    # l_r = Nx / r
    rec1 = open(od / "observables_1.csv").read().splitlines()[1].split(",")
    # the excited state starts from a clone of the ground state (grid.rs:95) that Gram-Schmidt
    # reduces to rounding noise; which level it settles on first (2.5, or the 3.5 plateau of the
    # ground state's own symmetry sector) is not comparable across implementations
    assert rec1[0] == "1" and e0 + 0.5 < float(rec1[1]) < 4.0
    # potential.csv: i,j,k,data over the WORK area, C order, the oracle's values exactly
    pot = np.loadtxt(od / "potential.csv", delimiter=",")
    assert pot.shape == (24 * 20 * 28, 4)
    assert np.array_equal(pot[:, 3].reshape(24, 20, 28), v[1:-1, 1:-1, 1:-1])
    assert np.array_equal(pot[:5, :3], [[0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 0, 3], [0, 0, 4]])
    # wavefunction_0.csv: the converged, normalised ground state (up to the oracle's trajectory)
    wf = np.loadtxt(od / "wavefunction_0.csv", delimiter=",")[:, 3].reshape(24, 20, 28)
    assert np.sum(wf * wf) == pytest.approx(1.0, abs=1e-12)
    assert np.allclose(wf, phi[1:-1, 1:-1, 1:-1], rtol=0, atol=1e-12)

    # restart: the saved ground state as ./input/wavefunction_0.csv, wavenum = 1 (grid.rs:35-39)
    (tmp_path / "in").mkdir()
    os.replace(od / "wavefunction_0.csv", tmp_path / "in" / "wavefunction_0.csv")
    restart = tmp_path / "restart.yaml"
    restart.write_text(open(CASE).read().replace("wavenum: 0", "wavenum: 1"))
    r2 = run(cli, "-c", str(restart), "--output-dir", str(tmp_path / "out2"), "--input-dir", str(tmp_path / "in"))
    assert r2.returncode == 0, r2.stderr
    assert "Ground state" not in r2.stdout and "1st excited state caclulation" in r2.stdout
    m = re.search(r"══▶ 1st excited state energy = ([0-9.eE+-]+)", r2.stdout)
    assert e0 + 0.5 < float(m.group(1)) < 4.0


@pytest.mark.gpu
def test_restart_from_lower_resolution(cli, tmp_path):
    """the reference's `FromFile` workflow (config.rs:153-160, input.rs:149-176): a converged
    low-resolution state in ./input is trilinearly upsampled and converges in fewer blocks"""
    text = open(CASE).read()
    coarse = text.replace("x: 24", "x: 12").replace("y: 20", "y: 10").replace("z: 28", "z: 14") \
                 .replace("dn: 0.5", "dn: 1.0").replace("wavemax: 1", "wavemax: 0")
    (tmp_path / "coarse.yaml").write_text(coarse)
    r = run(cli, "-c", str(tmp_path / "coarse.yaml"), "--output-dir", str(tmp_path / "o1"), "--input-dir", str(tmp_path / "none"))
    assert r.returncode == 0, r.stderr
    od = tmp_path / "o1" / os.listdir(tmp_path / "o1")[0]
    (tmp_path / "in").mkdir()
    os.replace(od / "wavefunction_0.csv", tmp_path / "in" / "wavefunction_0.csv")

    def blocks(stdout):
        return len([l for l in stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)])

    fine = text.replace("wavemax: 1", "wavemax: 0").replace("screen_update: 100", "screen_update: 10")
    (tmp_path / "cold.yaml").write_text(fine)
    (tmp_path / "warm.yaml").write_text(fine.replace("init_condition: Boolean", "init_condition: FromFile"))
    cold = run(cli, "-c", str(tmp_path / "cold.yaml"), "--progress", "--output-dir", str(tmp_path / "o2"), "--input-dir", str(tmp_path / "none"))
    warm = run(cli, "-c", str(tmp_path / "warm.yaml"), "--progress", "--output-dir", str(tmp_path / "o3"), "--input-dir", str(tmp_path / "in"))
    assert cold.returncode == 0 and warm.returncode == 0, warm.stderr
    assert "Interpolating from [14, 12, 16] to requested size of [26, 22, 30]" in warm.stderr
    e = lambda out: float(re.search(r"Ground state energy = ([0-9.eE+-]+)", out).group(1))
    assert e(warm.stdout) == pytest.approx(e(cold.stdout), abs=1e-5)
    assert blocks(warm.stdout) > 1 and blocks(cold.stdout) > 1


@pytest.mark.gpu
def test_messagepack_run_and_restart(cli, tmp_path):
    """file_type: Messagepack end to end (output.rs:172-181, 606-617; input.rs:113-147): outputs
    decode with an independent msgpack reader, and the saved state restarts a run from ./input"""
    import msgpack
    from oracle import wafer_oracle as wo
    text = open(CASE).read().replace("file_type: Csv", "file_type: Messagepack").replace("wavemax: 1", "wavemax: 0")
    (tmp_path / "mpk.yaml").write_text(text)
    r = run(cli, "-c", str(tmp_path / "mpk.yaml"), "--output-dir", str(tmp_path / "out"), "--input-dir", str(tmp_path / "none"))
    assert r.returncode == 0, r.stderr
    od = tmp_path / "out" / os.listdir(tmp_path / "out")[0]
    assert sorted(os.listdir(od)) == ["mpk.yaml", "observables_0.mpk", "potential.mpk", "wavefunction_0.mpk"]
    state, energy, binding, r_rms, l_r = msgpack.unpackb(open(od / "observables_0.mpk", "rb").read())
    e0 = float(re.search(r"Ground state energy = ([0-9.eE+-]+)", r.stdout).group(1))
    assert state == 0 and energy == pytest.approx(e0, abs=1e-9) and l_r == pytest.approx(24 / r_rms, rel=1e-12)
    assert binding == energy                                     # Harmonic: pot_sub is zero
    v, dim, data = msgpack.unpackb(open(od / "potential.mpk", "rb").read())
    wo.set_threads(4)
    cfg = wo.Config(24, 20, 28, ext=1, potential="Harmonic", dn=0.5, dt=0.04, mass=1.0)
    assert v == 1 and dim == [24, 20, 28]
    assert np.array_equal(np.array(data).reshape(24, 20, 28), wo.potential_generate(cfg)[1:-1, 1:-1, 1:-1])
    v, dim, data = msgpack.unpackb(open(od / "wavefunction_0.mpk", "rb").read())
    assert dim == [24, 20, 28] and np.sum(np.square(data)) == pytest.approx(1.0, abs=1e-12)

    # restart from the .mpk state: already converged, so the second observation ends the run
    (tmp_path / "in").mkdir()
    os.replace(od / "wavefunction_0.mpk", tmp_path / "in" / "wavefunction_0.mpk")
    (tmp_path / "warm.yaml").write_text(text.replace("init_condition: Boolean", "init_condition: FromFile"))
    r2 = run(cli, "-c", str(tmp_path / "warm.yaml"), "--progress", "--output-dir", str(tmp_path / "out2"), "--input-dir", str(tmp_path / "in"))
    assert r2.returncode == 0, r2.stderr
    rows = [l for l in r2.stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]
    assert len(rows) == 2
    assert float(re.search(r"Ground state energy = ([0-9.eE+-]+)", r2.stdout).group(1)) == pytest.approx(e0, abs=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,body", [("json", '{"pot_sub": 2.0}'), ("yaml", "---\npot_sub: 2.0\n"), ("csv", "2\n")])
def test_potential_sub_file_overrides_the_computed_value(cli, tmp_path, fmt, body):
    """potential.rs:113-131: ./input/potential_sub.* wins for every potential type; a singular
    value shifts binding_energy = (energy - v_infinity) / norm2 by exactly that value"""
    (tmp_path / "in").mkdir()
    (tmp_path / "in" / f"potential_sub.{fmt}").write_text(body)
    text = open(CASE).read().replace("wavemax: 1", "wavemax: 0").replace("save_wavefns: true", "save_wavefns: false")
    (tmp_path / "c.yaml").write_text(text)
    r = run(cli, "-c", str(tmp_path / "c.yaml"), "--output-dir", str(tmp_path / "out"), "--input-dir", str(tmp_path / "in"))
    assert r.returncode == 0, r.stderr
    assert "Potential_sub loaded from disk" in r.stderr
    od = tmp_path / "out" / os.listdir(tmp_path / "out")[0]
    rec = open(od / "observables_0.csv").read().splitlines()[1].split(",")
    assert float(rec[2]) == pytest.approx(float(rec[1]) - 2.0, abs=1e-12)
    assert open(od / "potential_sub.csv").read().strip() == "2"          # output::potential_sub, singular value
    # an ARRAY for a potential whose pot_sub is singular is the reference's WrongPotentialSubDims error
    (tmp_path / "in" / f"potential_sub.{fmt}").unlink()
    (tmp_path / "in" / "potential_sub.json").write_text('{"v":1,"dim":[1,1,2],"data":[1.0,2.0]}')
    r = run(cli, "-c", str(tmp_path / "c.yaml"), "--output-dir", str(tmp_path / "out3"), "--input-dir", str(tmp_path / "in"))
    assert r.returncode == 1 and "WrongPotentialSubDims" in r.stderr


@pytest.mark.gpu
def test_potential_sub_array_is_resampled_to_the_grid(cli, tmp_path, oracle):
    """input::fill_sub_data (input.rs:453-478): a potential_sub ARRAY of another size is
    interpolated to (nx, ny, nz) with trilerp_resize; the driver writes back what the run used
    (output::potential_sub), which must be the oracle's resample of the file, digit for digit"""
    (tmp_path / "in").mkdir()
    src = np.random.default_rng(8).uniform(0.5, 2.0, (5, 4, 6))
    (tmp_path / "in" / "potential_sub.json").write_text(json.dumps({"v": 1, "dim": list(src.shape), "data": src.ravel().tolist()}))
    text = (open(CASE).read().replace("wavemax: 1", "wavemax: 0").replace("save_wavefns: true", "save_wavefns: false")
            .replace("potential: Harmonic", "potential: FullCornell").replace("tolerance: 1e-7", "tolerance: 1e-3"))
    (tmp_path / "c.yaml").write_text(text)
    r = run(cli, "-c", str(tmp_path / "c.yaml"), "--output-dir", str(tmp_path / "out"), "--input-dir", str(tmp_path / "in"))
    assert r.returncode == 0, r.stderr
    assert "Interpolating potential_sub from [5, 4, 6] to requested size of [24, 20, 28]." in r.stderr
    od = tmp_path / "out" / os.listdir(tmp_path / "out")[0]
    got = np.loadtxt(od / "potential_sub.csv", delimiter=",")[:, 3].reshape(24, 20, 28)
    assert np.array_equal(got, oracle.trilerp_resize(src, (24, 20, 28)))


SCRIPT = """#!/usr/bin/env python3
import json, sys
g = json.load(sys.stdin)["grid"]
assert list(g) == ["dn", "x", "y", "z"]          # serde_json's (sorted) key order
for i in range(g["x"]):
    for j in range(g["y"]):
        for k in range(g["z"]):
            r2 = sum(((q + 1) * g["dn"] - g["dn"] * (n + 1) / 2.) ** 2 for q, n in ((i, g["x"]), (j, g["y"]), (k, g["z"])))
            print(%s)
"""


def write_script(path, expr):
    path.write_text(SCRIPT % expr)
    path.chmod(0o755)


@pytest.mark.parametrize("expr,msg", [
    ('"abc"', "ParseFloat"),                      # a line that is not a float (input.rs:224-227)
    ('0.5 * r2 if i else ""', "ParseFloat"),      # empty lines do not parse either
    ('*([0.5 * r2] if i else []), end=chr(10) if i else ""', "ArrayShape"),   # too few values (input.rs:229-231)
])
def test_script_potential_errors(cli, tmp_path, expr, msg):
    """potential: FromScript (input.rs:186-246) fails like the reference does -- before any GPU work,
    so these run on the CPU box"""
    text = open(CASE).read().replace("potential: Harmonic", "potential: FromScript")
    (tmp_path / "c.yaml").write_text(text)
    r = run(cli, "-c", "c.yaml", cwd=tmp_path)
    assert r.returncode == 1 and "SpawnPython" in r.stderr, r.stderr          # no ./gen_potential.py here
    write_script(tmp_path / "pot.py", expr)
    r = run(cli, "-c", "c.yaml", "-s", "pot.py", cwd=tmp_path)
    assert r.returncode == 1 and "LoadPotential" in r.stderr and msg in r.stderr, r.stderr


@pytest.mark.gpu
def test_script_potential_runs(cli, tmp_path):
    """the script's values become the work area of V; the same numbers supplied as a
    potential FILE give the same run, row for row"""
    text = (open(CASE).read().replace("potential: Harmonic", "potential: FromScript").replace("wavemax: 1", "wavemax: 0")
            .replace("save_wavefns: true", "save_wavefns: false"))
    (tmp_path / "c.yaml").write_text(text)
    write_script(tmp_path / "gen_potential.py", "repr(0.5 * r2)")
    r = run(cli, "-c", "c.yaml", cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Generating potential from script file: ./gen_potential.py" in r.stderr
    od = tmp_path / "output" / os.listdir(tmp_path / "output")[0]
    pot = np.loadtxt(od / "potential.csv", delimiter=",")[:, 3].reshape(24, 20, 28)   # output::potential: the work area
    dn = 0.5
    ax = [((np.arange(n) + 1) * dn - dn * (n + 1) / 2.) ** 2 for n in (24, 20, 28)]
    want = np.zeros((24, 20, 28))
    for i in range(24):       # the script's own summation order
        for j in range(20):
            want[i, j, :] = [0.5 * sum((ax[0][i], ax[1][j], ax[2][k])) for k in range(28)]
    assert np.array_equal(pot, want)
    # the same array as ./input/potential.csv with potential: FromFile
    (tmp_path / "input").mkdir(exist_ok=True)
    with open(tmp_path / "input" / "potential.csv", "w") as f:
        for i in range(24):
            for j in range(20):
                for k in range(28):
                    f.write(f"{i},{j},{k},{float(want[i, j, k])!r}\n")
    (tmp_path / "f.yaml").write_text(text.replace("potential: FromScript", "potential: FromFile"))
    r2 = run(cli, "-c", "f.yaml", "--output-dir", "out2", cwd=tmp_path)
    assert r2.returncode == 0, r2.stderr
    rows = lambda t: [l for l in t.splitlines() if "│" in l]
    assert rows(r.stdout) == rows(r2.stdout) and len(rows(r.stdout)) >= 2


def test_symmetry_needs_seven_point(cli, tmp_path):
    """config.rs:702-725 walks n + 6 cells: with a narrower frame the reference panics on an
    out-of-bounds index; the driver says so before touching the GPU"""
    bad = tmp_path / "sym.yaml"
    bad.write_text(open(CASE).read().replace("init_symmetry: NotConstrained", "init_symmetry: AboutZ"))
    r = run(cli, "-c", str(bad), "--output-dir", str(tmp_path / "o"))
    assert r.returncode == 1 and "SevenPoint" in r.stderr


@pytest.mark.gpu
def test_symmetry_constraint_is_applied_to_the_start(cli, tmp_path):
    """init_symmetry: AntisymAboutZ on a Constant start (config.rs:625).  The first table row is
    the energy of the constrained start -- the oracle's, to the last digits of the sums.  (The
    reference's mirror plane sits half a cell off the potential's centre and it constrains only
    the start and the snapshots, so the run still relaxes to the even ground state: 1.5.)"""
    from oracle import wafer_oracle as wo
    text = open(CASE).read().replace("central_difference: ThreePoint", "central_difference: SevenPoint") \
        .replace("init_symmetry: NotConstrained", "init_symmetry: AntisymAboutZ") \
        .replace("init_condition: Boolean", "init_condition: Constant") \
        .replace("wavemax: 1", "wavemax: 0").replace("dt: 0.04", "dt: 0.02").replace("tolerance: 1e-7", "tolerance: 1e-5") \
        .replace("x: 24", "x: 20").replace("z: 28", "z: 25")
    (tmp_path / "sym.yaml").write_text(text)
    r = run(cli, "-c", str(tmp_path / "sym.yaml"), "--progress", "--output-dir", str(tmp_path / "out"), "--input-dir", str(tmp_path / "none"))
    assert r.returncode == 0, r.stderr
    wo.set_threads(4)
    cfg = wo.Config(20, 20, 25, ext=3, potential="Harmonic", dn=0.5, dt=0.02, mass=1.0)
    v = wo.potential_generate(cfg)
    phi = wo.initial_condition(cfg, "Constant")
    plain = wo.observables(cfg, v, phi)
    wo.symmetrise(cfg, "AntisymAboutZ", phi)
    o = wo.observables(cfg, v, phi)
    rows = [l for l in r.stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]
    first = [c.strip() for c in rows[0].split("│")[1:5]]
    assert float(first[0]) == 0.0 and float(first[1]) == pytest.approx(o["energy"] / o["norm2"], rel=1e-9)
    assert abs(o["energy"] / o["norm2"] - plain["energy"] / plain["norm2"]) > 0.1     # the constraint did something
    e = float(re.search(r"Ground state energy = ([0-9.eE+-]+)", r.stdout).group(1))
    assert e == pytest.approx(1.5, abs=0.05)


@pytest.mark.gpu
def test_snapshots_and_resume_from_partial(cli, tmp_path):
    """snap_update (grid.rs:137-158): a `_partial` file per snapshot, written by a writer thread
    while evolve continues; it is what is left when max_steps ends the run (grid.rs:223-245), is
    removed on convergence, and restarts the state from ./input (input.rs:513-523)"""
    text = open(CASE).read().replace("wavemax: 1", "wavemax: 0").replace("# snap_update: 1000", "snap_update: 100")
    short = text.replace("max_steps: 200000", "max_steps: 150")
    (tmp_path / "short.yaml").write_text(short)
    r = run(cli, "-c", str(tmp_path / "short.yaml"), "--output-dir", str(tmp_path / "o1"), "--input-dir", str(tmp_path / "none"))
    assert r.returncode == 1 and "MaxStep" in r.stderr
    od = tmp_path / "o1" / os.listdir(tmp_path / "o1")[0]
    assert "wavefunction_0_partial.csv" in os.listdir(od) and "wavefunction_0.csv" not in os.listdir(od)
    assert not [n for n in os.listdir(od) if n.endswith(".tmp")]
    part = np.loadtxt(od / "wavefunction_0_partial.csv", delimiter=",")[:, 3]
    assert np.sum(part * part) == pytest.approx(1.0, abs=1e-12)          # normalised ONCE (not the reference's twice)

    # resume: the partial state in ./input starts wavefunction 0 (FromFile), converges, and the
    # run's own partial file is removed at the end
    (tmp_path / "in").mkdir()
    os.replace(od / "wavefunction_0_partial.csv", tmp_path / "in" / "wavefunction_0_partial.csv")
    (tmp_path / "resume.yaml").write_text(text.replace("init_condition: Boolean", "init_condition: FromFile"))
    r2 = run(cli, "-c", str(tmp_path / "resume.yaml"), "--progress", "--output-dir", str(tmp_path / "o2"), "--input-dir", str(tmp_path / "in"))
    assert r2.returncode == 0, r2.stderr
    od2 = tmp_path / "o2" / os.listdir(tmp_path / "o2")[0]
    assert "wavefunction_0.csv" in os.listdir(od2) and "wavefunction_0_partial.csv" not in os.listdir(od2)
    rows2 = [l for l in r2.stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]
    (tmp_path / "full.yaml").write_text(text)
    r3 = run(cli, "-c", str(tmp_path / "full.yaml"), "--progress", "--output-dir", str(tmp_path / "o3"), "--input-dir", str(tmp_path / "none"))
    rows3 = [l for l in r3.stdout.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]
    assert r3.returncode == 0 and 1 < len(rows2) < len(rows3)            # the resumed run had a head start
    e = lambda out: float(re.search(r"Ground state energy = ([0-9.eE+-]+)", out).group(1))
    assert e(r2.stdout) == pytest.approx(e(r3.stdout), abs=1e-6)
