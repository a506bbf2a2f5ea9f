"""The multi-process z-slab path with REAL processes and a real process group on the GPU box:
`python -m torch.distributed.run --nproc-per-node 2/3` ranks share the one GPU, the engine's
halo / all-reduce hooks are served by wafer_amd.slab.HostStagedSlabComm (gloo + host
staging; RCCL refuses several ranks per device, and the box has one).  What RCCL would add on an
8-GPU node is the transport only -- hooks, stream ordering, slab bookkeeping, overlap and the
launcher environment are the ones exercised here."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(nproc, script, *args, timeout=420):
    """script: a path, or "-m" followed by a module name in args"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(29000 + 17 * nproc + os.getpid() % 500),
           script, *args]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_slab_ranks_as_processes(world):
    r = launch(world, os.path.join(ROOT, "tests", "mp_slab_worker.py"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"MP-OK {world}" in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_device_side_allreduce_between_processes(world):
    """include/wafer_mailbox.h: mailboxes mapped between real processes through HIP IPC (on the box's one GPU, as over
    xGMI), the primitive bit for bit against the host's sum, excited-state steps on slabs within 1e-12 of the run whose
    all-reduce goes through gloo, and the microseconds one call takes"""
    r = launch(world, os.path.join(ROOT, "tests", "mailbox_worker.py"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"MAILBOX-OK {world}" in r.stdout
    us = float(r.stdout.split("us_per_allreduce=")[1].split()[0])
    assert 0.0 < us < 2000.0, us
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"mailbox_latency_world{world}.txt"), "w") as f:
            f.write(f"wafer_mailbox_allreduce, {world} processes on one MI355X, 4 doubles, 200 calls back to back: {us:.2f} us per call\n")


def _check_bench_two_rank_line(d, peers=False):
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["steps"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    ho = d["config"]["halo_overlap"]      # the exchange schedules were tried during set-up, one was chosen for all ranks
    assert ho["mode"] in (0, 1, 2, 3, 4, 5, 6) and len(ho["ms_per_step"]) == (11 if peers else 5) and all(v > 0 for v in ho["ms_per_step"].values())
    # the peer-store and peer-copy schedules (HIP IPC between the rank processes) were connected, timed and survived their bounded waits
    assert ("3_single_launch_peer_stores" in ho["ms_per_step"]) == peers and ("4_single_launch_peer_copies" in ho["ms_per_step"]) == peers
    if peers:   # ... after they had reproduced an exchange's bits on every rank (slab.overlap_modes_agree)
        assert ho["peer_store_check"]["identical"] is True and ho["peer_copy_check"]["identical"] is True
        # ... the copies also under mode 1's and mode 0's launches (modes 5, 6: every reader starts after the copy)
        assert ho["peer_copy_check"]["identical_mode_5"] is True and ho["peer_copy_check"]["identical_mode_6"] is True
        assert "5_boundary_first_peer_copies" in ho["ms_per_step"] and "6_no_overlap_peer_copies" in ho["ms_per_step"]
    assert d["roofline"]["kernel"].startswith("wafer_k_step3_fused<double, double, ")
    assert ho["fused_passes_per_exchange"] in (1, 2)
    assert d["config"]["parallelism"] == "zslab2" and d["config"]["points_per_gpu"] == 256 * 256 * 128
    assert d["roofline"]["steps_per_launch"] == 3 and d["config"]["kernel"] == "wafer_k_step3_fused" and "cpu_baseline" not in d
    # the same grid undecomposed on rank 0's GPU: T1 in the same line, and every slab's bits against it
    ref = d["single_gpu_ref"]
    assert ref["grid"] == [256, 256, 256] and ref["ms_per_step"] > 0 and d["single_gpu_ref_ms_per_step"] == ref["ms_per_step"]
    par = d["parity"]
    assert par["identical"] is True and par["slabs"] == 2 and par["steps"] == 10 and par["differing_slabs"] == []
    assert par["initial_condition"]["norm2"] == par["initial_condition"]["closed_form"] == 128 * 128 * 128
    assert d["comm"]["process_group_ranks"] == 2 and d["comm"]["halo_overlap_mode"] == ho["mode"]


@pytest.mark.parametrize("peers", [True, False])
def test_bench_multi_rank_path(peers):
    """bench.py's N > 1 leg end to end (slab partition, hooks, max-over-ranks timing, the undecomposed
    reference run and the slab-by-slab checksum comparison, one JSON line from rank 0) with two ranks on
    the one GPU over the host-staged transport.  peers: WAFER_BENCH_PEERS=force -- every line the first real
    multi-GPU run will execute: the ranks map each other's buffers through HIP IPC, the peer-store schedule
    (overlap mode 3) is timed with the others in the set-up trial, and whichever wins runs the timed steps"""
    import json
    env_extra = {"WAFER_BENCH_TRANSPORT": "host", "WAFER_HV_WAIT_MS": "5000"}
    if peers:
        env_extra["WAFER_BENCH_PEERS"] = "force"
    os.environ.update(env_extra)
    try:
        r = launch(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                   "--grid", "256,256,256")
    finally:
        for k in env_extra:
            os.environ.pop(k, None)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _check_bench_two_rank_line(json.loads(lines[0]), peers)


def test_bench_eight_rank_path():
    """the driver's N = 8 call shape on the one GPU (host-staged transport): eight slabs of 32 planes, the set-up
    trial over all exchange schedules -- the peer-store one included (WAFER_BENCH_PEERS=force: eight processes mapping
    their z-neighbours' buffers through HIP IPC; the grid is kept at 16 workgroups per rank so that all eight ranks'
    kernels are resident together and a polling workgroup cannot keep its neighbour's kernel off the CUs) -- three-step
    passes with a two-step remainder, every slab's bits against the undecomposed run"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, WAFER_BENCH_TRANSPORT="host", WAFER_BENCH_PEERS="force", WAFER_HV_WAIT_MS="5000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "11", "--warmup", "3",
                        "--grid", "128,128,256"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["config"]["parallelism"] == "zslab8"
    ms = d["config"]["halo_overlap"]["ms_per_step"]
    assert d["config"]["points_per_gpu"] == 128 * 128 * 32 and len(ms) in (7, 9, 11) and "3_single_launch_peer_stores" in ms and "4_single_launch_peer_copies" in ms and "6_no_overlap_peer_copies" in ms
    assert d["config"]["halo_overlap"]["peer_store_check"]["identical"] is True and d["config"]["halo_overlap"]["peer_copy_check"]["identical"] is True
    assert d["parity"]["identical"] is True and d["parity"]["slabs"] == 8 and d["parity"]["differing_slabs"] == []
    assert d["single_gpu_ref"]["grid"] == [128, 128, 256] and d["comm"]["process_group_ranks"] == 8


def test_bench_bare_call_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks itself
    (before touching a GPU) and forwards rank 0's line"""
    import json
    env = dict(os.environ, WAFER_BENCH_TRANSPORT="host", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                        "--grid", "256,256,256"], capture_output=True, text=True, env=env, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _check_bench_two_rank_line(json.loads(lines[0]))


def test_bench_refuses_more_ranks_than_gpus():
    """a bare `--gpus 8` on a box with fewer GPUs must fail, not run a smaller job under the label"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WAFER_BENCH_TRANSPORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "{" not in r.stdout and "refusing" in r.stderr


@pytest.mark.parametrize("ext", [1, 2])
def test_rccl_transport_self_neighbours(ext):
    """RCCL send / recv / all-reduce on tensors aliasing the engine's HBM, under the engine's
    streams: one rank that is its own z-neighbour (tests/rccl_selfloop_worker.py)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + ext + os.getpid() % 300),
               HSA_ENABLE_IPC_MODE_LEGACY="0", WAFER_TEST_EXT=str(ext), PYTHONUNBUFFERED="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_selfloop_worker.py")],
                       capture_output=True, text=True, env=env, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL-OK" in r.stdout


@pytest.mark.parametrize("peers", [False, True, "copies"])
def test_multi_rank_solve_driver_matches_the_native_driver(tmp_path, peers):
    """python -m wafer_amd.run on 2 ranks (z-slabs, host-staged transport on the one GPU) against
    wafer-hip on the whole grid: same table rows, same energies, the saved planes tile the state.
    peers: WAFER_PEER_STORES=force -- the driver connects the z-neighbours (HIP IPC between the two processes) and its
    ground-state passes run in overlap mode 3; "copies": WAFER_PEER_STORES=copies -- every exchange of the whole solve (ground AND excited
    state) is a device copy into the neighbour process's ghost planes (overlap mode 4)"""
    import re
    import numpy as np
    case = os.path.join(ROOT, "tests", "golden", "cli_case.yaml")
    cli = os.path.join(ROOT, "wafer_amd", "wafer-hip")
    one = subprocess.run([cli, "-c", case, "--progress", "--output-dir", str(tmp_path / "one"), "--input-dir", str(tmp_path / "none")],
                         capture_output=True, text=True)
    assert one.returncode == 0, one.stderr
    saved = {k: os.environ.get(k) for k in ("WAFER_TRANSPORT", "WAFER_PEER_STORES", "WAFER_PEER_SAME_DEVICE")}
    os.environ["WAFER_TRANSPORT"] = "host"
    if peers:
        os.environ["WAFER_PEER_STORES"] = "force" if peers is True else "copies"
        os.environ["WAFER_PEER_SAME_DEVICE"] = "1"      # the two ranks share the box's one GPU on purpose
    try:
        two = launch(2, "-m", "wafer_amd.run", "-c", case, "--progress", "--output-dir", str(tmp_path / "two"))
    finally:
        for k, v in saved.items():      # (test_gpu_slab.py sets WAFER_PEER_SAME_DEVICE for its own module at import: leave it as found)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    assert ("halo schedule: overlap mode 3 (peer stores)" in two.stderr) == (peers is True)
    assert ("halo schedule: overlap mode 4 (peer copies)" in two.stderr) == (peers == "copies")

    def rows(text):
        return [l for l in text.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]

    def ground_rows(text):
        return rows(text.split("1st excited state caclulation")[0])

    a, b = ground_rows(one.stdout), ground_rows(two.stdout)
    assert len(a) == len(b) > 3
    for la, lb in zip(a, b):
        ca, cb = [c.strip() for c in la.split("│")[1:5]], [c.strip() for c in lb.split("│")[1:5]]
        assert ca[0] == cb[0] and ca[2] == cb[2] and len(la) == len(lb)          # tau, r_rms, geometry
        assert float(ca[1]) == pytest.approx(float(cb[1]), abs=2e-9)             # energy: sums associate differently
    header = [l for l in two.stdout.splitlines() if "Ground state caclulation" in l]
    assert header and header[0] in one.stdout                                    # the same header line, character for character
    e1 = float(re.search(r"Ground state energy = ([0-9.eE+-]+)", one.stdout).group(1))
    e2 = float(re.search(r"Ground state energy = ([0-9.eE+-]+)", two.stdout).group(1))
    assert e2 == pytest.approx(e1, abs=2e-9)
    od = tmp_path / "two" / os.listdir(tmp_path / "two")[0]
    names = sorted(os.listdir(od))
    assert "observables_0.csv" in names and "observables_1.json" in names
    lo, hi = np.load(od / "wavefunction_0_z0-14.npy"), np.load(od / "wavefunction_0_z14-28.npy")
    state = np.concatenate([lo, hi], axis=2)
    assert state.shape == (24, 20, 28) and np.sum(state * state) == pytest.approx(1.0, abs=1e-12)
    one_dir = tmp_path / "one" / os.listdir(tmp_path / "one")[0]
    want = np.loadtxt(one_dir / "wavefunction_0.csv", delimiter=",")[:, 3].reshape(24, 20, 28)
    assert np.allclose(state, want, rtol=0, atol=1e-12)


def test_multi_rank_driver_reads_input_files(tmp_path):
    """potential: FromFile + a potential_sub override + a restart at wavenum 1 (grid.rs:35-39) on 2
    ranks: rank 0 stages the reference-format files as framed .npy, every rank memory-maps them and
    uploads only its own planes.  Same rows as wafer-hip reading the same ./input on one GPU."""
    import re
    import shutil
    case = os.path.join(ROOT, "tests", "golden", "cli_case.yaml")
    cli = os.path.join(ROOT, "wafer_amd", "wafer-hip")
    first = subprocess.run([cli, "-c", case, "--output-dir", str(tmp_path / "first"), "--input-dir", str(tmp_path / "none")],
                           capture_output=True, text=True)
    assert first.returncode == 0, first.stderr
    od = tmp_path / "first" / os.listdir(tmp_path / "first")[0]
    inp = tmp_path / "input"
    inp.mkdir()
    shutil.copy(od / "potential.csv", inp / "potential.csv")
    shutil.copy(od / "wavefunction_0.csv", inp / "wavefunction_0.csv")
    (inp / "potential_sub.json").write_text('{"pot_sub": 2.0}')
    # a deterministic start for state 1 (a clone of state 0 leaves rounding noise, which differs between
    # one and two ranks): x times the ground state, as a snapshot file (input.rs:513-523)
    import numpy as np
    w0 = np.loadtxt(od / "wavefunction_0.csv", delimiter=",")
    with open(inp / "wavefunction_1_partial.csv", "w") as f:
        for i, j, k, v in w0:
            f.write(f"{int(i)},{int(j)},{int(k)},{float(v * (i - 11.5))!r}\n")
    text = (open(case).read().replace("potential: Harmonic", "potential: FromFile").replace("wavenum: 0", "wavenum: 1"))
    (tmp_path / "restart.yaml").write_text(text)
    one = subprocess.run([cli, "-c", str(tmp_path / "restart.yaml"), "--progress", "--output-dir", str(tmp_path / "one"),
                          "--input-dir", str(inp)], capture_output=True, text=True)
    assert one.returncode == 0, one.stderr
    os.environ["WAFER_TRANSPORT"] = "host"
    try:
        two = launch(2, "-m", "wafer_amd.run", "-c", str(tmp_path / "restart.yaml"), "--progress",
                     "--output-dir", str(tmp_path / "two"), "--input-dir", str(inp))
    finally:
        os.environ.pop("WAFER_TRANSPORT", None)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    assert sorted(os.listdir(inp / ".wafer_amd")) == ["potential.pad1.npy", "potential_sub.pad0.npy", "wavefunction_0.pad1.npy",
                                                     "wavefunction_1_partial.pad1.npy"]
    assert "Ground state" not in two.stdout and "1st excited state caclulation" in two.stdout

    def rows(text):
        return [l for l in text.splitlines() if re.match(r"^\s+│\s*[0-9.]+ │", l)]

    a, b = rows(one.stdout), rows(two.stdout)
    assert len(a) == len(b) > 3
    for la, lb in zip(a, b):
        ca, cb = [c.strip() for c in la.split("│")[1:5]], [c.strip() for c in lb.split("│")[1:5]]
        assert ca[0] == cb[0] and ca[2] == cb[2]
        assert float(ca[1]) == pytest.approx(float(cb[1]), abs=5e-9)
    e = [float(re.search(r"1st excited state energy = ([0-9.eE+-]+)", t).group(1)) for t in (one.stdout, two.stdout)]
    be = [float(re.search(r"1st excited state binding energy = ([0-9.eE+-]+)", t).group(1)) for t in (one.stdout, two.stdout)]
    assert e[1] == pytest.approx(e[0], abs=5e-9)
    assert be[0] == pytest.approx(e[0] - 2.0, abs=1e-12) and be[1] == pytest.approx(e[1] - 2.0, abs=1e-12)
    # a ready-made framed .npy is used as it is (what one writes for grids too large for the text formats)
    pot = np.zeros((26, 22, 30))
    pot[1:-1, 1:-1, 1:-1] = np.loadtxt(od / "potential.csv", delimiter=",")[:, 3].reshape(24, 20, 28)
    np.save(inp / "potential.npy", pot)
    (inp / "potential.csv").unlink()
    (inp / ".wafer_amd" / "potential.pad1.npy").unlink()
    os.environ["WAFER_TRANSPORT"] = "host"
    try:
        again = launch(2, "-m", "wafer_amd.run", "-c", str(tmp_path / "restart.yaml"), "--output-dir", str(tmp_path / "again"),
                       "--input-dir", str(inp))
    finally:
        os.environ.pop("WAFER_TRANSPORT", None)
    assert again.returncode == 0, again.stdout[-2000:] + again.stderr[-4000:]
    assert not (inp / ".wafer_amd" / "potential.pad1.npy").exists()
    assert float(re.search(r"1st excited state energy = ([0-9.eE+-]+)", again.stdout).group(1)) == e[1]
    (inp / "potential.npy").unlink()
    # a wrong-sized array is refused with a pointer to the single-GPU driver, which resamples
    (inp / "potential.csv").write_text("0,0,0,1.0\n0,0,1,1.0\n")
    os.environ["WAFER_TRANSPORT"] = "host"
    try:
        bad = launch(2, "-m", "wafer_amd.run", "-c", str(tmp_path / "restart.yaml"), "--output-dir", str(tmp_path / "bad"),
                     "--input-dir", str(inp))
    finally:
        os.environ.pop("WAFER_TRANSPORT", None)
    assert bad.returncode != 0 and "resample it once with wafer-hip" in bad.stderr


def test_native_rccl_host_self_neighbours():
    """wafer-hip-slabs --self: the hooks served by RCCL's C API directly from a native host
    (no Python, no torch in that process), one rank that is its own z-neighbour, bit for bit
    against device copies (wafer_amd/csrc/wafer_rccl_host.cpp)"""
    exe = os.path.join(ROOT, "wafer_amd", "wafer-hip-slabs")
    if not os.path.exists(exe):
        from wafer_amd import build
        build.build_rccl_host()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "--self", "136", "40", "96", "9"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "SELF-OK" in r.stdout and "halo_calls=" in r.stdout
    # the third run of --self: peer stores (overlap mode 3), connected and checked against an exchange's bits as a multi-rank run does it
    assert "ms_per_step_peer_stores=" in r.stdout
    # the fourth: peer copies (overlap mode 4), after the same check
    assert "ms_per_step_peer_copies=" in r.stdout
    # the same binary as an ordinary single-rank run prints one JSON record
    import json
    r = subprocess.run([exe, "64", "48", "40", "10"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 10 and rec["norm2"] > 0 and rec["halo_overlap_mode"] == 2 and rec["peer_store_check"] == -1


def test_native_rccl_hooks_from_python_self_neighbours():
    """wafer_amd.slab.NativeRcclSlabComm (libwafer_rccl.so: RCCL's C API, torch's RCCL library in
    the process) with the rank as its own z-neighbour, bit for bit against device copies"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_native_selfloop_worker.py")],
                       capture_output=True, text=True, env=env, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "NATIVE-OK" in r.stdout


def test_transport_selection_and_fallback():
    """make_slab_comm: native RCCL hooks by default, all ranks fall back to the torch.distributed
    hooks together when they cannot be installed (tests/make_comm_worker.py)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1")
    env.pop("WAFER_TRANSPORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "make_comm_worker.py")],
                       capture_output=True, text=True, env=env, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "COMM-OK" in r.stdout and "falling back to torch.distributed" in r.stderr
