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


def test_bench_multi_rank_path():
    """bench.py's N > 1 leg end to end (slab partition, hooks, max-over-ranks timing, one JSON line
    from rank 0) with two ranks on the one GPU over the host-staged transport"""
    import json
    env_extra = {"WAFER_BENCH_TRANSPORT": "host"}
    os.environ.update(env_extra)
    try:
        r = launch(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                   "--grid", "256,256,128")
    finally:
        for k in env_extra:
            os.environ.pop(k, None)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "zslab2" and d["config"]["points_per_gpu"] == 256 * 256 * 64
    assert d["roofline"]["steps_per_launch"] == 2 and "cpu_baseline" not in d


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
