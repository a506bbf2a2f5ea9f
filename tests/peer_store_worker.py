"""Overlap mode 3 (peer stores) on WORLD slabs of one grid, all contexts in this process, against one context: prints which planes
differ after every evolve call (tests/test_gpu_slab.py runs it as a subprocess for four slabs: more contexts than the runtime's
default hardware queues make two ranks' kernels share a queue on ONE GPU, and a kernel that polls for a neighbour's stores must not
sit in front of that neighbour's kernel -- GPU_MAX_HW_QUEUES lifts the limit; with one process per GPU the situation cannot arise).

    python tests/peer_store_worker.py WORLD NX,NY,NZ STEPS[,STEPS...] [MODE [WAFER_HV_LAYOUT]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WAFER_FUSE3_MIN_NY"] = "1"
import numpy as np
import wafer_amd as wa
from tests.test_gpu_slab import run_slabs, assemble

world, shape = int(sys.argv[1]), tuple(int(x) for x in sys.argv[2].split(","))
calls = [int(x) for x in sys.argv[3].split(",")]
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 3
if len(sys.argv) > 5:
    os.environ["WAFER_HV_LAYOUT"] = sys.argv[5]
base = wa.Params(*shape, dn=0.2, dt=0.004, mass=1.0, central_difference=1, halo_depth=3)
with wa.Context(wa.Params(*shape, dn=0.2, dt=0.004, mass=1.0, central_difference=1)) as ctx:
    ctx.set_potential("Coulomb")
    ctx.set_initial_condition("Boolean")
    wants = []
    for n in calls:
        ctx.evolve(0, n)
        wants.append(ctx.download_phi())

def body(ctx, rank):
    ctx.set_overlap(mode)
    ctx.set_potential("Coulomb")
    ctx.set_initial_condition("Boolean")
    out = []
    for n in calls:
        ctx.evolve(0, n)
        out.append(ctx.download_phi())
    return out

res, fabric = run_slabs(wa, base, world, body)
ok = True
for i, n in enumerate(calls):
    got = assemble(base, world, [r[i] for r in res])
    bad = np.argwhere(got != wants[i])
    ok = ok and len(bad) == 0
    print("call", i, "steps", n, "differing cells", len(bad), "z planes", sorted(set(bad[:, 2].tolist()))[:40] if len(bad) else [],
          "max abs", float(np.max(np.abs(got - wants[i]))))
print("halo calls", fabric.halo_calls)
print("PEER-OK" if ok else "PEER-MISMATCH")
