import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
import wafer_amd
from wafer_amd.slab import TorchSlabComm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29477")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
class SelfN(TorchSlabComm):
    lower = 0; upper = 0
kw = dict(dn=0.02, dt=8e-5, mass=2.35, sig=0.223, central_difference=1)
par = wafer_amd.Params(1024, 1024, 1024, z_begin=512, z_count=128, halo_depth=2, **kw)
with wafer_amd.Context(par) as ctx:
    comm = SelfN(ctx, 0, 1, dev); comm.warm_up()
    calls = {"t": 0.0, "n": 0}
    h, r = comm._halo_hook, comm._allreduce_hook
    def hh(*a):
        t0 = time.perf_counter(); rc = h(*a); calls["t"] += time.perf_counter() - t0; calls["n"] += 1; return rc
    ctx.set_comm_hooks(hh, r)
    ctx.set_potential("SimpleCornell"); ctx.set_initial_condition("Boolean")
    ctx.evolve(0, 100); ctx.synchronize()
    calls["t"] = 0; calls["n"] = 0
    t0 = time.perf_counter(); ctx.evolve(0, 200); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    print("evolve returned after %.1f ms, sync took further %.1f ms; hook: %d calls, %.1f us each" % ((t1-t0)*1e3, (t2-t1)*1e3, calls["n"], calls["t"]/max(1,calls["n"])*1e6))
    del comm
torch.cuda.synchronize(); dist.destroy_process_group()
