import ctypes as C, os, sys, itertools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wafer_amd
wafer_amd.load_library()
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1]); break
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
def halo(slo, shi, rlo, rhi, nbytes, stream):
    if rlo: assert hip.hipMemcpyAsync(rlo, shi, nbytes, 3, stream) == 0
    if rhi: assert hip.hipMemcpyAsync(rhi, slo, nbytes, 3, stream) == 0
    return 0
def allreduce(ptr, count, stream): return 0
def run(shape, mode, calls, layout="0"):
    os.environ["WAFER_HV_LAYOUT"] = layout
    nx, ny, nz = shape
    per = nz // 8
    par = wafer_amd.Params(nx, ny, nz, z_begin=4 * per, z_count=per, halo_depth=3, dn=0.02, dt=8e-5, mass=2.35, sig=0.223, central_difference=1)
    with wafer_amd.Context(par) as ctx:
        ctx.set_comm_hooks(halo, allreduce)
        if mode == 3:
            rec = ctx.peer_export(); ctx.peer_connect(rec, rec)
        ctx.set_overlap(mode)
        ctx.set_potential("SimpleCornell")
        ctx.set_initial_condition("Boolean")
        ctx.normalise(1.0)
        out = []
        for n in calls:
            ctx.evolve(0, n); ctx.synchronize()
            out.append(ctx.checksum(par.z_begin, par.z_count))
        return out
for shape in [(136, 40, 96), (256, 64, 96)]:
    for calls in [(9,), (9, 9), (15,), (3, 3), (6, 3), (3,), (6,)]:
        ref = run(shape, 2, calls)
        for layout in ("3", "4"):
            got = run(shape, 3, calls, layout)
            print(shape, calls, "layout", layout, "mode3==mode2:", [a == b for a, b in zip(got, ref)], flush=True)
        print(shape, calls, "mode0==mode2:", [a == b for a, b in zip(run(shape, 0, calls), ref)], flush=True)
