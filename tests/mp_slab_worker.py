"""One rank of the multi-process z-slab run (launched by tests/test_gpu_multiprocess.py through
torch.distributed.run, gloo rendezvous on 127.0.0.1).  All ranks share the box's one GPU; halo
planes and scalars travel through wafer_amd.slab.HostStagedSlabComm.  Rank 0 also runs the SAME
problem in one undecomposed context and compares:

  * ground-state evolve (fused two-step kernel, overlap on): bit for bit,
  * all-reduced observables: 1e-12,
  * three-step passes with PEER STORES (overlap mode 3): every process maps its neighbours' buffers through HIP IPC and its
    boundary workgroups store straight into their ghost planes: bit for bit,
  * the same passes with PEER COPIES (overlap mode 4): every exchange a hipMemcpyAsync into the neighbour process's ghost planes
    through the same mapping, ordered by the credit / arrival rendezvous; the halo hook is never called: bit for bit,
  * solve of ground + first excited state from Gaussian starts: energies 5e-7.

Prints "MP-OK <world>" on success."""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import wafer_amd as wa
    from wafer_amd.slab import HostStagedSlabComm, partition

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    # ---- ground state: fused passes over slabs ----
    shape, ext, steps = (72, 40, 50), 1, 11
    whole = wa.Params(*shape, dn=0.2, dt=0.004, central_difference=ext)
    zb, zc = partition(shape[2], world, rank)
    mine = dataclasses.replace(whole, z_begin=zb, z_count=zc, halo_depth=2 * ext)
    with wa.Context(mine) as ctx:
        comm = HostStagedSlabComm(ctx, rank, world, dev)
        ctx.set_potential("Coulomb")
        ctx.set_initial_condition("Boolean")
        ctx.evolve(0, steps)
        ctx.evolve(0, 4)
        got = ctx.download_phi()
        obs = ctx.observables()
    pieces = [None] * world if rank == 0 else None
    dist.gather_object((zb, zc, got[:, :, zb + ext:zb + zc + ext]), pieces, dst=0)
    all_obs = [None] * world if rank == 0 else None
    dist.gather_object(obs, all_obs, dst=0)
    if rank == 0:
        with wa.Context(whole) as ctx:
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Boolean")
            ctx.evolve(0, steps)
            ctx.evolve(0, 4)
            want = ctx.download_phi()
            want_obs = ctx.observables()
        full = np.zeros_like(want)
        for b, c, p in pieces:
            full[:, :, b + ext:b + c + ext] = p
        assert np.array_equal(full, want), "slab evolve differs from the undecomposed run"
        for o in all_obs:
            for k, v in want_obs.items():
                assert abs(o[k] - v) <= 1e-12 * max(1.0, abs(v)), (k, o[k], v)

    # ---- peer stores (overlap mode 3): boundary workgroups write the neighbour PROCESS's ghost planes through HIP IPC ----
    from wafer_amd.slab import connect_peers
    os.environ["WAFER_FUSE3_MIN_NY"] = "1"
    os.environ["WAFER_PEER_SAME_DEVICE"] = "1"   # the ranks share the box's one GPU on purpose (read by wafer_ctx_create)
    shape3 = (140, 40, 45)
    whole3 = wa.Params(*shape3, dn=0.2, dt=0.004, central_difference=1)
    zb, zc = partition(shape3[2], world, rank)
    results = {}
    for mode in (3, 4):   # peer stores; peer COPIES (round 6: hipMemcpyAsync into the neighbour process's ghost planes through the same HIP IPC mapping)
        with wa.Context(dataclasses.replace(whole3, z_begin=zb, z_count=zc, halo_depth=3)) as ctx:
            comm = HostStagedSlabComm(ctx, rank, world, dev)
            assert connect_peers(ctx, rank, world), "peer connection (HIP IPC) failed"
            ctx.set_overlap(mode)
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Boolean")
            ctx.evolve(0, 12)
            ctx.evolve(0, 4)
            ctx.evolve(0, 9)
            got = ctx.download_phi()
            n2 = ctx.norm2()
            if mode == 4:
                assert comm.halo_calls == 0, "peer copies went through the halo hook"
        pieces = [None] * world if rank == 0 else None
        dist.gather_object((zb, zc, got[:, :, zb + 1:zb + zc + 1], n2), pieces, dst=0)
        results[mode] = pieces
    if rank == 0:
        with wa.Context(whole3) as ctx:
            ctx.set_potential("Coulomb")
            ctx.set_initial_condition("Boolean")
            ctx.evolve(0, 12)
            ctx.evolve(0, 4)
            ctx.evolve(0, 9)
            want = ctx.download_phi()
            want_n2 = ctx.norm2()
        for mode, pieces in results.items():
            full = np.zeros_like(want)
            for b, c, p, n2_ in pieces:
                full[:, :, b + 1:b + c + 1] = p
                assert abs(n2_ - want_n2) <= 1e-12 * want_n2
            assert np.array_equal(full, want), f"overlap mode {mode} ({'peer stores' if mode == 3 else 'peer copies'}) differs from the undecomposed run"

    # ---- ground + first excited state solve ----
    shape2 = (16, 16, 20)
    whole2 = wa.Params(*shape2, dn=0.55, dt=0.05, central_difference=1, max_states=2)
    zb, zc = partition(shape2[2], world, rank)

    def solve_all(ctx):
        ctx.set_potential("Harmonic")
        out = []
        for wnum in range(2):
            ctx.set_initial_condition("Gaussian", seed=9 + wnum)
            recs, final, conv = ctx.solve_state(wnum, 1e-7, 100, max_steps=20000)
            assert conv
            out.append((final["energy"], len(recs)))
        return out

    with wa.Context(dataclasses.replace(whole2, z_begin=zb, z_count=zc)) as ctx:
        comm = HostStagedSlabComm(ctx, rank, world, dev)
        mine_e = solve_all(ctx)
    every = [None] * world if rank == 0 else None
    dist.gather_object(mine_e, every, dst=0)
    if rank == 0:
        with wa.Context(whole2) as ctx:
            want_e = solve_all(ctx)
        for e in every:
            for (ge, gn), (we, wn) in zip(e, want_e):
                assert abs(ge - we) < 5e-7 and abs(gn - wn) <= 1, (e, want_e)
        assert abs(want_e[0][0] - 1.5) < 0.06 and abs(want_e[1][0] - 2.5) < 0.1
        print(f"MP-OK {world}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
