"""Multi-GPU host driver: Wafer's run for a `wafer.yaml` with the grid z-slab decomposed over the
ranks of a torch.distributed job, one process per GPU (BASELINE config #4: 1024^3 over 8 GPUs).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m wafer_amd.run -c wafer.yaml [--output-dir DIR] [--progress]

The single-GPU driver is the native `wafer-hip` (wafer_amd/csrc/wafer_cli.cpp); this module is the
same loop -- grid::run / solve, grid.rs:31-47, 50-246 -- above the same C ABI, with
wafer_amd.slab.TorchSlabComm serving the engine's halo / all-reduce hooks over RCCL.  Every rank
executes the loop; the observables it sees are already all-reduced, so all ranks take the same
branches; rank 0 prints the table (output.rs:421-603) and writes the files.

Configuration parsing and validation are the native driver's (`wafer-hip --check-config`), so both
drivers accept exactly the same files.  Arrays from ./input -- `potential: FromFile`, a
`potential_sub` override, `wavefunction_N` restarts (grid.rs:35-39, 60-100) in any of the reference's
five formats -- are staged once by rank 0 as a framed .npy next to the input (`wafer-hip --convert
IN OUT.npy --pad E`) and memory-mapped by every rank: the engine copies only the planes of its own
slab out of the mapping, so no rank ever holds the whole array (BASELINE config #5: a 2048^3
potential from file over 8 GPUs).  A ready-made framed `<stem>.npy` in ./input is used as is.
Not available across ranks: resampling an array of another resolution (do that once with
`wafer-hip`), potentials from a script, symmetry constraints about z.  WAFER_TRANSPORT=host selects
the host-staged gloo transport (ranks folded onto the GPUs present; tests/test_gpu_multiprocess.py).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "wafer_amd", "wafer-hip")
WIDTH = 100  # output.rs:733-745 without a terminal


# ---- Rust std::fmt restated (output.rs:421-603) ------------------------------------------------
def rust_lower_exp(v: float, prec: int) -> str:
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    m, e = f"{v:.{prec}e}".split("e")
    return f"{m}e{int(e)}"


def rust_display(v: float) -> str:
    """`{}` on f64: shortest round-trip digits, never an exponent, "1" for 1.0"""
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    s = np.format_float_positional(v, unique=True, trim="-")
    return s


def ordinal(n: int) -> str:
    suf = "th"
    if n % 100 < 11 or n % 100 > 13:
        suf = {1: "st", 2: "nd", 3: "rd"}.get(n % 10, "th")
    return f"{n}{suf}"


def _centre(s: str, w: int, fill: str) -> str:
    total = max(0, w - len(s))
    left = total // 2
    return fill * left + s + fill * (total - left)


def observable_header(wnum: int) -> str:
    spacer = (WIDTH - 69) // 2
    rspace = spacer + 1 if 2 * spacer + 69 < WIDTH else spacer
    title = " Ground state caclulation " if wnum == 0 else f" {ordinal(wnum)} excited state caclulation "
    return "\n".join([
        "",
        "╤".join([_centre("", spacer, "═"), _centre("", 12, "═"), _centre(title, 37, "═"), _centre("", 16, "═"),
                  _centre("", rspace, "═")]),
        "│".join([_centre("", spacer, " "), _centre("Time (τ)", 12, " "), _centre("Energy", 20, " "),
                  _centre("rᵣₘₛ", 16, " "), _centre("Difference", 16, " ")]) + "│",
        "┼".join([_centre("", spacer, "─"), _centre("", 12, "─"), _centre("", 20, "─"), _centre("", 16, "─"),
                  _centre("", 16, "─"), _centre("", rspace, "─")]),
    ])


def measurement_row(tau: float, diff: float, obs: dict) -> str:
    spacer = " " * ((WIDTH - 69) // 2)
    last = f"{rust_lower_exp(diff, 5):>15} │" if tau > 0 else f"{'--   ':>15} │"
    return (f"{spacer}│{tau:>11.3f} │{rust_lower_exp(obs['energy'] / obs['norm2'], 10):>19} │"
            f"{math.sqrt(obs['r2'] / obs['norm2']):15.5f} │{last}")


def summary(fin: dict) -> str:
    spacer = (WIDTH - 69) // 2
    rspace = spacer + 1 if 2 * spacer + 69 < WIDTH else spacer
    who = "Ground state" if fin["state"] == 0 else f"{ordinal(fin['state'])} excited state"
    return "\n".join([
        "╧".join([_centre("", spacer, "═"), _centre("", 12, "═"), _centre("", 20, "═"), _centre("", 16, "═"),
                  _centre("", 16, "═"), _centre("", rspace, "═")]),
        f"══▶ {who} energy = {rust_display(fin['energy'])}",
        f"══▶ {who} binding energy = {rust_display(fin['binding_energy'])}",
        f"══▶ rᵣₘₛ = {rust_display(fin['r'])}",
        f"══▶ L/rᵣₘₛ = {rust_display(fin['l_r'])}",
        "",
    ])


def load_config(path: str) -> dict:
    """parsed and validated by the native driver (config.rs:292-370 restated there)"""
    if not os.path.exists(CLI):
        from wafer_amd import build
        build.build()
    r = subprocess.run([CLI, "-c", path, "--check-config"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr.strip() or f"could not read {path}")
    return json.loads(r.stdout)


FILE_EXT = {"Messagepack": ".mpk", "Csv": ".csv", "Json": ".json", "Yaml": ".yaml", "Ron": ".ron"}


def find_input(input_dir: str, stem: str, file_type: str):
    """input.rs:75-110, 264-300, 542-575: <dir>/<stem>.{mpk,csv,json,yaml,ron}; with several present
    the configured type arbitrates, otherwise the first in that order (wafer_files.h find_input)"""
    present = [e for e in FILE_EXT.values() if os.path.exists(os.path.join(input_dir, stem + e))]
    if not present:
        return None
    pick = FILE_EXT[file_type] if len(present) > 1 and FILE_EXT[file_type] in present else present[0]
    return os.path.join(input_dir, stem + pick)


def staged_array(input_dir: str, stem: str, file_type: str, pad: int, rank: int, wait_s: float = 6 * 3600.0):
    """The array of ./input/<stem>.* as a read-only memory map with `pad` zero cells around the work
    area, or None if there is no such file.  Rank 0 converts (cached by modification time); the other
    ranks wait for the file -- no collective, so that nothing times out under a long conversion."""
    ready = os.path.join(input_dir, stem + ".npy")
    if os.path.exists(ready):
        return np.load(ready, mmap_mode="r")
    src = find_input(input_dir, stem, file_type)
    if src is None:
        return None
    cache = os.path.join(input_dir, ".wafer_amd")
    dst = os.path.join(cache, f"{stem}.pad{pad}.npy")
    fresh = lambda: os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src)  # noqa: E731
    if rank == 0 and not fresh():
        os.makedirs(cache, exist_ok=True)
        tmp = os.path.join(cache, f"{stem}.pad{pad}.tmp.npy")
        r = subprocess.run([CLI, "--convert", src, tmp, "--pad", str(pad)], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"could not read {src}: {r.stderr.strip()}")
        os.replace(tmp, dst)
    t0 = time.time()
    while not fresh():
        if time.time() - t0 > wait_s:
            raise SystemExit(f"rank {rank}: gave up waiting for rank 0 to stage {src}")
        time.sleep(0.05)
    return np.load(dst, mmap_mode="r")


def sanitize(cli_arg: str) -> str:
    return subprocess.run([CLI, "--sanitize", cli_arg], capture_output=True, text=True).stdout.rstrip("\n")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m wafer_amd.run")
    ap.add_argument("-c", "--config", default="wafer.yaml")
    ap.add_argument("--output-dir", default="./output")
    ap.add_argument("--input-dir", default="./input")
    ap.add_argument("--progress", action="store_true", help="print a table row per screen_update block")
    args = ap.parse_args(argv)

    # RCCL caches its parameters at first use (torch's own communicator): see wafer_rccl_hooks.h
    os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "8")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL's peer mappings (as bench.py)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = load_config(args.config)
    if cfg["potential"] == "FromScript":
        raise SystemExit("wafer_amd.run: script potentials are a single-GPU feature (wafer-hip -s); "
                         "save the potential there and use potential: FromFile")

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # the engine's two streams beside RCCL's and torch's: more hardware queues than the runtime's default of four, so that a kernel
        # waiting for a flag and the kernel (or copy) that sets it never share one (bench.py says why); read when HIP initialises
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import wafer_amd
    from wafer_amd import slab

    if not torch.cuda.is_available():
        raise SystemExit("wafer_amd.run needs a GPU: the engine has no CPU path")
    host_transport = os.environ.get("WAFER_TRANSPORT", "rccl") == "host"
    # a launcher that narrows each rank's view to its own GPU (ROCR_/HIP_VISIBLE_DEVICES per rank) leaves
    # one visible device with index 0; the host-staged test transport folds ranks onto the GPUs present
    if host_transport or local_rank >= torch.cuda.device_count():
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_transport:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ext = cfg["central_difference"]
    z_begin, z_count = (0, 0) if world == 1 else slab.partition(cfg["nz"], world, rank)
    par = wafer_amd.Params(cfg["nx"], cfg["ny"], cfg["nz"], dn=cfg["dn"], dt=cfg["dt"], mass=cfg["mass"], sig=cfg["sig"],
                           central_difference=ext, dtype=cfg["dtype"], max_states=cfg["wavemax"] + 1, device=local_rank,
                           z_begin=z_begin, z_count=z_count, halo_depth=(3 if ext == 1 else 2 * ext if ext <= 2 else 0) if world > 1 else 0)
    say = (lambda *a, **k: print(*a, **k, flush=True)) if rank == 0 else (lambda *a, **k: None)
    out_dir = None
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        out_dir = os.path.join(args.output_dir, sanitize(cfg["project_name"]) + "_" + time.strftime("%Y-%m-%d_%H:%M:%S"))
        os.makedirs(out_dir)
        with open(args.config) as src, open(os.path.join(out_dir, os.path.basename(args.config)), "w") as dst:
            dst.write(src.read())

    seed = int(time.time())  # the Gaussian start is keyed by the global cell index: one seed for all ranks
    if dist is not None:     # every rank writes its planes next to rank 0's files
        box = [out_dir, seed]
        dist.broadcast_object_list(box, src=0)
        out_dir, seed = box

    if world > 1 and os.environ.get("WAFER_PEER_STORES", "0") == "force":
        os.environ.setdefault("WAFER_PEER_SAME_DEVICE", "1")   # read by wafer_ctx_create
    symmetry = cfg["init_symmetry"]
    t0 = time.perf_counter()
    exit_code = 0
    with wafer_amd.Context(par) as ctx:
        comm = None
        use_peer_stores = use_peer_copies = False
        if world > 1:
            comm, _name = slab.make_slab_comm(ctx, rank, world, torch.device("cuda", local_rank),
                                              "host" if host_transport else None)
            comm.warm_up()
            # the 1 + k scalars of every excited-state step: ncclAllReduce or the device-side mailboxes, whichever the fabric
            # serves faster (a solve spends most of its steps in excited states); WAFER_MAILBOX=0 / 1 fixes the choice
            if hasattr(comm, "pick_allreduce") and os.environ.get("WAFER_MAILBOX", "") == "":
                choice = comm.pick_allreduce()
                if rank == 0:
                    print(f"scalar all-reduce: {choice}", file=sys.stderr, flush=True)
            # peer stores (overlap mode 3) for the ground-state passes, or peer copies (mode 4: every exchange a device copy into the
            # neighbour's ghost planes, the copy engines between GPUs) where every rank can map its neighbours (HIP IPC):
            # WAFER_PEER_STORES=1 asks for mode 3 with mode 4 as its fallback, WAFER_PEER_STORES=copies for mode 4 alone
            # (=force: mode 3 also with the host-staged test transport, ranks folded onto one GPU -- tests)
            want_peers = os.environ.get("WAFER_PEER_STORES", "0")
            if want_peers not in ("", "0") and (not host_transport or want_peers in ("force", "copies")):
                # (mode 3 is the ThreePoint three-step pass with 6 ext owned planes on every rank: a rank-invariant test -- a rank
                #  that alone refused the mode would leave the others waiting for its planes)
                thinnest = min(slab.partition(par.nz, world, r)[1] for r in range(world))
                if slab.connect_peers(ctx, rank, world):
                    use_peer_copies = True
                    use_peer_stores = want_peers != "copies" and ext == 1 and thinnest >= 6 * ext   # (switched on below, once the potential is set and the schedule has proved itself)
        def from_input(stem, pad, shape, what):
            a = staged_array(args.input_dir, stem, cfg["file_type"], pad, rank)
            if a is not None and (a.dtype != np.float64 or tuple(a.shape) != tuple(shape)):
                raise SystemExit(f"{what}: {stem} holds {tuple(a.shape)} {a.dtype}, this run needs {tuple(shape)} float64 "
                                 f"(the frame of {pad} cells included); resample it once with wafer-hip")
            return a

        if cfg["potential"] == "FromFile":                              # potential.rs:80-86
            v = from_input("potential", ext, par.padded_shape, "LoadPotential")
            if v is None:
                raise SystemExit(f"Error: LoadPotential: FileNotFound: {args.input_dir}/potential.*")
            say("Loading potential from file", file=sys.stderr)
            ctx.set_potential_host(v)
            del v
        else:
            ctx.set_potential(cfg["potential"])
        if use_peer_stores or use_peer_copies:
            # neither has ever crossed a link: a schedule has to reproduce the bits of an exchange through the halo hook on every
            # rank (15 ground-state steps from the Boolean start, every cell) before the solve is handed to it
            dev = "cpu" if host_transport else f"cuda:{local_rank}"
            chosen = 2
            if use_peer_stores and slab.overlap_modes_agree(ctx, rank, world, 3, 2, steps=15, device=dev):
                chosen = 3
            elif use_peer_copies and slab.overlap_modes_agree(ctx, rank, world, 4, 2, steps=15, device=dev):
                chosen = 4
            elif use_peer_copies and slab.overlap_modes_agree(ctx, rank, world, 5, 2, steps=15, device=dev):
                chosen = 5      # the copies under mode 1's launches: every reader of ghost planes starts after the copy
            ctx.set_overlap(chosen)
            if rank == 0:
                print({3: "halo schedule: overlap mode 3 (peer stores)", 4: "halo schedule: overlap mode 4 (peer copies)", 5: "halo schedule: overlap mode 5 (peer copies, boundary planes first)",
                       2: "halo schedule: neither peer stores nor peer copies reproduce the exchange's bits on this fabric; overlap mode 2"}[chosen],
                      file=sys.stderr, flush=True)
        sub = staged_array(args.input_dir, "potential_sub", cfg["file_type"], 0, rank)   # potential.rs:113-131
        if sub is not None:
            variable = cfg["potential"] == "FullCornell"
            if (sub.ndim == 0) == variable:
                raise SystemExit("Error: WrongPotentialSubDims: potential_sub input file does not suit the potential type")
            if sub.ndim == 0:
                ctx.set_potsub(1, float(sub))
            else:
                if tuple(sub.shape) != par.work_shape:
                    raise SystemExit(f"potential_sub holds {tuple(sub.shape)}, the grid is {par.work_shape}; resample it once with wafer-hip")
                ctx.set_potsub(2, 0.0, sub)
            say("Potential_sub loaded from disk", file=sys.stderr)
            del sub
        for w in range(cfg["wavenum"]):                                  # grid.rs:35-39: converged lower states from disk
            st = from_input(f"wavefunction_{w}", ext, par.padded_shape, f"LoadWavefunction({w})")
            if st is None:
                raise SystemExit(f"Error: LoadWavefunction({w}): FileNotFound: {args.input_dir}/wavefunction_{w}.*")
            ctx.upload_phi(st)
            ctx.push_state()
            del st
        for wnum in range(cfg["wavenum"], cfg["wavemax"] + 1):          # grid.rs:43-45
            cloned = False
            start = from_input(f"wavefunction_{wnum}", ext, par.padded_shape, f"LoadWavefunction({wnum})")
            if start is None:                                            # input.rs:513-523
                start = from_input(f"wavefunction_{wnum}_partial", ext, par.padded_shape, f"LoadWavefunction({wnum})")
            if wnum > 0:
                if start is not None:
                    ctx.upload_phi(start)
                else:                                                    # grid.rs:95
                    ctx.clone_state_to_phi(wnum - 1)
                    cloned = True
            elif cfg["init_condition"] == "FromFile":
                if start is None:
                    raise SystemExit(f"Error: SetInitialConditions: LoadWavefunction(0): FileNotFound: {args.input_dir}/wavefunction_0*.*")
                ctx.upload_phi(start)
            else:                                                        # grid.rs:99, config.rs:577-627
                ctx.set_initial_condition(cfg["init_condition"], seed=seed)
            del start
            if wnum == 0:
                ctx.symmetrise(symmetry)                                 # config.rs:625
            say(observable_header(wnum))
            step, last_energy, converged = 0, sys.float_info.max, False
            while True:                                                  # grid.rs:126-221
                obs = ctx.observables()
                norm_energy = obs["energy"] / obs["norm2"]
                tau = step * cfg["dt"]
                ctx.normalise(obs["norm2"])
                if wnum > 0:
                    ctx.orthogonalise(wnum)
                if cloned and step == 0:
                    n2 = ctx.norm2()       # the clone can be annihilated to exactly zero (see wafer_cli.cpp)
                    if not (n2 > 0.0) or not math.isfinite(n2):
                        say(f"Warning: the clone of state {wnum - 1} was annihilated exactly by Gram-Schmidt; "
                            f"starting state {wnum} from Gaussian noise instead.", file=sys.stderr)
                        ctx.set_initial_condition("Gaussian", seed=0x5EED + wnum)
                        cloned, last_energy = False, sys.float_info.max
                        continue
                diff = abs(norm_energy - last_energy)
                if not math.isfinite(norm_energy):
                    raise SystemExit(f"state {wnum}: energy is not finite at step {step}")
                if diff < cfg["tolerance"]:
                    say(measurement_row(tau, diff, obs))
                    converged = True
                    break
                if args.progress:
                    say(measurement_row(tau, diff, obs))
                last_energy = norm_energy
                if cfg["max_steps"] is not None and step > cfg["max_steps"]:   # grid.rs:211-213
                    break
                ctx.evolve(wnum, cfg["screen_update"])
                step += cfg["screen_update"]
            r_norm = math.sqrt(obs["r2"] / obs["norm2"])
            fin = dict(state=wnum, energy=obs["energy"] / obs["norm2"],
                       binding_energy=(obs["energy"] - obs["v_infinity"]) / obs["norm2"], r=r_norm, l_r=cfg["nx"] / r_norm)
            if converged and rank == 0:                                  # output.rs:533-558
                print(summary(fin), flush=True)
                with open(os.path.join(out_dir, f"observables_{wnum}.json"), "w") as f:
                    json.dump(fin, f, indent=2)
                with open(os.path.join(out_dir, f"observables_{wnum}.csv"), "w") as f:
                    f.write("state,energy,binding_energy,r,l_r\n%d,%r,%r,%r,%r\n" % (
                        wnum, fin["energy"], fin["binding_energy"], fin["r"], fin["l_r"]))
            if cfg["save_wavefns"]:      # each rank saves the planes it owns (work area, reference axis order)
                zb, zc = (0, cfg["nz"]) if world == 1 else (z_begin, z_count)
                piece = ctx.download_phi_owned()     # a host buffer the size of the slab, not of the grid
                np.save(os.path.join(out_dir, f"wavefunction_{wnum}{'' if converged else '_partial'}_z{zb}-{zb + zc}.npy"), piece)
            if not converged:                                            # grid.rs:243-245
                say(f"Error: MaxStep: maximum step limit reached for state {wnum}", file=sys.stderr)
                exit_code = 1
                break
            ctx.push_state()                                             # grid.rs:241
        ctx.synchronize()
        if hasattr(comm, "close"):
            comm.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    say(f"Simulation complete. Elapsed time: {time.perf_counter() - t0:.3f} seconds.\nOutput directory: {out_dir}")
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
