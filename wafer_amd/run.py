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
drivers accept exactly the same files.  Not available across ranks: FromFile potentials /
wavefunctions and restarts from ./input (each rank would have to read the whole array), symmetry
constraints about z.  WAFER_TRANSPORT=host selects the host-staged gloo transport (ranks folded
onto the GPUs present; tests/test_gpu_multiprocess.py).
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


def sanitize(cli_arg: str) -> str:
    return subprocess.run([CLI, "--sanitize", cli_arg], capture_output=True, text=True).stdout.rstrip("\n")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m wafer_amd.run")
    ap.add_argument("-c", "--config", default="wafer.yaml")
    ap.add_argument("--output-dir", default="./output")
    ap.add_argument("--progress", action="store_true", help="print a table row per screen_update block")
    args = ap.parse_args(argv)

    # RCCL caches its parameters at first use (torch's own communicator): see wafer_rccl_hooks.h
    os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "8")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = load_config(args.config)
    if cfg["potential"] in ("FromFile", "FromScript") or cfg["init_condition"] == "FromFile" or cfg["wavenum"] > 0:
        raise SystemExit("wafer_amd.run: file potentials / wavefunctions and restarts are single-GPU features (wafer-hip)")

    import torch
    import wafer_amd
    from wafer_amd import slab

    if not torch.cuda.is_available():
        raise SystemExit("wafer_amd.run needs a GPU: the engine has no CPU path")
    host_transport = os.environ.get("WAFER_TRANSPORT", "rccl") == "host"
    if host_transport:
        local_rank %= torch.cuda.device_count()
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
                           z_begin=z_begin, z_count=z_count, halo_depth=2 * ext if world > 1 and ext <= 2 else 0)
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

    symmetry = cfg["init_symmetry"]
    t0 = time.perf_counter()
    exit_code = 0
    with wafer_amd.Context(par) as ctx:
        comm = None
        if world > 1:
            comm, _name = slab.make_slab_comm(ctx, rank, world, torch.device("cuda", local_rank),
                                              "host" if host_transport else None)
            comm.warm_up()
        ctx.set_potential(cfg["potential"])
        for wnum in range(cfg["wavenum"], cfg["wavemax"] + 1):          # grid.rs:43-45
            cloned = False
            if wnum == 0:                                                # grid.rs:99, config.rs:577-627
                ctx.set_initial_condition(cfg["init_condition"], seed=seed)
                ctx.symmetrise(symmetry)
            else:                                                        # grid.rs:95
                ctx.clone_state_to_phi(wnum - 1)
                cloned = True
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
                e = ext
                zb, zc = (0, cfg["nz"]) if world == 1 else (z_begin, z_count)
                piece = ctx.download_phi()[e:-e, e:-e, zb + e:zb + zc + e]
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
