#!/usr/bin/env python3
"""Host<->device cost of the boundary's bulk calls at 512^3 fp64 (PCIe + on-device transpose),
next to the cost of one screen_update block -- the only per-block host traffic is 4 doubles."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wafer_amd
par = wafer_amd.Params(512, 512, 512, dn=0.05, dt=5e-4)
phi = np.random.default_rng(0).standard_normal(par.padded_shape)
with wafer_amd.Context(par) as ctx:
    ctx.set_potential("Coulomb")
    t = time.perf_counter(); ctx.upload_phi(phi); ctx.synchronize(); up = time.perf_counter() - t
    t = time.perf_counter(); out = ctx.download_phi(); dn = time.perf_counter() - t
    assert np.array_equal(out, phi)
    ctx.evolve(0, 100); ctx.synchronize()
    t = time.perf_counter(); ctx.evolve(0, 1000); obs = ctx.observables(); blk = time.perf_counter() - t
    gb = phi.nbytes / 1e9
    print(json.dumps({"upload_s": round(up, 3), "upload_GBps": round(gb / up, 1), "download_s": round(dn, 3),
                      "download_GBps": round(gb / dn, 1), "block_1000_steps_plus_observables_s": round(blk, 3),
                      "updates_per_s_resident": round(512**3 * 1000 / blk / 1e9, 1),
                      "updates_per_s_if_phi_crossed_pcie_every_block": round(512**3 * 1000 / (blk + up + dn) / 1e9, 1)}))
