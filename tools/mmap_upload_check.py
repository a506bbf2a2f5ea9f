"""How much host memory a rank touches when it uploads its slab of a big framed .npy potential
(wafer_amd.run's FromFile path): a sparse n^3 file, one slab of n/8 planes, peak RSS and time."""
import resource
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import wafer_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/potential_framed.npy"
a = np.lib.format.open_memmap(path, mode="w+", dtype=np.float64, shape=(n + 2, n + 2, n + 2))
z0, zc = n // 2, n // 8
a[1:-1:97, 1:-1:89, z0 - 1:z0 + zc + 3] = 0.25          # a few marked rows inside the slab's range
a.flush()
del a
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss


def now_gb():
    """resident set right now, file-backed and anonymous parts (GB)"""
    d = dict(l.split(":") for l in open("/proc/self/status") if l.startswith(("RssAnon", "RssFile", "RssShmem")))
    return {k: round(int(v.split()[0]) / 2**20, 2) for k, v in d.items()}


par = wafer_amd.Params(n, n, n, dn=0.02, dt=8e-5, dtype="f32", z_begin=z0, z_count=zc, halo_depth=2, max_states=1)
with wafer_amd.Context(par) as ctx:
    print("context created", now_gb())
    v = np.load(path, mmap_mode="r")
    print("mapped", now_gb())
    t0 = time.perf_counter()
    ctx.set_potential_host(v)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print("uploaded", now_gb())
    ctx.upload_phi(v)
    got = ctx.download_phi_owned()[0::97, 0::89, :]      # what a rank of wafer_amd.run saves: its own planes only
    assert got.shape[2] == zc and np.all(got == 0.25), "slab upload mismatch"
    rss2 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print("downloaded", now_gb())
print({"n": n, "slab_planes": zc, "file_GB": round((n + 2) ** 3 * 8 / 2**30, 2), "upload_s": round(dt, 2),
       "peak_rss_before_GB": round(rss0 / 2**20, 2), "peak_rss_after_upload_GB": round(rss1 / 2**20, 2),
       "peak_rss_after_download_GB": round(rss2 / 2**20, 2)})
