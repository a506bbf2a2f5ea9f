#!/bin/bash
# The three-step kernel with the potential's closed form evaluated at level 1 instead of V streamed (what the excited-state kernels
# do with their template parameter VG), against the kernel as built, same box.  The variant is a copy of the sources with eight edits,
# built here before the GPU call; one closed form per build (4 = Coulomb, 9 = Harmonic), fp64 only -- an experiment, not a product path:
#   bash tools/ab_closed_form_v_fused3.sh build 4           # -> wafer_amd/build/alt_vg4/libwafer_hip.so (travels with the snapshot)
#   gpurun -- 'bash tools/ab_closed_form_v_fused3.sh run 4 Coulomb'   # -> gpurun_out/vg4/{ab.log,bench_alt.json}
# Result (profiles/r06_ab_closed_form_v_fused3.jsonl): bit exact against the oracle; Coulomb 9 % SLOWER at 512^3 (0.2222 against 0.2018
# ms/step), 7 % at 256^3; Harmonic 2-3 % slower at 512^3, 1 % at 256^3.  The kernel keeps streaming V.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VG=${2:-4}
if [ "$1" = build ]; then
  python3 - "$ROOT" "$VG" <<'PY'
import glob, os, shutil, sys
ROOT, VG = sys.argv[1], int(sys.argv[2])
SRC = f"{ROOT}/wafer_amd/build/src_vg{VG}"
shutil.rmtree(SRC, ignore_errors=True); os.makedirs(SRC)
for f in glob.glob(f"{ROOT}/wafer_amd/csrc/*"):
    if f.endswith((".h", ".hip", ".inc")): shutil.copy(f, SRC)
def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old, s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)
edit(f"{SRC}/wafer_stencil_fused3.hip.h", [
 ("    const WaferDen<C> den = wafer_den<C>(a, vir);\n    // the extra slot:",
  f"    const WaferDen<C> den = wafer_den<C>(a, vir);\n    WaferPotArgs vgen; vgen.g = g; vgen.type = {VG}; vgen.dn = a.vg_dn; vgen.dt = a.dt; vgen.mass = a.vg_mass; vgen.sig = a.vg_sig; vgen.mu_t = vgen.alphas_2pit = vgen.xi_coef = vgen.xi_fac = 0.0;\n    // the extra slot:"),
 ("#pragma unroll\n        for (int r = 0; r < RY; ++r) vcur[r] = gload((pv + po + rowoff[r]) + xlu);\n        if (x_row) xv = gload((pv + po + xoff_row) + xlu);\n        else xv[0] = (T)pv[po + c_off];\n",
  "        (void)po;\n"),
])
edit(f"{SRC}/wafer_stencil_fused3_iter.inc.h", [
 ("                if constexpr (DIRECT) vcur[r] = gload((pv + zo + SD * g.plane + rowoff[r]) + xlu);\n                else pre_v[r] = gload_raw((pv + zo + SD * g.plane + rowoff[r]) + xlu);\n", "                (void)r;\n"),
 ("                xv = gload(pv + zo + SD * g.plane + xslot_off);\n", ""),
 ("                xpre_v = gload_raw(pv + zo + SD * g.plane + xslot_off);\n", ""),
 ("const T rs = update_keep(w, (C)vcur[r][v], S, ka, kb);", f"const T rs = update_keep(w, (C)wafer_vgen_at<{VG}>(vgen, xi + v + R, yrow[r] + R, g.zp_of(z)), S, ka, kb);"),
 ("const T rs = update_keep(w, (C)xv[v], S, ka, kb);", f"const T rs = update_keep(w, (C)wafer_vgen_at<{VG}>(vgen, xi + v + R, xy + R, g.zp_of(z)), S, ka, kb);"),
 ("rs = update_keep(w, (C)xv[0], S, ka, kb);", f"rs = update_keep(w, (C)wafer_vgen_at<{VG}>(vgen, cxw + R, cy + R, g.zp_of(z)), S, ka, kb);"),
])
PY
  SRC=$ROOT/wafer_amd/build/src_vg$VG; OUT=$ROOT/wafer_amd/build/alt_vg$VG
  mkdir -p "$OUT"; cd "$SRC"
  SRCS=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); from wafer_amd import build; print(' '.join(build.SOURCES))")
  for s in $SRCS; do
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -I"$ROOT/wafer_amd/csrc" -c $s -o "$OUT/${s%.hip}.o" &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $(for s in $SRCS; do echo "$OUT/${s%.hip}.o"; done) -o "$OUT/libwafer_hip.so"
  find "$OUT" -name "*.o" -delete
  exit 0
fi
POT=${3:-Coulomb}
cd "$ROOT"; mkdir -p gpurun_out/vg$VG
ALT=$ROOT/wafer_amd/build/alt_vg$VG/libwafer_hip.so
for rep in 1 2 3; do
  for g in 512,512,512 256,256,256; do
    for lib in base alt; do
      if [ $lib = alt ]; then export WAFER_HIP_LIB=$ALT; else unset WAFER_HIP_LIB; fi
      echo "== $g $lib rep $rep" >> gpurun_out/vg$VG/ab.log
      timeout 300 python tools/stencil_sweep.py --potential $POT --grid $g --rounds 5 --steps 60 --configs "v=-1" 2>/dev/null | grep config >> gpurun_out/vg$VG/ab.log
    done
  done
done
export WAFER_HIP_LIB=$ALT
timeout 600 python bench.py --potential $POT --steps 60 --warmup 6 --no-cpu-baseline --no-excited > gpurun_out/vg$VG/bench_alt.json 2> gpurun_out/vg$VG/bench_alt.err
cat gpurun_out/vg$VG/ab.log
