for spec in "f64 3 1" "f64 3 2" "f32 1 3" "f32 2 1" "f32 2 2" "f32 2 3" "f32 3 1" "f32 3 2" "f32 3 3"; do set -- $spec
  timeout 300 python3 tools/stencil_sweep.py --grid 512,512,512 --dtype $1 --cd $2 --wnum $3 --rounds 3 --steps 30 --configs "v=-1,xfnw=4" "v=-1,xfnw=8" 2>&1 | grep config | cut -c12-70 | sed "s/^/$1 cd=$2 k=$3 /"
done
