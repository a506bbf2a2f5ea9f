#!/bin/bash
# two steps per pass against one, per grid shape and number of stored states (same box, interleaved)
for grid in 256,256,256 384,384,384 512,512,128 512,512,256 1024,1024,128 1024,1024,256 1024,1024,1024; do
  for w in ${X2_K:-1 2 3}; do
    timeout 400 python3 tools/stencil_sweep.py --grid $grid --wnum $w --rounds 3 --steps 42 --configs "x2=0" "x2=1,x2k=3" 2>&1 | grep config | cut -c1-100 | sed "s/^/$grid k=$w /"
  done
done
