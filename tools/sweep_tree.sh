#!/bin/bash
# One-at-a-time sweep of the tree pool's thresholds on the tree scenes (PTGPU_TUNING overrides pt_config.tuning):
# tools/sweep_tree.sh [scene ...] > out.txt   (scenes: rtiow final cloud; default all three)
scene_args() {
  case $1 in
    rtiow) echo "random_scene 1200 800 50 50 26";;
    final) echo "final_book_two 1024 1024 32 50 25";;
    cloud) echo "sphere_cloud 1920 1080 8 10 24";;
  esac
}
for s in ${*:-rtiow final cloud}; do
  for t in "" tree_shade=16 tree_shade=24 tree_shade=40 tree_shade=48 tree_inner=4 tree_inner=8 tree_inner=24 tree_inner=32 tree_refill=4 tree_refill=8 tree_refill=24 tree_refill=32 tree_trav_plus1=1 tree_trav_plus1=3 tree_trav_plus1=9 tree_trav_plus1=17 pool_grid=1 pool_grid=2 pool_grid=4 pool_grid=16; do
    r=$(PTGPU_TUNING=$t timeout -k 5 120 python3 tools/prof_driver.py $(scene_args $s) 2>/dev/null | grep -o "[0-9.]* Mpaths/s")
    echo "$s ${t:-default} $r"
  done
done
