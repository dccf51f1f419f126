#!/bin/bash
# Counter passes of every benchmarked workload with the current build, and their per-segment models in
# profiles/pmc_model.json (bench.py scales them to the live run): tools/round3_models.sh <tag>
TAG=$1
OUT=gpurun_out/models_$TAG
mkdir -p $OUT
one() { # name kernel entry args...
  name=$1; kernel=$2; entry=$3; shift 3
  tools/run_pmc.sh ${TAG}_$name "$@" > $OUT/pmc_$name.txt 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.txt; exit 1; }
  cp gpurun_out/pmc_${TAG}_$name/summary.csv profiles/${TAG}_pmc_${name}_summary.csv
  python3 tools/pmc_model.py gpurun_out/pmc_${TAG}_$name $kernel "profiles/${TAG}_pmc_${name}_summary.csv: tools/run_pmc.sh $*, one rocprofv3 --pmc run per counter set" $entry > $OUT/model_$name.json || exit 1
}
one cornell k_path_pool k_path_pool cornell_box 1024 1024 64 50 26 &&
cp profiles/${TAG}_pmc_cornell_summary.csv profiles/${TAG}_pmc_summary.csv &&
one rtiow k_tree_pool random_scene random_scene 1200 800 50 50 26 &&
one final k_tree_pool final_book_two final_book_two 1024 1024 64 50 26 &&
one final_book k_tree_pool final_book_two:book final_book_two 1024 1024 64 50 26 1 1 &&
one smoke k_path_pool cornell_smoke cornell_smoke 1024 1024 64 50 26 &&
one smoke_book k_path_pool cornell_smoke:book cornell_smoke 1024 1024 64 50 26 1 1 &&
one cloud k_tree_pool sphere_cloud sphere_cloud 1920 1080 16 10 25 &&
cp profiles/pmc_model.json $OUT/pmc_model.json && echo models done
