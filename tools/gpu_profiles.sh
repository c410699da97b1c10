#!/bin/bash
# the round's judged profiles: one tools/profile.sh run per workload (writes profiles/<tag>_*), then the default bench line
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "r02_bistro4k --workload bistro" "r02_sponza4k --workload sponza" "r02_bistro4k_dense --workload bistro_dense" "r02_sanmiguel4k --workload san_miguel" \
            "r02_sanmiguel4k_alpha_tex --workload san_miguel --material-features 24" "r02_sponza4k_parallax --workload sponza --material-features 136" "$@"; do
  set -- $spec; tag=$1; shift
  bash tools/profile.sh $tag "$@" > /dev/null 2>&1
  echo "== $tag"; head -12 profiles/${tag}_pmc.txt | cut -c1-150
done
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json
