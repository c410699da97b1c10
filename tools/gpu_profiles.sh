#!/bin/bash
# the round's profile set (profiles/<round>_*): tools/gpu_profiles.sh r06   -- run through gpurun; copies land in gpurun_out/profiles/
R=${1:-r06}
bash tools/profile.sh ${R}_bistro4k --workload bistro
bash tools/profile.sh ${R}_sponza4k --workload sponza
bash tools/profile.sh ${R}_sanmiguel4k --workload san_miguel
bash tools/profile.sh ${R}_bistro4k_dense --workload bistro_dense
bash tools/profile.sh ${R}_zorah8k --workload zorah
