#!/bin/bash
# one iteration of the shading-pass work: HDR / light parity subset, then the SQ counter table and bench line of Sponza and Bistro 4K
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out/iter
timeout 1500 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "hdr or light or toggles or golden or idempotent or full_size_frames" > gpurun_out/iter/pytest.log 2>&1
tail -4 gpurun_out/iter/pytest.log
bash tools/sq_probe.sh iter/sponza --workload sponza 2>&1 | head -8
bash tools/sq_probe.sh iter/bistro --workload bistro 2>&1 | head -8 | tail -6
