#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/job2; rm -rf $O; mkdir -p $O
cd $ROOT
python3 scratch/dbg_shade.py tiny sponza_small bistro_small sponza_textured bistro_ownlod_skinned sponza_layer_textures > $O/dbg.log 2>&1; cat $O/dbg.log | tail -20
timeout 1500 python3 -m pytest tests/test_parity_gpu.py -q -m gpu -k "hdr or light or toggles or golden or idempotent" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
