for w in 2 3 4; do
mkdir -p /tmp/v$w; /opt/rocm/bin/hipcc -DBRMI_SHADE_WAVES=$w --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Iinclude -Wall basicrenderer_amd/csrc/*.hip -o /tmp/v$w/libbrmi.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "k_shadeILb0" | grep -E "VGPRs:|Scratch|Occupancy" | head -3 | tr '\n' ' '; echo
echo -n "waves $w: "; BRMI_LIB_PATH=/tmp/v$w/libbrmi.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms']['shade'])"
done
echo -n "default: "; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'])"
