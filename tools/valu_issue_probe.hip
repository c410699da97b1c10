// valu_issue_probe.hip -- what a SIMD of gfx950 issues per cycle, measured: independent v_fma_f32 / v_rcp_f32 / v_cndmask_b32 streams at
// 1, 2, 3 and 4 waves per SIMD on every CU, and the clock the chip holds meanwhile.
//
// Why: DESIGN.md 4.6 prices k_shade against "one wave64 VALU instruction per SIMD every 4 cycles", MI355X_MICROARCH.md says 2 cycles once more
// than one wave is resident (4 for a wave alone).  The two readings differ by 2x in what is left to gain; this settles it on the box.
//
// One 256 x W-thread workgroup per CU (a workgroup's waves go round the four SIMDs, so W waves land on each; 120 KB of LDS per workgroup keeps a
// second one off the CU).  Every wave runs TRIPS trips of 256 instructions on eight independent accumulators (no dependency closer than eight
// instructions), stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop.
//   wave-instructions per SIMD and cycle = W * TRIPS * 256 / median over waves of (cycles in the loop)
//   clock = cycles / (realtime ticks / 100 MHz)
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_issue_probe.hip -o build/valu_issue_probe      Run: build/valu_issue_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Op { OP_FMA = 0, OP_RCP = 1, OP_MIX = 2, OP_CNDMASK = 3, OP_MOV = 4, OP_CNDMASK_SGPR = 5, OP_EXP = 6, OP_SQRT = 7, OP_MED3 = 8, OP_MUL = 9, OP_PK_FMA = 10, OP_PK_MUL = 11, OP_PK_ADD = 12 };

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : );
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x));
#define CNS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(mask));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define SQR(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define MED(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
// round 6: the packed forms (two f32 operations per lane and instruction; operands are aligned register pairs).  Four independent accumulator pairs, eight instructions per REP8.
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(i) & 3]) : "v"(px), "v"(py));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) & 3]) : "v"(py));
#define PKA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i) & 3]) : "v"(py));

template <int OP>
__global__ void __launch_bounds__(1024) k_probe(float* out, unsigned long long* stamps, int trips, float x, float y) {
    extern __shared__ float pad[];
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = x + (float)(threadIdx.x + i);
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[4], px = {x, x}, py = {y, y};
    for (int i = 0; i < 4; i++) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    if (threadIdx.x == 9999) pad[0] = x;
    const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)trips;      // an SGPR pair
    if (OP == OP_CNDMASK) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(y) : "vcc");      // vcc written once, before the loop
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if (OP == OP_FMA) { REP8(FMA) }
            else if (OP == OP_RCP) { REP8(RCP) }
            else if (OP == OP_CNDMASK) { REP8(CND) }
            else if (OP == OP_MOV) { REP8(MOV) }
            else if (OP == OP_CNDMASK_SGPR) { REP8(CNS) }
            else if (OP == OP_EXP) { REP8(EXP) }
            else if (OP == OP_SQRT) { REP8(SQR) }
            else if (OP == OP_MED3) { REP8(MED) }
            else if (OP == OP_MUL) { REP8(MUL) }
            else if (OP == OP_PK_FMA) { REP8(PKF) }
            else if (OP == OP_PK_MUL) { REP8(PKM) }
            else if (OP == OP_PK_ADD) { REP8(PKA) }
            else { FMA(0) FMA(1) FMA(2) RCP(3) FMA(4) FMA(5) FMA(6) RCP(7) }      // three full-rate instructions per transcendental
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    if (s == 1234.5f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP> static int run(const char* name, int cus, int trips, float* out, unsigned long long* stamps) {
    for (int W = 1; W <= 4; W++) {
        const int threads = 256 * W, waves = cus * 4 * W;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; rep++) {      // the last repetition is reported (clocks settled)
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe<OP>, dim3(cus), dim3(threads), 120 * 1024, 0, out, stamps, trips, 1.0001f, 0.9999f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * waves);
        CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc(waves), clk(waves);
        for (int w = 0; w < waves; w++) { cyc[w] = (double)h[2 * w]; clk[w] = (double)h[2 * w] / ((double)h[2 * w + 1] / 100.0e6) / 1e9; }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        const double insts = (double)trips * 256.0, medCyc = cyc[waves / 2];
        const double perSimdCycle = W * insts / medCyc, ghz = clk[waves / 2];
        // by the launch's own duration (HIP events): what the SIMD sustains over the whole launch, start-up skew of the waves included
        const double wallPerSimd = W * insts / (ms * 1e-3), wallPerCycle = wallPerSimd / (ghz * 1e9);
        std::printf("%-22s waves/SIMD %d | in-loop: %5.2f cycles/inst/wave, one wave-inst per %4.2f cycles per SIMD | by launch time %7.3f ms: one per %4.2f cycles per SIMD, chip %5.3f T wave-insts/s | clock %4.2f GHz\n",
                    name, W, medCyc / insts, 1.0 / perSimdCycle, ms, 1.0 / wallPerCycle, wallPerSimd * cus * 4 / 1e12, ghz);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float* out; unsigned long long* stamps;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&stamps, (size_t)cus * 16 * 16));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_RCP>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_CNDMASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_MOV>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_MUL>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_MED3>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_CNDMASK_SGPR>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_SQRT>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_PK_FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_PK_MUL>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<OP_PK_ADD>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    const int trips = 4000;      // ~1 M instructions per wave: 1-4 ms per launch
    if (run<OP_FMA>("v_fma_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_RCP>("v_rcp_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_MIX>("3 v_fma : 1 v_rcp", cus, trips, out, stamps)) return 1;
    if (run<OP_CNDMASK>("v_cndmask_b32", cus, trips, out, stamps)) return 1;
    if (run<OP_MOV>("v_mov_b32", cus, trips, out, stamps)) return 1;
    if (run<OP_MUL>("v_mul_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_MED3>("v_med3_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_CNDMASK_SGPR>("v_cndmask_b32 (sgpr)", cus, trips, out, stamps)) return 1;
    if (run<OP_EXP>("v_exp_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_SQRT>("v_sqrt_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_PK_FMA>("v_pk_fma_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_PK_MUL>("v_pk_mul_f32", cus, trips, out, stamps)) return 1;
    if (run<OP_PK_ADD>("v_pk_add_f32", cus, trips, out, stamps)) return 1;
    return 0;
}
