// valu_rate.hip -- issue rate of the VALU instructions the env kernels are made of, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.  Prints cycles per
// wave-instruction per SIMD (wall time x clock / instructions issued per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 4096, UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8]; float f[8]; double d[8]; uint64_t q[8];
    const float fs = __builtin_bit_cast(float, 0x3f800001u + (seed & 1u));          // wave-uniform: lives in a scalar register
    const uint64_t msk = 0x5555555555555555ull + seed;
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; f[i] = (float)a[i] * 1e-3f; d[i] = (double)f[i]; q[i] = a[i]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = u & 7;
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "s"(0xD2511F53u), "v"(a[i]) : "vcc");
            if constexpr (OP == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(seed));
            if constexpr (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if constexpr (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if constexpr (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : );
            if constexpr (OP == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc");
            if constexpr (OP == 7) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if constexpr (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if constexpr (OP == 9) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 10) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if constexpr (OP == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if constexpr (OP == 12) asm volatile("s_add_u32 %0, %0, %1" : "+s"(seed) : "s"(seed) : "scc");
            if constexpr (OP == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 21) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 22) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
            if constexpr (OP == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 24) asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 25) asm volatile("v_bfe_u32 %0, %1, 8, 23" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 26) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 27) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 28) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 29) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7]));
            if constexpr (OP == 30) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 31) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 32) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]) : "vcc");
            if constexpr (OP == 33) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
            if constexpr (OP == 34) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "s"(seed));
            if constexpr (OP == 35) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if constexpr (OP == 36) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if constexpr (OP == 37) asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if constexpr (OP == 38) asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]), "v"(a[(i + 3) & 7]));
            if constexpr (OP == 39) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            // round 4: operand kinds -- does a scalar register, a 32-bit literal or an inline constant as a source change the rate?
            if constexpr (OP == 40) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "s"(fs));                       // VOP2, SGPR src0
            if constexpr (OP == 41) asm volatile("v_mul_f32 %0, 0x3ecccccd, %0" : "+v"(f[i]));                          // VOP2, literal
            if constexpr (OP == 42) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(f[i]));                                 // VOP2, inline constant
            if constexpr (OP == 43) asm volatile("v_fmac_f32 %0, 0x3ecccccd, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));  // VOP2 fmac, literal
            if constexpr (OP == 44) asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "s"(fs)); // VOP2 fmac, SGPR
            if constexpr (OP == 45) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "s"(fs));   // VOP3 fma, SGPR
            if constexpr (OP == 46) asm volatile("v_add_f32 %0, 0x42f60000, %0" : "+v"(f[i]));                          // VOP2 add, literal
            if constexpr (OP == 47) asm volatile("v_and_b32 %0, 0x3fff0, %0" : "+v"(a[i]));                             // VOP2 and, literal
            if constexpr (OP == 48) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));   // all VGPR
            if constexpr (OP == 49) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(a[(i + 1) & 7]), "v"(a[i]) : "vcc");            // all VGPR
            if constexpr (OP == 51) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 52) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 53) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(msk));                        // VOP3, SGPR-pair mask
            if constexpr (OP == 54) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if constexpr (OP == 55) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 56) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 57) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if constexpr (OP == 58) asm volatile("v_add_f32 %0, 0.5, %0" : "+v"(f[i]));                                 // inline constant
            if constexpr (OP == 60) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if constexpr (OP == 61) asm volatile("v_alignbit_b32 %0, %1, %2, 8" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if constexpr (OP == 62) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]), "v"(a[(i + 3) & 7]));
            if constexpr (OP == 63) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 64) asm volatile("v_ashrrev_i32 %0, 3, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 65) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if constexpr (OP == 66) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if constexpr (OP == 67) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            // dependent chain of the wide multiply (Philox round structure: mad -> use hi)
            if constexpr (OP == 13) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[0]) : "s"(0xD2511F53u), "v"(a[0]) : "vcc"); a[0] = (uint32_t)(q[0] >> 32); }
            if constexpr (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[0]) : "v"(f[1]));   // dependent fma chain
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (uint32_t)f[i] + (uint32_t)d[i] + (uint32_t)q[i];
    if (r == 0x12345678u) out[threadIdx.x] = r + seed;
}

template <int OP>
static int run(const char *name, int ncu, double mhz, uint32_t *out)
{
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD
        const int blocks = ncu * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = 5.0 * ITERS * UNROLL * wps;
        printf("  wps=%d: %6.2f cyc/instr/SIMD", wps, ms * 1e-3 * mhz * 1e6 / instr_per_simd);
    }
    printf("\n");
    return 0;
}

__global__ void clock_probe(unsigned long long *o)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < 2000000; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = t1 - t0; o[1] = r1 - r0; }
    if (x == 1.2345f) o[2] = 1;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
    printf("%s: %d CUs, clock %.0f MHz (cycles computed at that clock)\n", prop.name, ncu, mhz);
    uint32_t *out; CHECK(hipMalloc(&out, 4096));
    {
        unsigned long long *cp, h[2];
        CHECK(hipMalloc(&cp, 64));
        hipLaunchKernelGGL(clock_probe, dim3(ncu * 4), dim3(256), 0, 0, cp);
        CHECK(hipMemcpy(h, cp, 16, hipMemcpyDeviceToHost));
        printf("in-kernel clock under an all-CU VALU load: %.0f MHz (s_memtime / s_memrealtime x 100 MHz)\n", (double)h[0] / (double)h[1] * 100.0);
    }
    if (!getenv("VALU_RATE_NEW_ONLY")) {
    run<0>("v_fma_f32 (indep)", ncu, mhz, out);
    run<14>("v_fma_f32 (dependent)", ncu, mhz, out);
    run<1>("v_mad_u64_u32 (indep)", ncu, mhz, out);
    run<13>("v_mad_u64_u32 (dep chain)", ncu, mhz, out);
    run<2>("v_bitop3_b32", ncu, mhz, out);
    run<9>("v_xor_b32", ncu, mhz, out);
    run<5>("v_cndmask_b32", ncu, mhz, out);
    run<6>("v_cmp_lt_f32", ncu, mhz, out);
    run<7>("v_cvt_f32_u32", ncu, mhz, out);
    run<3>("v_add_f64", ncu, mhz, out);
    run<11>("v_mul_f64", ncu, mhz, out);
    run<4>("v_fma_f64", ncu, mhz, out);
    run<10>("v_cvt_f64_f32", ncu, mhz, out);
    run<8>("v_pk_fma_f32", ncu, mhz, out);
    run<12>("s_add_u32", ncu, mhz, out);
    run<20>("v_add_f32", ncu, mhz, out);
    run<21>("v_mul_f32", ncu, mhz, out);
    run<22>("v_fmac_f32 (VOP2)", ncu, mhz, out);
    run<29>("v_fma_f32 3 distinct srcs", ncu, mhz, out);
    run<23>("v_and_b32", ncu, mhz, out);
    run<34>("v_xor_b32 v,s", ncu, mhz, out);
    run<24>("v_lshrrev_b32", ncu, mhz, out);
    run<25>("v_bfe_u32", ncu, mhz, out);
    run<26>("v_add_u32", ncu, mhz, out);
    run<37>("v_lshl_add_u32", ncu, mhz, out);
    run<38>("v_xad_u32", ncu, mhz, out);
    run<27>("v_max_f32", ncu, mhz, out);
    run<33>("v_med3_f32", ncu, mhz, out);
    run<28>("v_mov_b32", ncu, mhz, out);
    run<30>("v_mul_lo_u32", ncu, mhz, out);
    run<31>("v_mul_hi_u32", ncu, mhz, out);
    run<32>("v_cmp+v_cndmask (pair, per 2)", ncu, mhz, out);
    run<35>("v_pk_mul_f32", ncu, mhz, out);
    run<36>("v_pk_add_f32", ncu, mhz, out);
    run<39>("v_cvt_f32_f64", ncu, mhz, out);
    }
    if (getenv("VALU_RATE_OLD_ONLY")) return 0;
    printf("-- round 4: operand kinds\n");
    run<21>("v_mul_f32 v,v (anchor)", ncu, mhz, out);
    run<40>("v_mul_f32 v,s", ncu, mhz, out);
    run<41>("v_mul_f32 v,literal", ncu, mhz, out);
    run<42>("v_mul_f32 v,inline 0.5", ncu, mhz, out);
    run<43>("v_fmac_f32 literal", ncu, mhz, out);
    run<44>("v_fmac_f32 sgpr", ncu, mhz, out);
    run<45>("v_fma_f32 sgpr", ncu, mhz, out);
    run<46>("v_add_f32 literal", ncu, mhz, out);
    run<58>("v_add_f32 inline", ncu, mhz, out);
    run<54>("v_sub_f32", ncu, mhz, out);
    run<47>("v_and_b32 literal", ncu, mhz, out);
    run<48>("v_bitop3_b32 v,v,v", ncu, mhz, out);
    run<49>("v_mad_u64_u32 v,v", ncu, mhz, out);
    run<51>("v_maximum3_f32", ncu, mhz, out);
    run<52>("v_min_f32", ncu, mhz, out);
    run<53>("v_cndmask_b32 sgpr mask", ncu, mhz, out);
    run<55>("v_lshlrev_b32", ncu, mhz, out);
    run<64>("v_ashrrev_i32", ncu, mhz, out);
    run<56>("v_or_b32", ncu, mhz, out);
    run<57>("v_and_or_b32", ncu, mhz, out);
    run<63>("v_sub_u32", ncu, mhz, out);
    run<60>("v_cvt_f32_ubyte0", ncu, mhz, out);
    run<61>("v_alignbit_b32", ncu, mhz, out);
    run<62>("v_perm_b32", ncu, mhz, out);
    run<65>("v_mul_u32_u24", ncu, mhz, out);
    run<66>("v_mad_u32_u24", ncu, mhz, out);
    run<67>("v_mul_hi_u32_u24", ncu, mhz, out);
    return 0;
}
