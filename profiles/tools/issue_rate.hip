// Measures, on the GPU it runs on, how many cycles one SIMD of a gfx950 CU needs per
// wave64 instruction of several classes, with 1, 2 and 4 waves resident per SIMD.
// Every wave runs a loop of 64 independent instructions of one class (16 chains, so
// neither latency nor register dependencies bound it) and stamps itself with the shader
// clock (s_memtime) and the 100 MHz real-time clock (s_memrealtime); the kernel is also
// timed with HIP events.  cycles per wave-instruction per SIMD =
//   wave's shader-clock cycles / (instructions per wave * waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 profiles/tools/issue_rate.hip -o profiles/tools/issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_ADD, OP_MAD24, OP_MULLO, OP_FMA64, OP_RCP32, OP_CVT, OP_FMA32, OP_CNDMASK, OP_ASHR, OP_DSMIN64, OP_DSREAD128, OP_MULHI, OP_LSHL64, OP_BFE, OP_ADD3, OP_MIN3, OP_MIX, OP_CND64, OP_CMP64, OP_SDWA, OP_ADDC, OP_MAD64, OP_LSHLADD64, OP_CVTF64, OP_RCP64, OP_CVTI64, OP_MULF64, OP_MULI24, OP_MULF32, OP_CVTIF32, OP_MAXI32, OP_MBCNT, OP_LSHLADD, OP_ANDOR, OP_READLANE, OP_SALU, OP_SEL4_VCC, OP_SEL4_E64, OP_SEL4_VCC_AND, OP_ANDLIT, OP_CMP32, OP_P1, OP_P2, OP_P3, OP_P4, OP_P5, OP_P6, OP_COUNT };
static const char *names[] = {"v_add_u32", "v_mad_u32_u24", "v_mul_lo_u32", "v_fma_f64", "v_rcp_f32", "v_cvt_f32_i32", "v_fma_f32",
                              "v_cndmask_b32", "v_ashrrev_i32", "ds_min_i64", "ds_read_b128", "v_mul_hi_u32", "v_lshlrev_b64",
                              "v_bfe_u32", "v_add3_u32", "v_min3_i32", "mix add/cndmask/cmp", "v_cndmask_b32_e64 (sgpr mask)", "v_cmp_lt_i32_e64 (sgpr dst)", "v_sub_u32_sdwa", "v_addc_co_u32_e64", "v_mad_u64_u32", "v_lshl_add_u64", "v_cvt_f64_i32", "v_rcp_f64", "v_cvt_i32_f64", "v_mul_f64", "v_mul_i32_i24", "v_mul_f32", "v_cvt_i32_f32", "v_max_i32", "v_mbcnt_lo+hi", "v_lshl_add_u32", "v_and_or_b32", "v_readlane_b32", "s_add_u32 x4",
                              "cmp_e32 + 4 cndmask_e32 on vcc", "cmp_e64 + 4 cndmask_e64 on sgprs", "cmp_e32 + s_and vcc + 4 cndmask_e32", "v_and_b32 with a 32-bit literal", "v_cmp_lt_i32_e32 (vcc)", "cmp_e32; cndmask_e32", "cmp_e32; s_nop 1; cndmask_e32", "cmp_e32; add; add; cndmask_e32", "cndmask_e64 with vcc as mask", "s_mov vcc; s_nop 1; cndmask_e32", "cmp_e32; s_nop 1; cndmask_e32 x2"};

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(unsigned long long *stamps, int iters, int seed)
{
    __shared__ long long lds[2048];
    const int t = threadIdx.x;
    for (int i = t; i < 2048; i += 256) lds[i] = 0x7fffffffffffffffll - i;
    __syncthreads();
    unsigned a[16];
    double d[16];
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + i * 7 + t; d[i] = 1.0 + i + seed; f[i] = 1.0f + i + seed; }
    const unsigned b = seed | 3u;
    unsigned long long smask = 0x5555555555555555ull * (unsigned)seed, sm[4] = {1, 2, 3, 4};
    unsigned ss[4] = {1, 2, 3, 4};
    const double db = 1.0000001;
    const float fb = 1.0000001f;
    const unsigned laddr = (unsigned)(size_t)(&lds[0]) + (unsigned)(t & 63) * 8u + (unsigned)(t >> 6) * 512u * 8u;
    const unsigned laddr16 = (unsigned)(size_t)(&lds[0]) + (unsigned)(t & 63) * 16u;
    unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db));
                if (OP == OP_RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if (OP == OP_CVT) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == OP_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fb));
                if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == OP_ASHR) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
                if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
                if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MIN3) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_LSHL64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[i]));
                if (OP == OP_DSMIN64) asm volatile("ds_min_i64 %0, %1 offset:%2" : : "v"(laddr), "v"(d[i]), "n"((i & 7) * 512 + 0) : "memory");
                if (OP == OP_DSREAD128) {
                    int4 q;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(laddr16), "n"(i * 1024) : "memory");
                    if (i == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; }
                }

                if (OP == OP_CND64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(smask));
                if (OP == OP_CMP64) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(sm[i & 3]) : "v"(a[i]), "v"(b));
                if (OP == OP_SDWA) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a[i]) : "v"(b));
                if (OP == OP_ADDC) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(a[i]), "+s"(sm[i & 3]) : "v"(b));
                if (OP == OP_MAD64) asm volatile("v_mad_u64_u32 %0, %1, %2, %2, %0" : "+v"(d[i]), "=s"(sm[i & 3]) : "v"(b));
                if (OP == OP_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(d[i]));
                if (OP == OP_CVTF64) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if (OP == OP_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
                if (OP == OP_CVTI64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                if (OP == OP_MULF64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
                if (OP == OP_MULI24) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MULF32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
                if (OP == OP_CVTIF32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (OP == OP_MAXI32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0" : "=v"(a[i]) : "v"(b));
                if (OP == OP_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_READLANE) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(ss[i & 3]) : "v"(a[i]));
                if (OP == OP_SALU) asm volatile("s_add_u32 %0, %0, 3" : "+s"(ss[i & 3]));
                if (OP == OP_SEL4_VCC && (i & 3) == 0)
                    asm volatile("v_cmp_lt_i32_e32 vcc, %0, %4\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\t"
                                 "v_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc"
                                 : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(b) : "vcc");
                if (OP == OP_SEL4_E64 && (i & 3) == 0)
                    asm volatile("v_cmp_lt_i32_e64 %5, %0, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %4, %5\n\tv_cndmask_b32_e64 %1, %1, %4, %5\n\t"
                                 "v_cndmask_b32_e64 %2, %2, %4, %5\n\tv_cndmask_b32_e64 %3, %3, %4, %5"
                                 : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(b), "s"(sm[(i >> 2) & 3]));
                if (OP == OP_SEL4_VCC_AND && (i & 3) == 0)
                    asm volatile("v_cmp_lt_i32_e32 vcc, %0, %4\n\ts_and_b64 vcc, vcc, %5\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\t"
                                 "v_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc"
                                 : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(b), "s"(smask) : "vcc");
                if (OP == OP_ANDLIT) asm volatile("v_and_b32_e32 %0, 0x12345, %0" : "+v"(a[i]));
                if (OP == OP_CMP32) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == OP_P1) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == OP_P2) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == OP_P3) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %2\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %1, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[i]), "+v"(a[(i + 5) & 15]) : "v"(b) : "vcc");
                if (OP == OP_P4) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == OP_P5) asm volatile("s_mov_b64 vcc, %2\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "s"(smask) : "vcc");
                if (OP == OP_P6) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %2\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(a[i]), "+v"(a[(i + 5) & 15]) : "v"(b) : "vcc");
                if (OP == OP_MIX) {
                    asm volatile("v_add_u32 %0, %0, %1\n\tv_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
                }
            }
        }
        if (OP == OP_DSMIN64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long c1 = __builtin_readcyclecounter();
    unsigned long long r1 = wall_clock64();
    unsigned s = 0;
    double ds = 0;
    float fs = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { s += a[i]; ds += d[i]; fs += f[i]; }
    s += (unsigned)(sm[0] + sm[1] + sm[2] + sm[3]) + ss[0] + ss[1] + ss[2] + ss[3];
    if (s == 0x12345 && ds == 3.0 && fs == 2.0f) stamps[0] = lds[t];     // keeps everything live
    if ((t & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * 4 + (t >> 6)) * 2;
        stamps[1 + w] = c1 - c0;
        stamps[2 + w] = r1 - r0;
    }
}

template <int OP>
static void run(int wps, int iters, int ncu, unsigned long long *dev, std::vector<unsigned long long> &host)
{
    const int blocks = ncu * wps;
    const double per_inst = OP == OP_MIX ? 4 : (OP == OP_SEL4_VCC || OP == OP_SEL4_E64 || OP == OP_SEL4_VCC_AND) ? 1.25 : (OP == OP_P1 || OP == OP_P2) ? 2 : OP == OP_P3 ? 4 : OP == OP_P6 ? 3 : 1;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, dev, 10, 1);       // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, dev, iters, 1);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(host.data(), dev, (1 + (size_t)blocks * 8) * 8, hipMemcpyDeviceToHost));
    double sc = 0, sr = 0;
    unsigned long long mx = 0;
    const size_t nw = (size_t)blocks * 4;
    for (size_t w = 0; w < nw; w++) { sc += (double)host[1 + 2 * w]; sr += (double)host[2 + 2 * w]; if (host[1 + 2 * w] > mx) mx = host[1 + 2 * w]; }
    const double ninst = (double)iters * 64 * per_inst;
    const double cyc = sc / nw, rt = sr / nw;                 // per wave
    // shader clock from the two counters (the real-time counter runs at 100 MHz)
    const double mhz = rt > 0 ? cyc / rt * 100.0 : 0.0;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instruction_per_simd\": %.3f, \"wave_cycles_avg\": %.0f, \"wave_cycles_max\": %llu, "
           "\"shader_clock_mhz\": %.0f, \"kernel_ms\": %.4f, \"by_kernel_time_at_that_clock\": %.3f}\n",
           names[OP], wps, cyc / (ninst * wps), cyc, mx, mhz, ms, ms * 1e-3 * mhz * 1e6 / (ninst * wps));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int OP>
static void sweep(int iters, int ncu, unsigned long long *dev, std::vector<unsigned long long> &host)
{
    for (int wps : {1, 2, 4}) run<OP>(wps, iters, ncu, dev, host);
}

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    fprintf(stderr, "%s: %d CUs, clockRate %d kHz\n", p.gcnArchName, ncu, p.clockRate);
    unsigned long long *dev;
    const size_t n = 1 + (size_t)ncu * 4 * 8;
    CHECK(hipMalloc(&dev, n * 8));
    std::vector<unsigned long long> host(n);
    sweep<OP_SEL4_VCC>(iters, ncu, dev, host);
    sweep<OP_SEL4_E64>(iters, ncu, dev, host);
    sweep<OP_SEL4_VCC_AND>(iters, ncu, dev, host);
    sweep<OP_ANDLIT>(iters, ncu, dev, host);
    sweep<OP_CMP32>(iters, ncu, dev, host);
    sweep<OP_CNDMASK>(iters, ncu, dev, host);
    sweep<OP_P1>(iters, ncu, dev, host);
    sweep<OP_P2>(iters, ncu, dev, host);
    sweep<OP_P3>(iters, ncu, dev, host);
    sweep<OP_P4>(iters, ncu, dev, host);
    sweep<OP_P5>(iters, ncu, dev, host);
    sweep<OP_P6>(iters, ncu, dev, host);
    if (argc > 2) { CHECK(hipFree(dev)); return 0; }       // a second argument: the select-group cases only
    sweep<OP_ADD>(iters, ncu, dev, host);
    sweep<OP_MAD24>(iters, ncu, dev, host);
    sweep<OP_MULLO>(iters, ncu, dev, host);
    sweep<OP_MULHI>(iters, ncu, dev, host);
    sweep<OP_FMA64>(iters, ncu, dev, host);
    sweep<OP_RCP32>(iters, ncu, dev, host);
    sweep<OP_CVT>(iters, ncu, dev, host);
    sweep<OP_FMA32>(iters, ncu, dev, host);
    sweep<OP_CNDMASK>(iters, ncu, dev, host);
    sweep<OP_ASHR>(iters, ncu, dev, host);
    sweep<OP_BFE>(iters, ncu, dev, host);
    sweep<OP_ADD3>(iters, ncu, dev, host);
    sweep<OP_MIN3>(iters, ncu, dev, host);
    sweep<OP_LSHL64>(iters, ncu, dev, host);
    sweep<OP_MIX>(iters, ncu, dev, host);
    sweep<OP_CND64>(iters, ncu, dev, host);
    sweep<OP_CMP64>(iters, ncu, dev, host);
    sweep<OP_SDWA>(iters, ncu, dev, host);
    sweep<OP_ADDC>(iters, ncu, dev, host);
    sweep<OP_MAD64>(iters, ncu, dev, host);
    sweep<OP_LSHLADD64>(iters, ncu, dev, host);
    sweep<OP_CVTF64>(iters, ncu, dev, host);
    sweep<OP_RCP64>(iters, ncu, dev, host);
    sweep<OP_CVTI64>(iters, ncu, dev, host);
    sweep<OP_MULF64>(iters, ncu, dev, host);
    sweep<OP_MULI24>(iters, ncu, dev, host);
    sweep<OP_MULF32>(iters, ncu, dev, host);
    sweep<OP_CVTIF32>(iters, ncu, dev, host);
    sweep<OP_MAXI32>(iters, ncu, dev, host);
    sweep<OP_MBCNT>(iters, ncu, dev, host);
    sweep<OP_LSHLADD>(iters, ncu, dev, host);
    sweep<OP_ANDOR>(iters, ncu, dev, host);
    sweep<OP_READLANE>(iters, ncu, dev, host);
    sweep<OP_SALU>(iters, ncu, dev, host);

    sweep<OP_DSMIN64>(iters / 4, ncu, dev, host);
    sweep<OP_DSREAD128>(iters / 4, ncu, dev, host);
    CHECK(hipFree(dev));
    return 0;
}
