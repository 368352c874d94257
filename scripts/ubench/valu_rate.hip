// microbenchmark: issue rate of VALU ops on gfx950 (inline asm, 8 independent accumulators)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R2(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#define R3(OP) asm volatile(OP " %0, %0, %8, %8\n" OP " %1, %1, %8, %8\n" OP " %2, %2, %8, %8\n" OP " %3, %3, %8, %8\n" OP " %4, %4, %8, %8\n" OP " %5, %5, %8, %8\n" OP " %6, %6, %8, %8\n" OP " %7, %7, %8, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#define R2P(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
template <int OP> __global__ void __launch_bounds__(256) k(int* out, int n, int seed)
{
    int a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int c = seed | 0x10003;
    double p0 = seed, p1 = seed + 1, p2 = seed + 2, p3 = seed + 3, pc = 1.0;
    for (int it = 0; it < n; ++it)
    {
        if (OP == 0) { R2("v_add_u32") R2("v_add_u32") R2("v_add_u32") R2("v_add_u32") }
        if (OP == 1) { R2("v_sub_u32") R2("v_sub_u32") R2("v_sub_u32") R2("v_sub_u32") }
        if (OP == 2) { R2("v_max_i32") R2("v_max_i32") R2("v_max_i32") R2("v_max_i32") }
        if (OP == 3) { R2("v_min_u32") R2("v_min_u32") R2("v_min_u32") R2("v_min_u32") }
        if (OP == 4) { R2("v_max_u32") R2("v_max_u32") R2("v_max_u32") R2("v_max_u32") }
        if (OP == 5) { R2("v_and_b32") R2("v_and_b32") R2("v_and_b32") R2("v_and_b32") }
        if (OP == 6) { R2("v_or_b32") R2("v_or_b32") R2("v_or_b32") R2("v_or_b32") }
        if (OP == 7) { R2("v_xor_b32") R2("v_xor_b32") R2("v_xor_b32") R2("v_xor_b32") }
        if (OP == 8) { R2("v_lshlrev_b32") R2("v_lshlrev_b32") R2("v_lshlrev_b32") R2("v_lshlrev_b32") }
        if (OP == 9) { R2("v_add_f32") R2("v_add_f32") R2("v_add_f32") R2("v_add_f32") }
        if (OP == 10) { R2("v_max_f32") R2("v_max_f32") R2("v_max_f32") R2("v_max_f32") }
        if (OP == 11) { R2("v_pk_add_f16") R2("v_pk_add_f16") R2("v_pk_add_f16") R2("v_pk_add_f16") }
        if (OP == 12) { R2("v_pk_max_f16") R2("v_pk_max_f16") R2("v_pk_max_f16") R2("v_pk_max_f16") }
        if (OP == 13) { R2("v_pk_add_u16") R2("v_pk_add_u16") R2("v_pk_add_u16") R2("v_pk_add_u16") }
        if (OP == 14) { R2("v_pk_max_i16") R2("v_pk_max_i16") R2("v_pk_max_i16") R2("v_pk_max_i16") }
        if (OP == 15) { asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); }
        if (OP == 16) { asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); }
        if (OP == 17) { R2("v_max_i16") R2("v_max_i16") R2("v_max_i16") R2("v_max_i16") }
        if (OP == 18) { R2("v_add_u16") R2("v_add_u16") R2("v_add_u16") R2("v_add_u16") }
        if (OP == 19) { R2("v_max_f16") R2("v_max_f16") R2("v_max_f16") R2("v_max_f16") }
        if (OP == 20) { R2("v_add_f16") R2("v_add_f16") R2("v_add_f16") R2("v_add_f16") }
        if (OP == 21) { R2("v_mul_f32") R2("v_mul_f32") R2("v_mul_f32") R2("v_mul_f32") }
        if (OP == 22) { R3("v_max3_i32") R3("v_max3_i32") R3("v_max3_i32") R3("v_max3_i32") }
        if (OP == 23) { R3("v_max3_f32") R3("v_max3_f32") R3("v_max3_f32") R3("v_max3_f32") }
        if (OP == 24) { R3("v_add3_u32") R3("v_add3_u32") R3("v_add3_u32") R3("v_add3_u32") }
        if (OP == 25) { R3("v_lshl_add_u32") R3("v_lshl_add_u32") R3("v_lshl_add_u32") R3("v_lshl_add_u32") }
        if (OP == 26) { R3("v_lshl_or_b32") R3("v_lshl_or_b32") R3("v_lshl_or_b32") R3("v_lshl_or_b32") }
        if (OP == 27) { R3("v_and_or_b32") R3("v_and_or_b32") R3("v_and_or_b32") R3("v_and_or_b32") }
        if (OP == 28) { R3("v_bfi_b32") R3("v_bfi_b32") R3("v_bfi_b32") R3("v_bfi_b32") }
        if (OP == 29) { R3("v_bfe_u32") R3("v_bfe_u32") R3("v_bfe_u32") R3("v_bfe_u32") }
        if (OP == 30) { R3("v_mad_u32_u24") R3("v_mad_u32_u24") R3("v_mad_u32_u24") R3("v_mad_u32_u24") }
        if (OP == 31) { R3("v_mad_i32_i24") R3("v_mad_i32_i24") R3("v_mad_i32_i24") R3("v_mad_i32_i24") }
        if (OP == 32) { R3("v_fma_f32") R3("v_fma_f32") R3("v_fma_f32") R3("v_fma_f32") }
        if (OP == 33) { R3("v_pk_fma_f16") R3("v_pk_fma_f16") R3("v_pk_fma_f16") R3("v_pk_fma_f16") }
        if (OP == 34) { R3("v_pk_mad_u16") R3("v_pk_mad_u16") R3("v_pk_mad_u16") R3("v_pk_mad_u16") }
        if (OP == 35) { R3("v_med3_i32") R3("v_med3_i32") R3("v_med3_i32") R3("v_med3_i32") }
        if (OP == 36) { R3("v_alignbit_b32") R3("v_alignbit_b32") R3("v_alignbit_b32") R3("v_alignbit_b32") }
        if (OP == 37) { R3("v_perm_b32") R3("v_perm_b32") R3("v_perm_b32") R3("v_perm_b32") }
        if (OP == 38) { R3("v_max3_i16") R3("v_max3_i16") R3("v_max3_i16") R3("v_max3_i16") }
        if (OP == 39) { R3("v_max3_f16") R3("v_max3_f16") R3("v_max3_f16") R3("v_max3_f16") }
        if (OP == 40) { R3("v_mad_u16") R3("v_mad_u16") R3("v_mad_u16") R3("v_mad_u16") }
        if (OP == 41) { R3("v_fma_f16") R3("v_fma_f16") R3("v_fma_f16") R3("v_fma_f16") }
        if (OP == 42) { asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc)); }
        if (OP == 43) { R3("v_or3_b32") R3("v_or3_b32") R3("v_or3_b32") R3("v_or3_b32") }
        if (OP == 44) { R3("v_xad_u32") R3("v_xad_u32") R3("v_xad_u32") R3("v_xad_u32") }
        if (OP == 45) { R3("v_add_lshl_u32") R3("v_add_lshl_u32") R3("v_add_lshl_u32") R3("v_add_lshl_u32") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(p0 + p1 + p2 + p3);
}
template <int OP> void run(int* d, const char* name)
{
    const int n = 4000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * n * 32.0;
    printf("%-18s %.3f ms  = %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, 1024.0 * 2.4e9 * ms * 1e-3 / insts);
}
int main()
{
    int* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(int) * 4);
    run<0>(d, "v_add_u32");
    run<1>(d, "v_sub_u32");
    run<2>(d, "v_max_i32");
    run<3>(d, "v_min_u32");
    run<4>(d, "v_max_u32");
    run<5>(d, "v_and_b32");
    run<6>(d, "v_or_b32");
    run<7>(d, "v_xor_b32");
    run<8>(d, "v_lshlrev_b32");
    run<9>(d, "v_add_f32");
    run<10>(d, "v_max_f32");
    run<11>(d, "v_pk_add_f16");
    run<12>(d, "v_pk_max_f16");
    run<13>(d, "v_pk_add_u16");
    run<14>(d, "v_pk_max_i16");
    run<15>(d, "v_pk_add_f32");
    run<16>(d, "v_pk_mul_f32");
    run<17>(d, "v_max_i16");
    run<18>(d, "v_add_u16");
    run<19>(d, "v_max_f16");
    run<20>(d, "v_add_f16");
    run<21>(d, "v_mul_f32");
    run<22>(d, "v_max3_i32");
    run<23>(d, "v_max3_f32");
    run<24>(d, "v_add3_u32");
    run<25>(d, "v_lshl_add_u32");
    run<26>(d, "v_lshl_or_b32");
    run<27>(d, "v_and_or_b32");
    run<28>(d, "v_bfi_b32");
    run<29>(d, "v_bfe_u32");
    run<30>(d, "v_mad_u32_u24");
    run<31>(d, "v_mad_i32_i24");
    run<32>(d, "v_fma_f32");
    run<33>(d, "v_pk_fma_f16");
    run<34>(d, "v_pk_mad_u16");
    run<35>(d, "v_med3_i32");
    run<36>(d, "v_alignbit_b32");
    run<37>(d, "v_perm_b32");
    run<38>(d, "v_max3_i16");
    run<39>(d, "v_max3_f16");
    run<40>(d, "v_mad_u16");
    run<41>(d, "v_fma_f16");
    run<42>(d, "v_pk_fma_f32");
    run<43>(d, "v_or3_b32");
    run<44>(d, "v_xad_u32");
    run<45>(d, "v_add_lshl_u32");
    return 0;
}
