// development aid: VALU issue rate of gfx950 per instruction kind (wave-instructions per cycle per SIMD), by waves per SIMD.
// The row kernels and the X-drop kernel are priced against these numbers.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_old/valu_rates tools/ubench/valu_rates.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(x) x x x x x x x x
// eight independent chains per lane, one instruction each per round: no dependency stalls with >= 1 wave
#define BODY(INS)                                                                                  \
    asm volatile(INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7)                    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                 : "v"(b), "v"(c))

#define I_ADD(r) "v_add_u32 " #r ", " #r ", %8\n"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 1, %8\n"
#define I_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define I_PKADD(r) "v_pk_add_u16 " #r ", " #r ", %8\n"
#define I_PKMIN(r) "v_pk_min_u16 " #r ", " #r ", %8\n"
#define I_PKADDCL(r) "v_pk_add_i16 " #r ", " #r ", %8 clamp\n"
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define I_MIN(r) "v_min_u32 " #r ", " #r ", %8\n"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define I_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
#define I_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define I_CMP(r) "v_cmp_lt_u32 vcc, " #r ", %8\n"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 7\n"
#define I_LSHR(r) "v_lshrrev_b32 " #r ", 1, " #r "\n"
#define I_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define I_MOVDPP(r) "v_mov_b32_dpp " #r ", " #r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CNDMASK64(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, s[20:21]\n"
#define I_PKMAD(r) "v_pk_mad_i16 " #r ", " #r ", %8, %9\n"
#define I_BFI(r) "v_bfi_b32 " #r ", %9, " #r ", %8\n"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8\n"
#define I_OR(r) "v_or_b32 " #r ", " #r ", %8\n"
#define I_ANDLIT(r) "v_and_b32 " #r ", 0x00010001, " #r "\n"
#define I_PKMAXSEL(r) "v_pk_max_i16 " #r ", " #r ", " #r " op_sel:[0,1] op_sel_hi:[1,0]\n"
#define I_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, 16\n"
#define I_LSHLOR(r) "v_lshl_or_b32 " #r ", " #r ", 15, %8\n"
#define I_CMPI16(r) "v_cmp_gt_i16 vcc, " #r ", %8\n"
#define I_MOV(r) "v_mov_b32 " #r ", %8\n"
#define I_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define I_SUB(r) "v_sub_u32 " #r ", " #r ", %8\n"
#define I_MAXI32(r) "v_max_i32 " #r ", " #r ", %8\n"
#define I_READLANE(r) "v_readfirstlane_b32 s20, " #r "\n"

template <int OP>
__global__ void k(uint32_t* out, int iters, unsigned long long* cyc) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = blockIdx.x | 1u, c = 0x07060504u;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(BODY(I_ADD);) }
        else if (OP == 1) { REP8(BODY(I_XOR);) }
        else if (OP == 2) { REP8(BODY(I_LSHLADD);) }
        else if (OP == 3) { REP8(BODY(I_ANDOR);) }
        else if (OP == 4) { REP8(BODY(I_MAD24);) }
        else if (OP == 5) { REP8(BODY(I_MULLO);) }
        else if (OP == 6) { REP8(BODY(I_PKADD);) }
        else if (OP == 7) { REP8(BODY(I_PKMIN);) }
        else if (OP == 8) { REP8(BODY(I_PKADDCL);) }
        else if (OP == 9) { REP8(BODY(I_PERM);) }
        else if (OP == 10) { REP8(BODY(I_MIN);) }
        else if (OP == 11) { REP8(BODY(I_FMA);) }
        else if (OP == 12) { REP8(BODY(I_CNDMASK);) }
        else if (OP == 13) { REP8(BODY(I_CMP);) }
        else if (OP == 14) { REP8(BODY(I_BFE);) }
        else if (OP == 15) { REP8(BODY(I_LSHR);) }
        else if (OP == 16) { REP8(BODY(I_ADD3);) }
        else if (OP == 17) { REP8(BODY(I_MOVDPP);) }
        else if (OP == 18) { REP8(BODY(I_CNDMASK64);) }
        else if (OP == 19) { REP8(BODY(I_PKMAD);) }
        else if (OP == 20) { REP8(BODY(I_BFI);) }
        else if (OP == 21) { REP8(BODY(I_AND);) }
        else if (OP == 22) { REP8(BODY(I_OR);) }
        else if (OP == 23) { REP8(BODY(I_ANDLIT);) }
        else if (OP == 24) { REP8(BODY(I_PKMAXSEL);) }
        else if (OP == 25) { REP8(BODY(I_ALIGNBIT);) }
        else if (OP == 26) { REP8(BODY(I_LSHLOR);) }
        else if (OP == 27) { REP8(BODY(I_CMPI16);) }
        else if (OP == 28) { REP8(BODY(I_MOV);) }
        else if (OP == 29) { REP8(BODY(I_LSHL);) }
        else if (OP == 30) { REP8(BODY(I_SUB);) }
        else if (OP == 31) { REP8(BODY(I_MAXI32);) }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = (unsigned long long)(t1 - t0);
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345679u) out[0] = a0;
}

template <int OP>
void run(const char* name, uint32_t* d, unsigned long long* dc) {
    printf("%-22s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256, blocks = 256 * wps;          // wps waves per SIMD on every CU
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, threads>>>(d, 10, dc);
        hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(d, iters, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc = 0; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        const double ins_per_simd = (double)wps * iters * 64.0;    // wave-instructions per SIMD
        printf("  w/SIMD=%d: %.2f cyc/inst (clock64) %.2f (2.4GHz wall)", wps, (double)cyc / ins_per_simd, ms * 1e-3 * 2.4e9 / ins_per_simd);
    }
    printf("\n");
}

int main() {
    uint32_t* d; hipMalloc(&d, 64);
    unsigned long long* dc; hipMalloc(&dc, 8);
    run<0>("v_add_u32", d, dc);
    run<1>("v_xor_b32", d, dc);
    run<2>("v_lshl_add_u32", d, dc);
    run<3>("v_and_or_b32", d, dc);
    run<4>("v_mad_u32_u24", d, dc);
    run<5>("v_mul_lo_u32", d, dc);
    run<6>("v_pk_add_u16", d, dc);
    run<7>("v_pk_min_u16", d, dc);
    run<8>("v_pk_add_i16 clamp", d, dc);
    run<9>("v_perm_b32", d, dc);
    run<10>("v_min_u32", d, dc);
    run<11>("v_fma_f32", d, dc);
    run<12>("v_cndmask_b32", d, dc);
    run<13>("v_cmp_lt_u32", d, dc);
    run<14>("v_bfe_u32", d, dc);
    run<15>("v_lshrrev_b32", d, dc);
    run<16>("v_add3_u32", d, dc);
    run<17>("v_mov_b32_dpp", d, dc);
    run<18>("v_cndmask_b32 sgpr", d, dc);
    run<19>("v_pk_mad_i16", d, dc);
    run<20>("v_bfi_b32", d, dc);
    run<21>("v_and_b32", d, dc);
    run<22>("v_or_b32", d, dc);
    run<23>("v_and_b32 literal", d, dc);
    run<24>("v_pk_max_i16 op_sel", d, dc);
    run<25>("v_alignbit_b32", d, dc);
    run<26>("v_lshl_or_b32", d, dc);
    run<27>("v_cmp_gt_i16", d, dc);
    run<28>("v_mov_b32", d, dc);
    run<29>("v_lshlrev_b32", d, dc);
    run<30>("v_sub_u32", d, dc);
    run<31>("v_max_i32", d, dc);
    return 0;
}
