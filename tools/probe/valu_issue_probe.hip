// valu_issue_probe.hip -- how many cycles does a SIMD of gfx950 need per wave64 instruction of the kinds the compress
// kernel issues?  (bench.py charged 4 cycles per VALU wave-instruction; the micro-architecture
// guide measured 2 for v_fma_f32.  This settles it for the integer / LDS-permute / scalar mix of tamp_compress_kernel.)
//
// Method: 256-thread workgroups (one wavefront per SIMD), N of them per CU (N = 1, 2, 4, 7: the LDS request makes exactly N
// fit), as many as the device holds.  Every wavefront runs `iters` x 128 instructions of one kind -- eight INDEPENDENT
// chains, or ONE dependent chain -- between two s_memtime reads and records (start, end, HW_ID).  Per SIMD (waves grouped
// by XCC / SE / CU / SIMD id) the cost of a wave-instruction is
//         (latest end - earliest start) / (waves on that SIMD x instructions per wave)        [shader cycles]
// and the table prints the median over all SIMDs (and the mean wave count per SIMD, which must equal N).
// With one wave per SIMD the dependent column is the instruction's issue-to-issue latency.
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/probe/valu_issue_probe.hip -o /tmp/valu_issue_probe && /tmp/valu_issue_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define R8(x) x x x x x x x x
#define R16(x) R8(x) R8(x)

enum Mode : int {
    kAndI, kAndD, kShlI, kShlD, kAlignI, kAlignD, kCmpSelI, kCmpSelD, kBpermI, kBpermD, kMix21, kMix21D,
    kAddI, kMulLoI, kMad24I, kBfeI, kLshlOrI, kMaxI, kFfblI, kFmaI, kFmaD, kShl64I, kReadlaneI, kDsReadI, kDsReadD,
    kAnd3I, kXorD, kCndVccD, kSaluOnly,
    kAndV, kOrV, kSubV, kLshrV, kMinV, kXorS, kAddS, kMov, kBfi, kPerm, kAdd3, kLshlAdd, kOr3, kXad, kCmpOnly, kCndSel, kCndVccI, kDpp, kRfl, kDsU8, kDsB64, kDsB128, kDsW32, kDsAdd, kDsAddRtn, kMixAdd, kMixLds,
    kAndC1, kAndLit, kAddC4, kAddLit, kShlC, kShlV, kShrC, kMul24V, kAndE64, kCmpVccOnly, kCmp1Sel, kCmp2Sel, kCmp4Sel, kCmp4SelE64, kFastSlow, kMbcnt, kAddCo, kDsU16, kDsW8, kDsW64, kModes
};
static const char* kNames[kModes] = {
    "v_and_b32 independent x8", "v_and_b32 dependent", "v_lshlrev_b32 independent x8", "v_lshlrev_b32 dependent",
    "v_alignbyte_b32 independent x8", "v_alignbyte_b32 dependent", "v_cmp_eq_u32_e64 + v_cndmask_b32 independent x4 pairs",
    "v_cmp_eq_u32 (vcc) + v_cndmask_b32 dependent pair", "ds_bpermute_b32 independent x8 (one wait per 8)",
    "ds_bpermute_b32 dependent (wait each)", "2 VALU : 1 SALU mix, independent (per instruction of the 3)",
    "2 VALU : 1 SALU mix, VALU dependent", "v_add_u32 independent x8", "v_mul_lo_u32 independent x8",
    "v_mad_u32_u24 independent x8", "v_bfe_u32 independent x8", "v_lshl_or_b32 independent x8", "v_max_u32 independent x8",
    "v_ffbl_b32 independent x8", "v_fma_f32 independent x8", "v_fma_f32 dependent", "v_lshlrev_b64 independent x4",
    "v_readlane_b32 + s_add (per pair)", "ds_read_b32 independent x8 (one wait per 8)", "ds_read_b32 dependent (wait each)",
    "v_and_or_b32 (VOP3, 3 sources) independent x8", "v_xor_b32 dependent", "v_cndmask_b32 (vcc fixed) dependent",
    "s_add_u32 only, independent x8",
    "v_and_b32 (VGPR operand) independent x8",
    "v_or_b32 (VGPR operand) independent x8",
    "v_sub_u32 (VGPR operand) independent x8",
    "v_lshrrev_b32 (VGPR operand) independent x8",
    "v_min_u32 (VGPR operand) independent x8",
    "v_xor_b32 (SGPR operand) independent x8",
    "v_add_u32 (SGPR operand) independent x8",
    "v_mov_b32 independent x8",
    "v_bfi_b32 independent x8",
    "v_perm_b32 independent x8",
    "v_add3_u32 independent x8",
    "v_lshl_add_u32 independent x8",
    "v_or3_b32 independent x8",
    "v_xad_u32 independent x8",
    "v_cmp_eq_u32_e64 only, x4 SGPR pairs",
    "v_cndmask_b32_e64 (fixed SGPR-pair mask) independent x8",
    "v_cndmask_b32 (vcc, fixed) independent x8",
    "v_mov_b32_dpp row_shr:1 independent x8",
    "v_readfirstlane_b32 + s_add (per pair)",
    "ds_read_u8 independent x8 (one wait per 8)",
    "ds_read_b64 independent x4 (one wait per 4)",
    "ds_read_b128 independent x2 (one wait per 2)",
    "ds_write_b32 independent x8 (one wait per 8)",
    "ds_add_u32 (no return) independent x8 (one wait per 8)",
    "ds_add_rtn_u32 independent x8 (one wait per 8)",
    "2 v_add_u32 : 1 SALU mix, independent (per instruction of the 3)",
    "4 v_and_b32 : 1 ds_read_b32 mix (per instruction of the 5; one wait per 10)",
    "v_and_b32 inline constant (1) independent x8",
    "v_and_b32 literal (0xffff) independent x8",
    "v_add_u32 inline constant (4) independent x8",
    "v_add_u32 literal (0x12345) independent x8",
    "v_lshlrev_b32 inline constant shift (3) independent x8",
    "v_lshlrev_b32 (VGPR shift) independent x8",
    "v_lshrrev_b32 inline constant shift (3) independent x8",
    "v_mul_u32_u24 (VGPR operand) independent x8",
    "v_and_b32_e64 (VOP3 encoding, VGPRs) independent x8",
    "v_cmp_eq_u32 vcc (VOPC e32) only x8",
    "v_cmp vcc + 1 v_cndmask vcc (x2 groups; per instruction)",
    "v_cmp vcc + 2 v_cndmask vcc (x2 groups; per instruction)",
    "v_cmp vcc + 4 v_cndmask vcc (x2 groups; per instruction)",
    "v_cmp_e64 sgpr + 4 v_cndmask_e64 (x2 groups; per instruction)",
    "v_xor_b32 (VGPR) alternating with v_bfe_u32 (per instruction)",
    "v_mbcnt_lo_u32_b32 independent x8",
    "v_add_co_u32 vcc independent x8",
    "ds_read_u16 independent x8 (one wait per 8)",
    "ds_write_b8 independent x8 (one wait per 8)",
    "ds_write_b64 independent x4 (one wait per 4)"};
static const int kInstrPerBlock[kModes] = {8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 12, 12, 8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 2, 8, 8, 8, 12, 10, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 6, 10, 10, 8, 8, 8, 8, 8, 4};

__global__ void __launch_bounds__(256) probe(uint64_t* out, int mode, int iters) {
    extern __shared__ uint32_t lds[];
    uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 11 + 4, r5 = r0 * 13 + 5, r6 = r0 * 17 + 6,
             r7 = r0 * 19 + 7;
    uint32_t m = 0xFFFFFFF7u + (uint32_t)(iters & 1), sh = (uint32_t)(iters & 1), c9 = threadIdx.x ^ 0x55u;
    uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    float f0 = r0, f1 = r1, f2 = r2, f3 = r3, f4 = r4, f5 = r5, f6 = r6, f7 = r7, fm = 1.0f, fa = 0.0f;
    uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3;
    const uint32_t lane4 = (threadIdx.x & 63u) * 4u;
    lds[threadIdx.x] = lane4;  // (ds_read chains: every lane reads its own address back)
    lds[threadIdx.x + 256] = threadIdx.x;
    __syncthreads();
    r0 = mode == kBpermI || mode == kBpermD || mode == kDsReadI || mode == kDsReadD ? lane4 : r0;
    if (mode == kBpermI || mode == kDsReadI) r1 = r2 = r3 = r4 = r5 = r6 = r7 = lane4;
    uint64_t t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
#define LOOP16(ASM) \
    for (int it = 0; it < iters; it++) { R16(ASM;) }
    switch (mode) {
    case kAndI: LOOP16(asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(m))) break;
    case kAndD: LOOP16(asm volatile(R8("v_and_b32 %0, %1, %0\n") : "+v"(r0) : "s"(m))) break;
    case kXorD: LOOP16(asm volatile(R8("v_xor_b32 %0, %1, %0\n") : "+v"(r0) : "v"(c9))) break;
    case kShlI: LOOP16(asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sh))) break;
    case kShlD: LOOP16(asm volatile(R8("v_lshlrev_b32 %0, %1, %0\n") : "+v"(r0) : "s"(sh))) break;
    case kAlignI: LOOP16(asm volatile("v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9))) break;
    case kAlignD: LOOP16(asm volatile(R8("v_alignbyte_b32 %0, %0, %1, 1\n") : "+v"(r0) : "v"(c9))) break;
    case kCmpSelI: {
        uint64_t k0, k1, k2, k3;
        LOOP16(asm volatile("v_cmp_eq_u32_e64 %4, %0, %8\n v_cmp_eq_u32_e64 %5, %1, %8\n v_cmp_eq_u32_e64 %6, %2, %8\n v_cmp_eq_u32_e64 %7, %3, %8\n"
                            "v_cndmask_b32_e64 %0, %0, %8, %4\n v_cndmask_b32_e64 %1, %1, %8, %5\n v_cndmask_b32_e64 %2, %2, %8, %6\n v_cndmask_b32_e64 %3, %3, %8, %7"
                            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3) : "v"(c9)))
        break;
    }
    case kCmpSelD: LOOP16(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r0) : "v"(c9) : "vcc")) break;
    case kCndVccD: asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(r1), "v"(c9) : "vcc"); LOOP16(asm volatile(R8("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(r0) : "v"(c9) : "vcc")) break;
    case kBpermI: LOOP16(asm volatile("ds_bpermute_b32 %0, %0, %0\n ds_bpermute_b32 %1, %1, %1\n ds_bpermute_b32 %2, %2, %2\n ds_bpermute_b32 %3, %3, %3\n ds_bpermute_b32 %4, %4, %4\n ds_bpermute_b32 %5, %5, %5\n ds_bpermute_b32 %6, %6, %6\n ds_bpermute_b32 %7, %7, %7\n s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kBpermD: LOOP16(asm volatile(R8("ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(r0))) break;
    case kDsReadI: LOOP16(asm volatile("ds_read_b32 %0, %0\n ds_read_b32 %1, %1\n ds_read_b32 %2, %2\n ds_read_b32 %3, %3\n ds_read_b32 %4, %4\n ds_read_b32 %5, %5\n ds_read_b32 %6, %6\n ds_read_b32 %7, %7\n s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kDsReadD: LOOP16(asm volatile(R8("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(r0))) break;
    case kMix21: LOOP16(asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n s_add_u32 %9, %9, 1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n s_add_u32 %10, %10, 1\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n s_add_u32 %11, %11, 1\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n s_add_u32 %12, %12, 1" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(m), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "scc")) break;
    case kMix21D: LOOP16(asm volatile("v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n s_add_u32 %2, %2, 1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n s_add_u32 %3, %3, 1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n s_add_u32 %4, %4, 1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n s_add_u32 %5, %5, 1" : "+v"(r0) : "s"(m), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "scc")) break;
    case kSaluOnly: LOOP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc")) break;
#define VOP2_I(OP) LOOP16(asm volatile(OP " %0, %8, %0\n " OP " %1, %8, %1\n " OP " %2, %8, %2\n " OP " %3, %8, %3\n " OP " %4, %8, %4\n " OP " %5, %8, %5\n " OP " %6, %8, %6\n " OP " %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9)))
#define VOP3_I(OP) LOOP16(asm volatile(OP " %0, %0, %8, %8\n " OP " %1, %1, %8, %8\n " OP " %2, %2, %8, %8\n " OP " %3, %3, %8, %8\n " OP " %4, %4, %8, %8\n " OP " %5, %5, %8, %8\n " OP " %6, %6, %8, %8\n " OP " %7, %7, %8, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9)))
    case kAddI: VOP2_I("v_add_u32") break;
    case kMaxI: VOP2_I("v_max_u32") break;
    case kMulLoI: LOOP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9))) break;
    case kMad24I: VOP3_I("v_mad_u32_u24") break;
    case kBfeI: VOP3_I("v_bfe_u32") break;
    case kLshlOrI: VOP3_I("v_lshl_or_b32") break;
    case kAnd3I: VOP3_I("v_and_or_b32") break;
    case kFfblI: LOOP16(asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kFmaI: LOOP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fm), "v"(fa))) break;
    case kFmaD: LOOP16(asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(f0) : "v"(fm), "v"(fa))) break;
    case kShl64I: LOOP16(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "s"(sh))) break;
    case kReadlaneI: LOOP16(asm volatile("v_readlane_b32 %4, %0, 3\n s_add_u32 %8, %8, %4\n v_readlane_b32 %5, %1, 5\n s_add_u32 %8, %8, %5\n v_readlane_b32 %6, %2, 7\n s_add_u32 %8, %8, %6\n v_readlane_b32 %7, %3, 9\n s_add_u32 %8, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "+s"(sh) : : "scc")) break;
#define VOP2_S(OP) LOOP16(asm volatile(OP " %0, %8, %0\n " OP " %1, %8, %1\n " OP " %2, %8, %2\n " OP " %3, %8, %3\n " OP " %4, %8, %4\n " OP " %5, %8, %5\n " OP " %6, %8, %6\n " OP " %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(m)))
#define VOP1_I(OP) LOOP16(asm volatile(OP " %0, %0\n " OP " %1, %1\n " OP " %2, %2\n " OP " %3, %3\n " OP " %4, %4\n " OP " %5, %5\n " OP " %6, %6\n " OP " %7, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)))
#define DS_I(OP) LOOP16(asm volatile(OP " %0, %8\n " OP " %1, %8\n " OP " %2, %8\n " OP " %3, %8\n " OP " %4, %8\n " OP " %5, %8\n " OP " %6, %8\n " OP " %7, %8\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(lane4)))
    case kAndV: VOP2_I("v_and_b32") break;
    case kOrV: VOP2_I("v_or_b32") break;
    case kSubV: VOP2_I("v_sub_u32") break;
    case kLshrV: VOP2_I("v_lshrrev_b32") break;
    case kMinV: VOP2_I("v_min_u32") break;
    case kXorS: VOP2_S("v_xor_b32") break;
    case kAddS: VOP2_S("v_add_u32") break;
    case kMov: VOP1_I("v_mov_b32") break;
    case kBfi: VOP3_I("v_bfi_b32") break;
    case kPerm: VOP3_I("v_perm_b32") break;
    case kAdd3: VOP3_I("v_add3_u32") break;
    case kLshlAdd: VOP3_I("v_lshl_add_u32") break;
    case kOr3: VOP3_I("v_or3_b32") break;
    case kXad: VOP3_I("v_xad_u32") break;
    case kCmpOnly: { uint64_t k0, k1, k2, k3; LOOP16(asm volatile("v_cmp_eq_u32_e64 %4, %0, %8\n v_cmp_eq_u32_e64 %5, %1, %8\n v_cmp_eq_u32_e64 %6, %2, %8\n v_cmp_eq_u32_e64 %7, %3, %8\n v_cmp_eq_u32_e64 %4, %0, %8\n v_cmp_eq_u32_e64 %5, %1, %8\n v_cmp_eq_u32_e64 %6, %2, %8\n v_cmp_eq_u32_e64 %7, %3, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3) : "v"(c9))) } break;
    case kCndSel: { uint64_t km = 0x5555aaaa5555aaaaull + (uint64_t)iters; LOOP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9), "s"(km))) } break;
    case kCndVccI: asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(r1), "v"(c9) : "vcc"); LOOP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9) : "vcc")) break;
    case kDpp: LOOP16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kRfl: LOOP16(asm volatile("v_readfirstlane_b32 %4, %0\n s_add_u32 %8, %8, %4\n v_readfirstlane_b32 %5, %1\n s_add_u32 %8, %8, %5\n v_readfirstlane_b32 %6, %2\n s_add_u32 %8, %8, %6\n v_readfirstlane_b32 %7, %3\n s_add_u32 %8, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "+s"(sh) : : "scc")) break;
    case kDsU8: DS_I("ds_read_u8") break;
    case kDsB64: LOOP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n s_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(lane4 * 2))) break;
    case kDsB128: { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 a, b; LOOP16(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(lane4 * 4))) r1 ^= a.x ^ b.y; } break;
    case kDsW32: LOOP16(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)" : : "v"(lane4 + 2048), "v"(c9) : "memory")) break;
    case kDsAdd: LOOP16(asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:256\n ds_add_u32 %0, %1 offset:512\n ds_add_u32 %0, %1 offset:768\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:256\n ds_add_u32 %0, %1 offset:512\n ds_add_u32 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)" : : "v"(lane4 + 2048), "v"(c9) : "memory")) break;
    case kDsAddRtn: LOOP16(asm volatile("ds_add_rtn_u32 %0, %8, %9\n ds_add_rtn_u32 %1, %8, %9 offset:256\n ds_add_rtn_u32 %2, %8, %9 offset:512\n ds_add_rtn_u32 %3, %8, %9 offset:768\n ds_add_rtn_u32 %4, %8, %9\n ds_add_rtn_u32 %5, %8, %9 offset:256\n ds_add_rtn_u32 %6, %8, %9 offset:512\n ds_add_rtn_u32 %7, %8, %9 offset:768\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(lane4 + 2048), "v"(c9) : "memory")) break;
    case kMixAdd: LOOP16(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n s_add_u32 %9, %9, 1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n s_add_u32 %10, %10, 1\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n s_add_u32 %11, %11, 1\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n s_add_u32 %12, %12, 1" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "scc")) break;
    case kMixLds: LOOP16(asm volatile("ds_read_b32 %4, %6\n v_and_b32 %0, %7, %0\n v_and_b32 %1, %7, %1\n v_and_b32 %2, %7, %2\n v_and_b32 %3, %7, %3\n ds_read_b32 %5, %6\n v_and_b32 %0, %7, %0\n v_and_b32 %1, %7, %1\n v_and_b32 %2, %7, %2\n v_and_b32 %3, %7, %3\n s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(r4), "=&v"(r5) : "v"(lane4), "v"(c9))) break;
    case kAndC1: LOOP16(asm volatile("v_and_b32 %0, 1, %0\n v_and_b32 %1, 1, %1\n v_and_b32 %2, 1, %2\n v_and_b32 %3, 1, %3\n v_and_b32 %4, 1, %4\n v_and_b32 %5, 1, %5\n v_and_b32 %6, 1, %6\n v_and_b32 %7, 1, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kAndLit: LOOP16(asm volatile("v_and_b32 %0, 0xffff, %0\n v_and_b32 %1, 0xffff, %1\n v_and_b32 %2, 0xffff, %2\n v_and_b32 %3, 0xffff, %3\n v_and_b32 %4, 0xffff, %4\n v_and_b32 %5, 0xffff, %5\n v_and_b32 %6, 0xffff, %6\n v_and_b32 %7, 0xffff, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kAddC4: LOOP16(asm volatile("v_add_u32 %0, 4, %0\n v_add_u32 %1, 4, %1\n v_add_u32 %2, 4, %2\n v_add_u32 %3, 4, %3\n v_add_u32 %4, 4, %4\n v_add_u32 %5, 4, %5\n v_add_u32 %6, 4, %6\n v_add_u32 %7, 4, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kAddLit: LOOP16(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kShlC: LOOP16(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kShlV: VOP2_I("v_lshlrev_b32") break;
    case kShrC: LOOP16(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kMul24V: VOP2_I("v_mul_u32_u24") break;
    case kAndE64: LOOP16(asm volatile("v_and_b32_e64 %0, %8, %0\n v_and_b32_e64 %1, %8, %1\n v_and_b32_e64 %2, %8, %2\n v_and_b32_e64 %3, %8, %3\n v_and_b32_e64 %4, %8, %4\n v_and_b32_e64 %5, %8, %5\n v_and_b32_e64 %6, %8, %6\n v_and_b32_e64 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9))) break;
    case kCmpVccOnly: LOOP16(asm volatile("v_cmp_eq_u32 vcc, %0, %8\n v_cmp_eq_u32 vcc, %1, %8\n v_cmp_eq_u32 vcc, %2, %8\n v_cmp_eq_u32 vcc, %3, %8\n v_cmp_eq_u32 vcc, %4, %8\n v_cmp_eq_u32 vcc, %5, %8\n v_cmp_eq_u32 vcc, %6, %8\n v_cmp_eq_u32 vcc, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9) : "vcc")) break;
    case kCmp1Sel: { uint32_t d4 = 0, d5 = 0; LOOP16(asm volatile("v_cmp_eq_u32 vcc, %0, %6\n v_cndmask_b32 %0, %0, %6, vcc\n v_cmp_eq_u32 vcc, %1, %6\n v_cndmask_b32 %0, %0, %6, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(d4), "+v"(d5) : "v"(c9) : "vcc")) r4 ^= d4 ^ d5; } break;
    case kCmp2Sel: { uint32_t d4 = 0, d5 = 0; LOOP16(asm volatile("v_cmp_eq_u32 vcc, %0, %6\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cmp_eq_u32 vcc, %1, %6\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(d4), "+v"(d5) : "v"(c9) : "vcc")) r4 ^= d4 ^ d5; } break;
    case kCmp4Sel: { uint32_t d4 = 0, d5 = 0; LOOP16(asm volatile("v_cmp_eq_u32 vcc, %0, %6\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %6, vcc\n v_cmp_eq_u32 vcc, %1, %6\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %6, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(d4), "+v"(d5) : "v"(c9) : "vcc")) r4 ^= d4 ^ d5; } break;
    case kCmp4SelE64: { uint64_t k0, k1; LOOP16(asm volatile("v_cmp_eq_u32_e64 %4, %0, %6\n v_cndmask_b32_e64 %0, %0, %6, %4\n v_cndmask_b32_e64 %1, %1, %6, %4\n v_cndmask_b32_e64 %2, %2, %6, %4\n v_cndmask_b32_e64 %3, %3, %6, %4\n v_cmp_eq_u32_e64 %5, %1, %6\n v_cndmask_b32_e64 %0, %0, %6, %5\n v_cndmask_b32_e64 %1, %1, %6, %5\n v_cndmask_b32_e64 %2, %2, %6, %5\n v_cndmask_b32_e64 %3, %3, %6, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(k0), "=&s"(k1) : "v"(c9))) } break;
    case kFastSlow: LOOP16(asm volatile("v_xor_b32 %0, %8, %0\n v_bfe_u32 %1, %1, %8, %8\n v_xor_b32 %2, %8, %2\n v_bfe_u32 %3, %3, %8, %8\n v_xor_b32 %4, %8, %4\n v_bfe_u32 %5, %5, %8, %8\n v_xor_b32 %6, %8, %6\n v_bfe_u32 %7, %7, %8, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9))) break;
    case kMbcnt: LOOP16(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_lo_u32_b32 %3, -1, %3\n v_mbcnt_lo_u32_b32 %4, -1, %4\n v_mbcnt_lo_u32_b32 %5, -1, %5\n v_mbcnt_lo_u32_b32 %6, -1, %6\n v_mbcnt_lo_u32_b32 %7, -1, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))) break;
    case kAddCo: LOOP16(asm volatile("v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %8, %2\n v_add_co_u32 %3, vcc, %8, %3\n v_add_co_u32 %4, vcc, %8, %4\n v_add_co_u32 %5, vcc, %8, %5\n v_add_co_u32 %6, vcc, %8, %6\n v_add_co_u32 %7, vcc, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c9) : "vcc")) break;
    case kDsU16: DS_I("ds_read_u16") break;
    case kDsW8: LOOP16(asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %1 offset:256\n ds_write_b8 %0, %1 offset:512\n ds_write_b8 %0, %1 offset:768\n ds_write_b8 %0, %1\n ds_write_b8 %0, %1 offset:256\n ds_write_b8 %0, %1 offset:512\n ds_write_b8 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)" : : "v"(lane4 + 2048), "v"(c9) : "memory")) break;
    case kDsW64: LOOP16(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n s_waitcnt lgkmcnt(0)" : : "v"(lane4 * 2 + 4096), "v"(q0) : "memory")) break;
    default: break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    uint32_t sink = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ s0 ^ s1 ^ s2 ^ s3 ^ sh ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[w * 4 + 0] = t0, out[w * 4 + 1] = t1, out[w * 4 + 2] = ((uint64_t)(xcc & 15) << 32) | hw, out[w * 4 + 3] = sink;
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    printf("# %s, %d CUs, clock %d kHz; %d x 16 blocks per wave and mode; cycles per wave64-instruction and SIMD (median over SIMDs)\n", prop.gcnArchName, cus,
           prop.clockRate, iters);
    const int per_simd[4] = {1, 2, 4, 7};
    uint64_t* d;
    (void)hipMalloc(&d, (size_t)cus * 8 * 4 * 4 * sizeof(uint64_t));
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-78s", "waves per SIMD ->");
    for (int n : per_simd) printf(" %8d", n);
    printf("   (mean waves found per SIMD)\n");
    for (int mode = 0; mode < kModes; mode++) {
        printf("%-78s", kNames[mode]);
        double found[4];
        for (int ni = 0; ni < 4; ni++) {
            const int n = per_simd[ni], grid = cus * n;
            const size_t lds = (160 * 1024 / n) & ~1023u;  // exactly n workgroups fit per CU
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds >= 2048 ? lds - 1024 : 1024, 0, d, mode, iters);
                (void)hipDeviceSynchronize();
            }
            std::vector<uint64_t> h((size_t)grid * 16);
            (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            struct Acc { uint64_t lo = ~0ull, hi = 0; int n = 0; };
            std::map<uint64_t, Acc> simd;
            for (int w = 0; w < grid * 4; w++) {
                const uint64_t id = h[w * 4 + 2];
                const uint32_t hw = (uint32_t)id;
                const uint64_t key = (id >> 32) << 32 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 10 | ((hw >> 8) & 15) << 4 | ((hw >> 4) & 3);
                Acc& a = simd[key];
                a.lo = std::min(a.lo, h[w * 4]), a.hi = std::max(a.hi, h[w * 4 + 1]), a.n++;
            }
            std::vector<double> cyc;
            double nsum = 0;
            const double instr = (double)iters * 16 * kInstrPerBlock[mode];
            for (auto& kv : simd) cyc.push_back((double)(kv.second.hi - kv.second.lo) / (kv.second.n * instr)), nsum += kv.second.n;
            std::sort(cyc.begin(), cyc.end());
            printf(" %8.2f", cyc[cyc.size() / 2]);
            found[ni] = nsum / simd.size();
        }
        printf("   (%.2f %.2f %.2f %.2f)\n", found[0], found[1], found[2], found[3]);
    }
    return 0;
}
