// tamp_common.hpp -- constants and small device helpers shared by the gfx950 kernels.
//
// Format facts restated from the reference (paths relative to its repository root):
//   prefix code for match lengths ......... tamp/_c_src/tamp/compressor.c:33-36
//   RLE / extended-match token layout ..... tamp/_c_src/tamp/compressor.c:257-263,342-415
//   tamp_res status numbering ............. tamp/_c_src/tamp/common.h:145-168
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tamp_amd {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr uint32_t kRing = 16;     // the reference's input ring: look-ahead of one parse step
constexpr uint32_t kPendMax = 256; // >= 241 (longest RLE run) and >= 134 (longest extended match)
constexpr uint32_t kRleMax = 241;  // (14 << 4) + 15 + 2
constexpr uint32_t kRleWindowMax = 8;
constexpr uint32_t kExtExtraMax = 120;  // (14 << 3) + 7 + 1
constexpr int kSymRle = 12, kSymExt = 13, kSymFlush = 14;

enum : int8_t {
    kOk = 0,
    kOutputFull = 1,
    kInputExhausted = 2,
    kError = -1,
    kExcessBits = -2,
    kInvalidConf = -3,
    kOob = -4,
    kBadArgument = -21,  // TAMP_AMD_BAD_ARGUMENT (include/tamp_amd.h)
};

// The decoders count input in BITS in 32-bit registers: one call takes less than 2^29 bytes per stream.  The one-shot
// batch call reports longer streams as kBadArgument; the resumable call offers itself kMaxDecodeCall bytes and reports
// them consumed (the caller offers the rest again, as after any TAMP_INPUT_EXHAUSTED).
constexpr uint32_t kMaxDecodeIn = (1u << 29) - 1;
constexpr uint32_t kMaxDecodeCall = 1u << 28;

// Match-length prefix code: code without the leading 0 flag, length including it.
__device__ __constant__ uint8_t d_code[15] = {0x00, 0x03, 0x08, 0x0b, 0x14, 0x24, 0x26, 0x2b,
                                              0x4b, 0x54, 0x94, 0x95, 0xaa, 0x27, 0xab};
__device__ __constant__ uint8_t d_nbits[15] = {2, 3, 5, 5, 6, 7, 7, 7, 8, 8, 9, 9, 9, 7, 9};

__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) & ~(a - 1); }

__host__ __device__ inline int min_pattern_size(int window, int literal) {
    return 2 + (window > 10 + 2 * (literal - 5));
}

// gfx950 LDS serves misaligned ds_read_b32 / _b128 in hardware (the compiler emits them for align-1 types); a
// misaligned access costs the LDS pipe extra passes, so the hot bucket scan keeps aligned dwords + v_alignbyte.
struct __attribute__((packed, aligned(1))) LdsU32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) LdsU128 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint32_t lds_u32_hw(const uint8_t* base, uint32_t off) {
    return reinterpret_cast<const LdsU32*>(base + off)->v;
}

// 4 bytes at an arbitrary LDS byte offset: two aligned dwords funnel-shifted by the byte phase.
__device__ __forceinline__ uint32_t lds_u32_unaligned(const uint8_t* base, uint32_t off) {
#ifdef TAMP_HWUA_ALL
    return lds_u32_hw(base, off);
#else
    const uint32_t* w = reinterpret_cast<const uint32_t*>(base + (off & ~3u));
    return __builtin_amdgcn_alignbyte(w[1], w[0], off & 3u);
#endif
}

// Inclusive scans over the 64 lanes with DPP row shifts and row broadcasts (six data-parallel-primitive moves instead of
// six ds_bpermute round trips through the LDS pipe with their lane-index arithmetic: __shfl_up compiles to the latter).
// row_shr:n keeps `old` (the identity) where the source lane lies outside the 16-lane row; row_bcast:15 / :31 hand the
// last lane of a row / of the lower half to the rows the row mask selects.
// The DPP controls used below (row_bcast:15 / :31 here, wave_shr:1 in the resolve kernel) exist on the GFX9 family
// (GCN / CDNA, 64-wide wavefronts) only -- which is all this library builds for.  Call sites must be wave-uniform with
// every lane enabled: the row broadcasts read lanes that a partial EXEC mask would leave stale.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "tamp_amd device code is written for gfx9-family wave64 targets (gfx950); its DPP scans have no gfx10+ form"
#endif
template <class Op>
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x, uint32_t identity, Op op) {
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, 0x111, 0xF, 0xF, false));  // row_shr:1
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, 0x112, 0xF, 0xF, false));  // row_shr:2
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, 0x114, 0xF, 0xF, false));  // row_shr:4
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, 0x118, 0xF, 0xF, false));  // row_shr:8
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
    return wave_scan_incl(x, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
    return wave_scan_incl(x, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off);
        v = o > v ? o : v;
    }
    return v;
}

}  // namespace tamp_amd
