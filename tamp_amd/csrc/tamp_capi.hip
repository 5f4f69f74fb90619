// tamp_capi.hip -- the C ABI of include/tamp_amd.h over the gfx950 kernels.
//
// Thin host shim: argument checks, H2D/D2H staging for host buffers, launch geometry, the seeded
// default dictionaries (tamp/_c_src/tamp/common.c:18-52, computed once per device and kept in HBM),
// and the hipEvent timing hook bench.py reads.  All codec work happens in the kernels; there is no
// CPU code path for it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <cstring>

#include "tamp_amd.h"
#include "tamp_compat.h"
#include "tamp_compress_kernel.hpp"
#include "tamp_decompress_kernel.hpp"
#include "tamp_decompress_split_kernel.hpp"
#include "tamp_decompress_long_kernel.hpp"
#include "tamp_decompress_wave_kernel.hpp"
#include "tamp_decompress_resume_kernel.hpp"
#include "tamp_compress_resume_kernel.hpp"

using namespace tamp_amd;

namespace {

constexpr int kMaxDevices = 64;
constexpr size_t kSeedTable = (size_t)1 << 15;

struct DeviceCtx {
    bool ready = false;
    int cu_count = 0;
    size_t lds_per_block = 0;
    uint8_t* seed_dicts = nullptr;  // 3 x 32 KiB: literal<=5, ==6, >=7
    uint32_t* work_counters = nullptr;  // persistent-grid builds: one stream counter per launch, handed out round robin
    std::atomic<uint32_t> next_counter{0};
    static constexpr uint32_t kCounters = 4096;
    // decoder window slabs, one per HIP stream that ever needed one: launches on one stream are ordered and may share
    // a slab, launches on different streams run concurrently and may not
    struct Slab {
        uint8_t* p = nullptr;
        size_t bytes = 0;
        uint32_t* scan = nullptr;  // header pre-pass results (largest window / longest stream / window bytes / largest out_cap)
        uint8_t* split = nullptr;  // split decoder: token records, per-stream meta words, lag lists, fallback flags
        size_t split_bytes = 0;
        // held from the first look at the slab to the last launch of a call: a second host thread using the same stream
        // could otherwise grow (synchronise, free, allocate) the slab between this call's pointer read and its launch
        std::mutex launch_mu;
    };
    std::map<hipStream_t, Slab> slabs;
    // host-memory batch calls (TAMP_AMD_MEM_HOST): kept staging buffers and the library's own streams, so that a
    // call costs no hipMalloc and a large batch runs as overlapping chunks (copy in / kernel / copy out)
    struct HostPipe {
        static constexpr int kDepth = 3;
        std::mutex mu;  // one host-memory batch call per device at a time
        hipStream_t s[kDepth] = {nullptr, nullptr, nullptr};
        struct Grow {
            void* p = nullptr;
            size_t bytes = 0;
            hipError_t need(size_t n) {
                if (n <= bytes) return hipSuccess;
                if (p) (void)hipFree(p);
                p = nullptr, bytes = 0;
                n += n / 4 + 4096;
                hipError_t e = hipMalloc(&p, n);
                if (e == hipSuccess) bytes = n;
                return e;
            }
        } in[kDepth], out[kDepth], meta[kDepth], dict;
        // pinned host staging for output slabs that do not tile their extent (gaps, padding, permuted offsets): the chunk's
        // extent comes back in ONE transfer and the produced bytes are placed by the host
        struct Pinned {
            void* p = nullptr;
            size_t bytes = 0;
            hipError_t need(size_t n) {
                if (n <= bytes) return hipSuccess;
                if (p) (void)hipHostFree(p);
                p = nullptr, bytes = 0;
                n += n / 4 + 4096;
                hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
                if (e == hipSuccess) bytes = n;
                return e;
            }
        } stage[kDepth];
    } pipe;
    // block mode (one long v1 stream over all workgroups): per-block tables of pass 1 / positions of pass 2, and the four
    // table words of the stream read back before the launch
    std::mutex blk_mu;
    std::map<hipStream_t, HostPipe::Grow> blk_scratch;  // (one per HIP stream: two calls in flight on two streams must not share tables)
    // expensive-first ordering: gathered tables, one buffer per HIP stream (launches on a stream are ordered, so the next
    // launch's kernels find the previous one's done with it; a stream-ordered allocation per launch cost 0.7 ms of host time)
    std::mutex long_mu;
    std::map<hipStream_t, HostPipe::Grow> long_scratch;  // one long stream decoded by the whole device: chunk tables, records
    std::map<hipStream_t, HostPipe::Grow> long_maps;     // ... and the groups' tail maps (groups x window x 2 bytes)
    std::map<hipStream_t, HostPipe::Grow> long_spec;     // ... extended format: the list of tokens that can lag, the lag lists
    std::mutex lpt_mu;
    std::map<hipStream_t, HostPipe::Grow> lpt_scratch;
    // one enqueue at a time per HIP stream for the compress launches that keep per-stream scratch (the expensive-first
    // tables): held from the scratch look-up to the last kernel of the call, so that a second host thread on the same stream
    // can neither interleave its helper kernels with this call's nor grow (free) the buffer this call is about to launch
    // with; tamp_amd_trim takes it before it frees.  (Looked up under lpt_mu; map nodes do not move.)
    std::map<hipStream_t, std::mutex> enqueue_mu;
    std::mutex& enqueue_lock(hipStream_t st) {
        std::lock_guard<std::mutex> lock(lpt_mu);
        return enqueue_mu[st];
    }
};

DeviceCtx g_ctx[kMaxDevices];
std::mutex g_mu;

thread_local bool t_timing = false;
thread_local hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
thread_local bool t_ev_valid = false;

thread_local char t_last_error[512] = "";
unsigned long long* g_prof = nullptr;  // -DTAMP_PROF builds: device buffer of per-phase cycle sums

#define HIP_OK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            snprintf(t_last_error, sizeof t_last_error, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,     \
                     hipGetErrorString(_e));                                                              \
            return TAMP_AMD_NO_DEVICE;                                                                    \
        }                                                                                                 \
    } while (0)

void seed_dictionary_host(unsigned char* buf, size_t size, uint8_t literal) {
    // common.c:18-52: xorshift32 from 3758097560, one draw per 8 bytes, nibble selects from a 16-entry table.
    static const char text16[] = " etaoinshrdlcumw";
    static const unsigned char markup16[16] = {' ', 0, '0', 'e', 'i', '>', 't', 'o', '<', 'a', 'n', 's', '\n', 'r', '/', '.'};
    unsigned char table[16];
    for (int k = 0; k < 16; k++)
        table[k] = literal <= 5 ? (unsigned char)(text16[k] & 0x1F)
                                : (literal == 6 ? (unsigned char)(text16[k] & 0x3F) : markup16[k]);
    uint32_t s = 3758097560u, draw = 0;
    for (size_t i = 0; i < size; i++) {
        if ((i & 7) == 0) {
            s ^= s << 13;
            s ^= s >> 17;
            s ^= s << 5;
            draw = s;
        }
        buf[i] = table[draw & 15];
        draw >>= 4;
    }
}

int get_ctx(int device, DeviceCtx** out) {
    if (device < 0 || device >= kMaxDevices) return TAMP_AMD_BAD_ARGUMENT;
    int count = 0;
    HIP_OK(hipGetDeviceCount(&count));
    if (device >= count) {
        snprintf(t_last_error, sizeof t_last_error, "device %d requested, %d visible", device, count);
        return TAMP_AMD_NO_DEVICE;
    }
    HIP_OK(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceCtx& c = g_ctx[device];
    if (!c.ready) {
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, device));
        c.cu_count = prop.multiProcessorCount;
        c.lds_per_block = prop.sharedMemPerBlock;
        std::vector<unsigned char> host(3 * kSeedTable);
        seed_dictionary_host(host.data() + 0 * kSeedTable, kSeedTable, 5);
        seed_dictionary_host(host.data() + 1 * kSeedTable, kSeedTable, 6);
        seed_dictionary_host(host.data() + 2 * kSeedTable, kSeedTable, 8);
        HIP_OK(hipMalloc(&c.seed_dicts, 3 * kSeedTable));
        HIP_OK(hipMemcpy(c.seed_dicts, host.data(), 3 * kSeedTable, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&c.work_counters, DeviceCtx::kCounters * sizeof(uint32_t)));
        c.ready = true;
    }
    *out = &c;
    return TAMP_OK;
}

struct DevBuf {  // RAII device allocation for host-memory calls
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

thread_local bool t_timing_outer = false;  // a caller's event pair spans several inner launches (up to sixteen long streams)
void timing_begin(hipStream_t st) {
    if (t_timing_outer) return;
    t_ev_valid = false;
    if (!t_timing) return;
    if (!t_ev0) {
        (void)hipEventCreate(&t_ev0);
        (void)hipEventCreate(&t_ev1);
    }
    (void)hipEventRecord(t_ev0, st);
}
void timing_end(hipStream_t st) {
    if (!t_timing) return;
    (void)hipEventRecord(t_ev1, st);
    t_ev_valid = true;
}

struct SegmentSpec {  // streaming Compressor over the engine: how this piece of the stream opens and closes
    uint8_t nlead;
    uint16_t lead;
    uint8_t flags;  // kSegResume | kSegSave | kSegFlushToken
};

uint32_t pick_block(uint32_t W, uint32_t max_in_len, bool packed, bool lazy, bool runlist = false, uint32_t hb = kHashBits) {
    // Positions matched per epoch (a multiple of 64: the walk chases 64 positions per register; of 256 when it can be:
    // the index is scattered in tiles of 256 positions, two barriers each, and a last tile that is mostly empty costs
    // as much as a full one).  The whole stream when it is short.  For longer ones what counts is how many workgroups a
    // CU holds -- the kernel is bound by instruction issue and a third of a real-text stream's time is the one-wavefront
    // walk -- so: the LARGEST block that still allows as many workgroups per CU as a 1,024-position block does (the
    // registers allow TAMP_WG_PER_CU = 8 for the run-aware builds since round 6, 6 for the lean and 5 for the lazy ones).  At
    // W = 1024 that is 1,024 positions at eight per CU (19.2 KB with 1,024 buckets; round 4-5: seven at 21.4 KB; rounds 1-3: 1,536
    // positions at six, 26.3 KB): four epochs instead of three for a 4 KiB stream, and faster on every input measured
    // (profiles/ab/r4_seven_workgroups_per_cu.log, profiles/ab/r6_experiments.log).
    uint32_t blk = max_in_len ? align_up(max_in_len, 64) : 2048;
    if (blk > 2048) blk = 2048;
    if (blk > 1024) {
        // (LDS is handed out in coarse granules: 26,960 B per workgroup measured as five per CU, 25,424 B as six)
        const uint32_t lds_cu = 160u * 1024u, granule = 2048u;
        const uint32_t reg_cap = lazy ? (uint32_t)TAMP_LAZY_PER_CU : (runlist ? (uint32_t)TAMP_WG_PER_CU : (uint32_t)TAMP_LEAN_PER_CU);
        auto per_cu = [&](uint32_t b) {
            const uint32_t v = lds_cu / align_up(CompressLds(W, b, packed, lazy, runlist, hb).total, granule);
            return v < reg_cap ? v : reg_cap;
        };
        const uint32_t want = per_cu(1024);
        uint32_t best = 1024;
        for (uint32_t b = 1280; b <= 2048; b += 256)
            if (b <= blk && per_cu(b) == want) best = b;
        // (the stream's own length: one epoch instead of two for streams a little over 1 KiB, when that costs no workgroup)
        if (blk > best && blk < 2048 && per_cu(blk) == want) best = blk;
        blk = best;
    }
    if (const char* e = getenv("TAMP_AMD_BLK")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 64 && v <= 2048) blk = align_up(v, 64); }
    if (blk < 64) blk = 64;
    {
        // The cursor region of LDS also serves as the sorted query list (blk x u16) and, in the run-aware builds, as the walk's
        // explicit pieces + the step table (kSlowCap x 8 + blk bytes): a tuning override must not outgrow it.  (Checked here and
        // not by sizing the region from the block: that arithmetic inside the kernel cost the 64-VGPR builds their last register.)
        const uint32_t cur = (hb < kHashBits ? (1u << hb) : kHashBuckets) * 2;
        while (blk > 64 && (blk * 2 > cur || (runlist && kSlowCap * 8 + blk > cur))) blk -= 64;
    }
    while (W + blk + 16 > 65536) blk >>= 1;  // 16-bit buffer positions
    return blk;
}

// Expensive streams first (round 5).  One stream = one workgroup, so a batch cannot finish before its slowest stream does, and
// a batch of only a few rounds of the persistent grid -- 3,052 streams per GPU when BASELINE configs[2] runs on eight -- waits
// for whichever slow stream happened to start last.  What makes a stream slow are its lags and searches, and a
// cheap proxy ranks them well: the number of ALIGNED DWORDS OF FOUR EQUAL BYTES (of the stand-in's 2,304 chunks the slowest
// ones rank 0-7 of 768 by it on prose, 1-57 on Python sources).  Three tiny kernels around the compress launch: score per
// stream, one-workgroup counting sort (descending), the batch's tables gathered in that order -- the compress kernel reads
// row i of the gathered tables, so its claims ARE the order -- and sizes / statuses scattered back afterwards.
__global__ void __launch_bounds__(256) tamp_stream_score_kernel(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                                                                 uint32_t n_streams, uint32_t* score) {
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_streams) return;
    const uint8_t* p = in + in_off[s];
    const uint32_t n = in_len[s];
    const uint32_t head = (uint32_t)((4 - (reinterpret_cast<uintptr_t>(p) & 3)) & 3);
    uint32_t cnt = 0;
    if (n > head + 4) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p + head);
        const uint32_t nw = (n - head) >> 2;
        for (uint32_t k = lane; k < nw; k += 64) {
            const uint32_t d = w[k];
            cnt += d == (d & 0xFFu) * 0x01010101u;
        }
    }
    cnt = wave_scan_add(cnt);  // (inclusive scan: the last lane holds the sum)
    if (lane == 63) score[s] = cnt;
}
__global__ void __launch_bounds__(1024) tamp_stream_order_kernel(const uint32_t* score, uint32_t n_streams, uint32_t* order) {
    // counting sort by min(score, 1023), descending, one workgroup.  Lanes of a wavefront that hold the same bin go to the LDS
    // counter together (one atomic per distinct bin and wavefront: a batch whose streams all score alike -- synthetic text:
    // zero everywhere -- would otherwise queue 65,536 atomics on one word, 0.11 ms)
    __shared__ uint32_t bins[1024];
    __shared__ uint32_t wsum[16];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    auto grouped_add = [&](uint32_t bin, bool live) -> uint32_t {  // -> this lane's slot in its bin
        uint32_t slot = 0;
        uint64_t todo = __ballot(live);
        while (todo) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bin, (int)leader);
            const uint64_t same = __ballot(live && bin == b) & todo;
            // (real text: the 64 streams of a wavefront score 30-60 different values, and a round of this loop per value
            // is a dependent LDS atomic each -- 0.18-0.34 ms per 32,768 streams.  Small groups go to the counters one lane
            // each, which the LDS serves in parallel: 0.0x ms)
            if (__builtin_popcountll(same) < 8) break;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&bins[b], (uint32_t)__builtin_popcountll(same));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
            if (live && bin == b) slot = base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1));
            todo &= ~same;
        }
        if ((todo >> lane) & 1ull) slot = atomicAdd(&bins[bin], 1u);
        return slot;
    };
    const uint32_t rounds = (n_streams + 1023) / 1024;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t s = r * 1024 + threadIdx.x;
        const bool live = s < n_streams;
        (void)grouped_add(live ? 1023u - min(score[s], 1023u) : 0u, live);
    }
    __syncthreads();
    const uint32_t v = bins[threadIdx.x];  // exclusive scan of the 1,024 bins: 16 wavefronts
    const uint32_t incl = wave_scan_add(v);
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wsum[w];
    bins[threadIdx.x] = base + incl - v;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t s = r * 1024 + threadIdx.x;
        const bool live = s < n_streams;
        const uint32_t slot = grouped_add(live ? 1023u - min(score[s], 1023u) : 0u, live);
        if (live) order[slot] = s;
    }
}
__global__ void tamp_gather_rows_kernel(const uint32_t* order, uint32_t n, const uint64_t* in_off, const uint32_t* in_len,
                                        const uint64_t* out_off, const uint32_t* out_cap, uint64_t* g_in_off, uint32_t* g_in_len,
                                        uint64_t* g_out_off, uint32_t* g_out_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = order[i];
    g_in_off[i] = in_off[s], g_in_len[i] = in_len[s], g_out_off[i] = out_off[s], g_out_cap[i] = out_cap[s];
}
__global__ void tamp_scatter_results_kernel(const uint32_t* order, uint32_t n, const uint32_t* g_out_len, const int8_t* g_status,
                                            uint32_t* out_len, int8_t* status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = order[i];
    out_len[s] = g_out_len[i], status[s] = g_status[i];
}

// Block mode, pass 2: entry offset and bit position of every block from the tables of pass 1 -- a chain of one dependent
// table look-up per block.  Two levels keep it short: chunks of 512 blocks.  tamp_block_scan_chunks: per chunk, for each of
// the 15 entry offsets (a lane each) where the chain leaves the chunk and the bits it takes.  tamp_block_scan_kernel: per
// chunk, the chain over the CHUNK tables in front of it (a few hundred steps at most for 4 GiB), then the chunk's own 512
// blocks from its true entry.  (One workgroup walking all 97,657 blocks of a 100 MB stream took 4 ms of the call's 11.)
constexpr uint32_t kScanChunk = 512;
__global__ void __launch_bounds__(256) tamp_block_scan_chunks(const uint32_t* table, uint32_t* chunk_table /* n_chunks x 16 x 2 */,
                                                              uint32_t n_blocks) {
    __shared__ uint32_t t[kScanChunk * 16];
    const uint32_t b0 = blockIdx.x * kScanChunk;
    const uint32_t cnt = n_blocks - b0 < kScanChunk ? n_blocks - b0 : kScanChunk;
    for (uint32_t i = threadIdx.x; i < cnt * 16; i += blockDim.x) t[i] = table[(size_t)b0 * 16 + i];
    __syncthreads();
    if (threadIdx.x < 16) {
        uint32_t entry = threadIdx.x;
        unsigned long long bits = 0;
        if (threadIdx.x < 15)
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t v = t[i * 16 + entry];
                entry = v & 15u;
                bits += v >> 4;
            }
        // (exit offset | bits << 4 does not fit 32 bits for 512 blocks of 9 Kbit: two words)
        chunk_table[((size_t)blockIdx.x * 16 + threadIdx.x) * 2] = entry;
        chunk_table[((size_t)blockIdx.x * 16 + threadIdx.x) * 2 + 1] = (uint32_t)bits;
    }
}
__global__ void __launch_bounds__(256) tamp_block_scan_kernel(const uint32_t* table, const uint32_t* chunk_table, unsigned long long* info,
                                                              uint32_t n_blocks, uint32_t lead_bits) {
    __shared__ uint32_t t[kScanChunk * 16];
    __shared__ unsigned long long res[kScanChunk];
    const uint32_t b0 = blockIdx.x * kScanChunk;
    const uint32_t cnt = n_blocks - b0 < kScanChunk ? n_blocks - b0 : kScanChunk;
    for (uint32_t i = threadIdx.x; i < cnt * 16; i += blockDim.x) t[i] = table[(size_t)b0 * 16 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bp = lead_bits;
        uint32_t entry = 0;
        for (uint32_t c = 0; c < blockIdx.x; c++) {  // the chunks in front of this one
            const uint32_t* e = chunk_table + ((size_t)c * 16 + entry) * 2;
            bp += e[1];
            entry = e[0];
        }
        for (uint32_t i = 0; i < cnt; i++) {
            res[i] = (bp << 4) | entry;
            const uint32_t v = t[i * 16 + entry];
            entry = v & 15u;
            bp += v >> 4;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) info[b0 + i] = res[i];
}

// Block mode (tamp_compress_kernel<.., BLOCKM>): ONE long stream of the v1 format, literal 8, default parse, fresh window.
// -> TAMP_OK when the stream was taken this way, 1 when the call does not qualify (the caller goes on with the batch kernel).
int launch_compress_blocks(DeviceCtx* ctx, CompressArgs a0, const TampAmdConf* conf, uint32_t max_in_len, hipStream_t st, size_t n_streams) {
    const uint32_t W = 1u << conf->window;
    uint32_t min_len = 256u << 10;
    if (const char* e = getenv("TAMP_AMD_BLOCK_MIN")) min_len = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 0xFFFFFFFFu;  // (tuning / tests; 0 = off)
    if (conf->extended || conf->lazy_matching || conf->literal != 8 || conf->window > 14 || a0.state || a0.seg_flags || max_in_len < min_len ||
        n_streams == 0 || n_streams > 64)
        return 1;
    std::lock_guard<std::mutex> lock(ctx->blk_mu);
    // the streams' table rows: lengths, capacities (the launch geometry and the zero fill depend on them); a handful of LONG
    // streams is taken one after the other, each over all workgroups -- any shorter one among them and the batch kernel takes all
    uint64_t in_off[64], out_off[64];
    uint32_t in_len[64], out_cap[64];
    HIP_OK(hipMemcpyAsync(in_off, a0.in_off, 8 * n_streams, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(out_off, a0.out_off, 8 * n_streams, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(in_len, a0.in_len, 4 * n_streams, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(out_cap, a0.out_cap, 4 * n_streams, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    uint32_t n_max = 0;
    for (size_t i = 0; i < n_streams; i++) {
        if (in_len[i] < min_len) return 1;
        n_max = std::max(n_max, in_len[i]);
    }
    const bool runs_build = !getenv("TAMP_AMD_BLOCK_LEAN");  // (the run-aware build, as for every long stream; tuning: the lean one)
    const uint32_t hb = runs_build && conf->window == 10 ? kHb1024 : kHashBits;
    a0.blk = pick_block(W, 0, true, false, runs_build, hb);
    if (a0.blk > 1024) a0.blk = 1024;  // (more, smaller blocks: the unit of parallelism here)
    const CompressLds L(W, a0.blk, true, false, runs_build, hb);
    if (L.total > ctx->lds_per_block) return 1;
    auto kernel = !runs_build ? tamp_compress_kernel<true, false, false, 0, kHashBits, true, true>
                  : conf->window == 10 ? tamp_compress_kernel<true, false, true, 1024, kHb1024, true, true>
                                       : tamp_compress_kernel<true, false, true, 0, kHashBits, true, true>;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
    int per_cu = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), 256, L.total));
    if (per_cu < 1) per_cu = 1;
    // scratch for the longest of them
    const uint32_t nb_max = (n_max + a0.blk - 1) / a0.blk, nc_max = (nb_max + kScanChunk - 1) / kScanChunk;
    const size_t table_bytes = ((size_t)nb_max * 16 * 4 + 255) & ~(size_t)255, info_bytes = ((size_t)nb_max * 8 + 255) & ~(size_t)255;
    const size_t chunk_bytes = ((size_t)nc_max * 16 * 8 + 255) & ~(size_t)255;
    // (the match results of pass 1, 3 bytes per input byte, when that stays under 1.5 GiB: pass 3 then does not match again)
    const size_t len_bytes = ((size_t)n_max + 1024 + 255) & ~(size_t)255;
    const bool keep_tables = (size_t)n_max * 3 <= ((size_t)3 << 29) && !getenv("TAMP_AMD_BLOCK_REMATCH");
    DeviceCtx::HostPipe::Grow& scratch = ctx->blk_scratch[st];
    HIP_OK(scratch.need(table_bytes + info_bytes + chunk_bytes + 256 + (keep_tables ? 3 * len_bytes : 0)));
    uint8_t* const base = static_cast<uint8_t*>(scratch.p);
    uint32_t* const chunk_table = reinterpret_cast<uint32_t*>(base + table_bytes + info_bytes);
    uint8_t* const tables = base + table_bytes + info_bytes + chunk_bytes;
    timing_begin(st);
    for (size_t i = 0; i < n_streams; i++) {
        CompressArgs a = a0;
        a.in_off += i, a.in_len += i, a.out_off += i, a.out_cap += i, a.out_len += i, a.status += i;  // (the kernel reads row 0)
        const uint32_t n = in_len[i];
        const uint32_t n_blocks = (n + a.blk - 1) / a.blk;
        const uint32_t n_chunks = (n_blocks + kScanChunk - 1) / kScanChunk;
        a.blk_table = reinterpret_cast<uint32_t*>(base);
        a.blk_info = reinterpret_cast<unsigned long long*>(base + table_bytes);
        a.blk_len = keep_tables ? tables : nullptr;
        a.blk_idx = keep_tables ? reinterpret_cast<uint16_t*>(tables + len_bytes) : nullptr;
        a.n_blocks = n_blocks, a.n_streams = n_blocks, a.first_stream = 0, a.claim = 1, a.cut_run = 0;
        const uint32_t g = (uint32_t)std::min<size_t>((size_t)per_cu * (size_t)ctx->cu_count, n_blocks);
        // every byte the emitters may OR into: header + 9 bits per input byte at most (all literals), capped by the caller's room
        const uint64_t bound = (uint64_t)a.nlead + ((uint64_t)n * 9 + 7) / 8 + 8;
        HIP_OK(hipMemsetAsync(a.out + out_off[i], 0, (size_t)std::min<uint64_t>(bound, out_cap[i]), st));
        for (uint32_t pass = 1; pass <= 3; pass++) {
            if (pass == 2) {
                hipLaunchKernelGGL(tamp_block_scan_chunks, dim3(n_chunks), dim3(256), 0, st, a.blk_table, chunk_table, n_blocks);
                hipLaunchKernelGGL(tamp_block_scan_kernel, dim3(n_chunks), dim3(256), 0, st, a.blk_table, chunk_table, a.blk_info, n_blocks, 8u * a.nlead);
                continue;
            }
            const uint32_t slot = ctx->next_counter.fetch_add(1) % DeviceCtx::kCounters;
            a.work_counter = ctx->work_counters + slot;
            a.block_pass = pass;
            HIP_OK(hipMemsetAsync(a.work_counter, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL(kernel, dim3(g), dim3(256), L.total, st, a);
        }
    }
    timing_end(st);
    HIP_OK(hipGetLastError());
    return TAMP_OK;
}

int launch_compress(DeviceCtx* ctx, const TampAmdConf* conf, const uint8_t* d_dict, const uint8_t* d_in,
                    const uint64_t* d_in_off, const uint32_t* d_in_len, uint8_t* d_out, const uint64_t* d_out_off,
                    const uint32_t* d_out_cap, uint32_t* d_out_len, int8_t* d_status, size_t n_streams,
                    uint32_t max_in_len, hipStream_t st, const SegmentSpec* seg = nullptr, uint8_t* d_state = nullptr) {
    if (n_streams == 0) return TAMP_OK;
    CompressArgs a;
    a.in = d_in, a.in_off = d_in_off, a.in_len = d_in_len;
    a.out = d_out, a.out_off = d_out_off, a.out_cap = d_out_cap, a.out_len = d_out_len, a.status = d_status;
    a.wbits = conf->window, a.lbits = conf->literal, a.extended = conf->extended != 0;
    a.dict_reset = conf->dictionary_reset != 0;
    a.lazy = conf->lazy_matching != 0;
    // header byte, compressor.c:236-241 (+ a zero second byte when dictionary_reset is set)
    const uint8_t header = (uint8_t)(((conf->window - 8) << 5) | ((conf->literal - 5) << 3) |
                                     ((conf->use_custom_dictionary != 0) << 2) | ((conf->extended != 0) << 1) |
                                     (conf->dictionary_reset != 0));
    a.nlead = conf->dictionary_reset ? 2 : 1;
    a.lead = (uint16_t)(header << 8);
    a.seg_flags = 0;
    a.state = nullptr;
    if (seg) {
        a.nlead = seg->nlead, a.lead = seg->lead, a.seg_flags = seg->flags, a.state = d_state;
    }
    if (conf->use_custom_dictionary) {
        a.dict = d_dict;
    } else {
        // compressor.c:224-225: non-extended streams always use the literal-8 table
        const int lit = conf->extended ? conf->literal : 8;
        a.dict = ctx->seed_dicts + (lit <= 5 ? 0 : (lit == 6 ? 1 : 2)) * kSeedTable;
    }
    a.n_streams = (uint32_t)n_streams;
    a.prof = g_prof;
    a.work_counter = nullptr;
    a.claim = 1;
    // epoch cut at long runs (extended format only: the v1 format has no RLE token), tamp_compress_kernel.hpp
    a.cut_run = conf->extended ? 3u : 0u;  // (doubles per stream whenever a cut turns out to be superfluous)
    if (const char* e = getenv("TAMP_AMD_CUT_RUN")) { const int v = atoi(e); a.cut_run = (conf->extended && v >= 2 && v <= 64) ? (uint32_t)v : 0u; }
    a.dbg = getenv("TAMP_AMD_DBG") ? (uint32_t)atoi(getenv("TAMP_AMD_DBG")) : 0;
    a.blk_table = nullptr, a.blk_info = nullptr, a.block_pass = 0, a.n_blocks = 0, a.blk_len = nullptr, a.blk_idx = nullptr;
    if (n_streams <= 64 && !seg) {  // a handful of LONG v1 streams: each one's blocks over all workgroups (tamp_compress_kernel<.., BLOCKM>)
        const int rc = launch_compress_blocks(ctx, a, conf, max_in_len, st, n_streams);
        if (rc != 1) return rc;
    }
    const uint32_t W = 1u << conf->window;
    const bool packed = conf->window <= 14;  // u32 index entries; 2^15 windows fall back to u16 positions
    // run-list build: long runs of one byte leave the bigram index; default parse only
    // AUTO goes by stream length alone, for host and device batches alike (no look at the data): the run-aware build for
    // streams of 1 KiB and more -- it settles short RLE runs and most extended matches in the match phase and is the faster
    // one on every kind of text measured, runs or not (config 2: 6.89 against 7.29 ms) -- the lean build for short
    // messages (256-byte telemetry: 2.2 against 2.5 ms), where its per-epoch run search does not pay
    // Round 3: six builds instead of nine.  Streams of 1 KiB and more (256-thread workgroups) always take the run-aware
    // build -- it was the faster one on every kind of text in both formats, so the lean 256-thread builds only served the
    // PLAIN hint; the hint now matters for short messages alone, where the lean one-wavefront build is ahead.  The 2^15
    // window (u16 index entries) has the lean and the lazy build only.
    const bool long_streams = max_in_len == 0 || align_up(max_in_len, 64) >= 1024;  // (= 256-thread workgroups, pick_block)
    bool runlist = packed && !a.lazy && (long_streams || conf->input_hint == TAMP_AMD_HINT_RUNS);
    if (const char* e = getenv("TAMP_AMD_RUNS")) { if (!long_streams) runlist = packed && !a.lazy && atoi(e) != 0; }  // tuning / tests
    const uint32_t hb = runlist && conf->window == 10 ? kHb1024 : kHashBits;
    a.blk = pick_block(W, max_in_len, packed, a.lazy != 0, runlist, hb);
    if (runlist && CompressLds(W, a.blk, packed, false, true, hb).total > ctx->lds_per_block) {
        snprintf(t_last_error, sizeof t_last_error, "LDS %u B > %zu B per block", CompressLds(W, a.blk, packed, false, true, hb).total, ctx->lds_per_block);
        return TAMP_AMD_BAD_ARGUMENT;  // (cannot happen for windows up to 2^14: 105 KB at most)
    }
    const CompressLds L(W, a.blk, packed, a.lazy != 0, runlist, hb);
    if (L.total > ctx->lds_per_block) {
        snprintf(t_last_error, sizeof t_last_error, "LDS %u B > %zu B per block", L.total, ctx->lds_per_block);
        return TAMP_AMD_BAD_ARGUMENT;
    }
    const uint32_t threads = (a.blk >= 1024 || (long_streams && a.blk >= 512 && getenv("TAMP_AMD_BLK"))) ? 256 : 64;  // (tuning: smaller blocks for long streams)
    const uint32_t grid = (uint32_t)(n_streams < (1u << 20) ? n_streams : (1u << 20));
    // the six builds: lazy (u32 / u16 entries), run-aware (generic window / 2^10 with the scan constants as immediates),
    // lean one-wavefront build for short messages (512 buckets: a quarter of the cursors to zero and scan per message),
    // lean u16 build for the 2^15 window.
    // All but the short-message build run as a PERSISTENT GRID (LOOP in the kernel): as many workgroups as the device holds
    // at once, each taking streams from a counter until none is left.  Workgroup i of a grid runs on XCD i % 8 whatever the
    // other XCDs are doing, so with one workgroup per stream an eighth of the batch is pinned to each XCD before anything
    // of the streams' costs is known -- and real text is heavy-tailed (and the frozen corpora, 768 chunks long, handed every
    // XCD the same 96 chunks over and over: 17 % lost, profiles/ab/r3_persistent_grid.log).  Short messages keep one
    // workgroup per stream: two million fetches from one counter cost more than the balance is worth (20 instead of
    // 41 GB/s when tried), and their costs are even.
    const bool short_build = packed && !a.lazy && !runlist;
    auto kernel = a.lazy ? (packed ? tamp_compress_kernel<true, true, false, 0, kHashBits, true> : tamp_compress_kernel<false, true, false, 0, kHashBits, true>)
                  : !packed ? tamp_compress_kernel<false, false, false, 0, kHashBits, true>
                  : runlist ? (conf->window == 10 ? tamp_compress_kernel<true, false, true, 1024, kHb1024, true> : tamp_compress_kernel<true, false, true, 0, kHashBits, true>)
                            : tamp_compress_kernel<true, false, false, 0, 9>;
    if (short_build && threads != 64) {  // (short messages only: long streams are run-aware above)
        snprintf(t_last_error, sizeof t_last_error, "no lean build for %u-thread workgroups", threads);
        return TAMP_AMD_BAD_ARGUMENT;
    }
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)L.total));
    if (!short_build) {
        static std::mutex occ_mu;
        static std::map<std::pair<const void*, uint64_t>, int> occ;  // (the occupancy query costs ~10 us: once per shape)
        int per_cu = 0;
        {
            std::lock_guard<std::mutex> lock(occ_mu);
            const auto key = std::make_pair(reinterpret_cast<const void*>(kernel), (uint64_t)L.total << 16 | threads);
            auto it = occ.find(key);
            if (it == occ.end()) {
                HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), (int)threads, L.total));
                if (per_cu < 1) per_cu = 1;
                occ.emplace(key, per_cu);
            } else {
                per_cu = it->second;
            }
        }
        if (const char* e = getenv("TAMP_AMD_GRID_PER_CU")) {  // (tuning)
            if (atoi(e) > 0) per_cu = atoi(e);
            else fprintf(stderr, "tamp_amd: %d workgroups of %u threads, %u B LDS per CU\n", per_cu, threads, L.total);
        }
        a.first_stream = 0;
        // streams per fetch from the counter: one for 256-thread workgroups (streams of 1 KiB and more), sixteen for
        // one-wavefront ones (lazy / 2^15-window / hinted short messages)
        a.claim = threads == 256 ? 1u : 16u;
        const size_t claims = (n_streams + a.claim - 1) / a.claim;
        size_t g = std::min<size_t>((size_t)per_cu * (size_t)ctx->cu_count, claims);
        // TAMP_AMD_STATIC_GRID=1 (tuning): one workgroup per claim instead -- each takes its claim when it starts and finds
        // the counter exhausted afterwards
        if (getenv("TAMP_AMD_STATIC_GRID")) g = claims;
        // (every workgroup fetches one claim beyond the last: the counter ends at (claims + g) * claim)
        if ((claims + g) * a.claim > 0xFFFFFFFFull || g > 0x7FFFFFFFull) {
            snprintf(t_last_error, sizeof t_last_error, "%zu streams: beyond the 32-bit work counter", n_streams);
            return TAMP_AMD_BAD_ARGUMENT;
        }
        // (every argument check lies in front of the event pair: a refused call leaves no half-recorded timing)
        std::lock_guard<std::mutex> enqueue(ctx->enqueue_lock(st));  // (per HIP stream; see DeviceCtx::enqueue_mu)
        timing_begin(st);
        // expensive streams first, for batches of more than one and at most ~18 rounds of the grid (beyond, the tail is short
        // next to the batch; TAMP_AMD_LPT=0 / =1 force it off / on)
        bool lpt = threads == 256 && !seg && n_streams > g && n_streams <= 32768;
        if (const char* e = getenv("TAMP_AMD_LPT")) lpt = atoi(e) != 0 && threads == 256 && !seg && n_streams > 1 && n_streams <= (1u << 20);
        uint8_t* lpt_mem = nullptr;
        uint32_t* lpt_order = nullptr;
        uint32_t* lpt_out_len = nullptr;
        int8_t* lpt_status = nullptr;
        if (lpt) {
            const size_t n = n_streams;
            const size_t bytes = n * (4 + 4 + 8 + 4 + 8 + 4 + 4 + 1) + 256;
            {
                std::lock_guard<std::mutex> lock(ctx->lpt_mu);
                DeviceCtx::HostPipe::Grow& gbuf = ctx->lpt_scratch[st];
                if (gbuf.need(bytes) == hipSuccess) lpt_mem = static_cast<uint8_t*>(gbuf.p);
            }
            if (!lpt_mem) {
                (void)hipGetLastError();
                lpt = false;
            } else {
                uint64_t* g_in_off = reinterpret_cast<uint64_t*>(lpt_mem);
                uint64_t* g_out_off = g_in_off + n;
                uint32_t* score = reinterpret_cast<uint32_t*>(g_out_off + n);
                lpt_order = score + n;
                uint32_t* g_in_len = lpt_order + n;
                uint32_t* g_out_cap = g_in_len + n;
                lpt_out_len = g_out_cap + n;
                lpt_status = reinterpret_cast<int8_t*>(lpt_out_len + n);
                hipLaunchKernelGGL(tamp_stream_score_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, a.in, a.in_off, a.in_len, (uint32_t)n, score);
                hipLaunchKernelGGL(tamp_stream_order_kernel, dim3(1), dim3(1024), 0, st, score, (uint32_t)n, lpt_order);
                hipLaunchKernelGGL(tamp_gather_rows_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lpt_order, (uint32_t)n, a.in_off,
                                   a.in_len, a.out_off, a.out_cap, g_in_off, g_in_len, g_out_off, g_out_cap);
                a.in_off = g_in_off, a.in_len = g_in_len, a.out_off = g_out_off, a.out_cap = g_out_cap;
                a.out_len = lpt_out_len, a.status = lpt_status;
            }
        }
        const uint32_t slot = ctx->next_counter.fetch_add(1) % DeviceCtx::kCounters;
        a.work_counter = ctx->work_counters + slot;
        HIP_OK(hipMemsetAsync(a.work_counter, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(kernel, dim3((uint32_t)g), dim3(threads), L.total, st, a);
        if (lpt) {
            hipLaunchKernelGGL(tamp_scatter_results_kernel, dim3((uint32_t)((n_streams + 255) / 256)), dim3(256), 0, st, lpt_order,
                               (uint32_t)n_streams, lpt_out_len, lpt_status, d_out_len, d_status);
        }
    } else {
        timing_begin(st);
        const size_t launch_step = grid;
        for (size_t first = 0; first < n_streams; first += launch_step) {  // one stream per workgroup
            a.first_stream = (uint32_t)first;
            const uint32_t g = (uint32_t)std::min<size_t>(grid, n_streams - first);
            hipLaunchKernelGGL(kernel, dim3(g), dim3(threads), L.total, st, a);
        }
    }
    timing_end(st);
    HIP_OK(hipGetLastError());
    return TAMP_OK;
}

// Largest window (bits) any stream header of the batch asks for, among those the caller's limit admits.  LDS rows of the
// decoders are sized from it instead of from the limit: a caller that passes the API default (15) for 1 KiB-window
// streams would otherwise run at a fraction of the occupancy.
__global__ void tamp_header_scan_kernel(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                                        const uint32_t* out_cap, uint32_t n, uint32_t limit, uint32_t* result) {
    uint32_t m = 0, longest = 0, wsum = 0;  // wsum: window bytes / 256, summed over the streams within the limit
    uint32_t maxcap = 0;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const uint32_t len = in_len[s];
        maxcap = out_cap[s] > maxcap ? out_cap[s] : maxcap;
        if (len == 0) continue;
        longest = len > longest ? len : longest;
        const uint32_t w = 8u + (in[in_off[s]] >> 5);  // header byte, decompressor.c:276-297
        if (w <= limit) {
            m = w > m ? w : m;
            wsum += 1u << (w - 8);
        }
    }
    m = wave_max_u32(m);
    longest = wave_max_u32(longest);
    maxcap = wave_max_u32(maxcap);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wsum += (uint32_t)__shfl_xor((int)wsum, off);
    // one set of atomics per WORKGROUP: with one per wavefront, 16,384 wavefronts of a million-stream batch queued 65,536
    // atomics on four addresses -- 0.39 ms for a pre-pass that reads 13 MB (round 4: block reduction through LDS)
    __shared__ uint32_t red[4][4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0) red[wave][0] = m, red[wave][1] = longest, red[wave][2] = wsum, red[wave][3] = maxcap;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nw = blockDim.x >> 6;
        for (uint32_t w = 1; w < nw; w++) {
            m = red[w][0] > m ? red[w][0] : m;
            longest = red[w][1] > longest ? red[w][1] : longest;
            wsum += red[w][2];
            maxcap = red[w][3] > maxcap ? red[w][3] : maxcap;
        }
        if (m) atomicMax(result, m);
        atomicMax(result + 1, longest);
        atomicAdd(result + 2, wsum);
        atomicMax(result + 3, maxcap);
    }
}

// ONE long v1 stream (tamp_decompress_long_kernel.hpp): start positions settled by rounds of a lane per 512 compressed bytes,
// records per chunk, then the split decoder's RESOLVE over groups of at most kSplitMaxOut output bytes, in order, each with the
// W bytes in front of it as its dictionary.  -> 1 when the call is not one (or anything is off: the exact decoders take it),
// TAMP_OK when the stream has been decoded, an error code otherwise.  Nothing is written before the fall-back decision.
int launch_decompress_long(DeviceCtx* ctx, const uint8_t* d_dict, size_t dict_len, uint8_t max_wbits, const uint8_t* d_in,
                           const uint64_t* d_in_off, const uint32_t* d_in_len, uint8_t* d_out, const uint64_t* d_out_off,
                           const uint32_t* d_out_cap, uint32_t* d_out_len, int8_t* d_status, uint32_t* d_consumed, hipStream_t st) {
    if (const char* e = getenv("TAMP_AMD_LONGDEC")) { if (atoi(e) == 0) return 1; }
    if (getenv("TAMP_AMD_DECODER")) return 1;
    uint32_t min_len = 256u << 10;
    if (const char* e = getenv("TAMP_AMD_LONGDEC_MIN")) { const long v = atol(e); if (v >= 64) min_len = (uint32_t)v; }
    uint64_t in_off = 0, out_off = 0;
    uint32_t n = 0, cap = 0;
    HIP_OK(hipMemcpyAsync(&n, d_in_len, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    // (the chunk kernels count bits in 32-bit registers: (i + 1) * kLongChunkBits wraps for the last chunk of the top 512 bytes of
    // the accepted range -- those streams stay with the exact decoder)
    if (n < min_len || n > kMaxDecodeIn - 512) return 1;
    HIP_OK(hipMemcpyAsync(&in_off, d_in_off, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&out_off, d_out_off, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&cap, d_out_cap, 4, hipMemcpyDeviceToHost, st));
    uint8_t hdr[2] = {0, 0};
    HIP_OK(hipMemcpyAsync(hdr, d_in + in_off, 2, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    const uint8_t* const in = d_in + in_off;
    uint8_t* const out = d_out + out_off;
    const uint32_t h0 = hdr[0], hs = 1 + (h0 & 1);
    const uint32_t wbits = ((h0 >> 5) & 7) + 8, lbits = ((h0 >> 3) & 3) + 5;
    const bool custom = (h0 >> 2) & 1, extended = (h0 >> 1) & 1, dreset = h0 & 1;
    if (dreset || (hs == 2 && hdr[1]) || wbits > (uint32_t)(max_wbits & 0x7F) || (max_wbits & 0x7F) > 15) return 1;
    if (extended && getenv("TAMP_AMD_LONGDEC_EXT") && atoi(getenv("TAMP_AMD_LONGDEC_EXT")) == 0) return 1;  // (tests: the exact decoder)
    const uint32_t W = 1u << wbits;
    if (custom && (!d_dict || dict_len < W)) return 1;
    // the fresh decoder's window: the custom dictionary, or the seeded table for the stream's literal size (v1: the literal >= 7
    // table whatever the literal size, decompressor.c:318-319)
    const uint32_t dict_sel = (!extended || lbits >= 7) ? 2u : (lbits == 6 ? 1u : 0u);
    const uint8_t* const dict0 = custom ? d_dict : ctx->seed_dicts + ((size_t)dict_sel << 15);

    const uint64_t total_bits = 8ull * n;
    const uint32_t chunk_bits = extended ? kLongChunkBitsExt : kLongChunkBits;
    const uint32_t N = (uint32_t)((total_bits + chunk_bits - 1) / chunk_bits);
    // scratch: g, g_next (N + 1 each), flags (4), ntok, outb, tokbase, rot (N each), the extended format's seven per-chunk tables,
    // then what the groups need
    const size_t b_tab = ((size_t)(14 * (size_t)N + 16) * 4 + 255) & ~(size_t)255;
    uint8_t* tab = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->long_mu);
        DeviceCtx::HostPipe::Grow& gb = ctx->long_scratch[st];
        // worst case records: the shortest token is a literal, 1 + literal bits
        const size_t max_tok = (size_t)(total_bits / (1 + lbits)) + 4096;
        const size_t b_groups = ((size_t)(max_tok / 256 + N + 64) * (16 + sizeof(LongGroup) + 4) + 255) & ~(size_t)255;
        const size_t bytes = b_tab + max_tok * 4 + b_groups + 4 * (size_t)(1u << 15) + 4096;
        if (gb.need(bytes) != hipSuccess) return 1;
        tab = static_cast<uint8_t*>(gb.p);
    }
    uint32_t* const g0 = reinterpret_cast<uint32_t*>(tab);
    uint32_t* const g1 = g0 + (N + 1);
    uint32_t* const flags = g1 + (N + 1);
    uint32_t* const d_ntok = flags + 8;
    uint32_t* const d_outb = d_ntok + N;
    uint32_t* const d_tokbase = d_outb + N;
    uint32_t* const d_rot = d_tokbase + N;
    uint32_t* const d_nspec = d_rot + N;          // (extended format from here)
    uint32_t* const d_specbase = d_nspec + N;
    uint32_t* const d_chunk_lag = d_specbase + N;  // 2 N
    uint32_t* const d_chunk_o0 = d_chunk_lag + 2 * (size_t)N;
    uint32_t* const d_chunk_lag0 = d_chunk_o0 + N;
    uint32_t* const d_lagbase = d_chunk_lag0 + N;
    uint32_t* const recs = reinterpret_cast<uint32_t*>(tab + b_tab);

    timing_begin(st);
    LongArgs la;
    memset(&la, 0, sizeof la);
    la.in = in, la.n = n, la.first_bit = 8 * hs, la.n_chunks = N, la.wbits = wbits, la.lbits = lbits;
    la.chunk_bits = chunk_bits, la.extended = extended ? 1u : 0u, la.nspec = d_nspec;
    la.flags = flags, la.ntok = d_ntok, la.outb = d_outb, la.tokbase = d_tokbase, la.rot = d_rot, la.recs = recs, la.write = 0;
    // start guesses: the chunk boundaries themselves (chunk 0: behind the header)
    {
        std::vector<uint32_t> init(N + 1);
        for (uint32_t i = 0; i <= N; i++) init[i] = i * chunk_bits;
        init[0] = 8 * hs;
        HIP_OK(hipMemcpyAsync(g0, init.data(), (size_t)(N + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    const uint32_t lg = (N + 63) / 64;
    uint32_t* cur = g0;
    uint32_t* nxt = g1;
    bool settled = false;
    int rounds = 0;
    for (int round = 0; round < 512 && !settled; round++, rounds++) {
        HIP_OK(hipMemsetAsync(flags, 0, 8, st));
        la.g = cur, la.g_next = nxt;
        hipLaunchKernelGGL(tamp_long_sync_kernel, dim3(lg), dim3(64), 0, st, la);
        uint32_t changed = 1;
        HIP_OK(hipMemcpyAsync(&changed, flags, 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        std::swap(cur, nxt);
        settled = changed == 0;
    }
    const bool dbg_long = getenv("TAMP_AMD_LONGDEC_DEBUG") != nullptr;
    if (dbg_long) fprintf(stderr, "[tamp_amd long decode] %u bytes, %u chunks, %d sync rounds, settled %d\n", n, N, rounds, (int)settled);
    if (!settled) { timing_end(st); return 1; }
    la.g = cur, la.g_next = nullptr, la.write = 0;
    HIP_OK(hipMemsetAsync(flags, 0, 8, st));
    hipLaunchKernelGGL(tamp_long_parse_kernel, dim3(lg), dim3(64), 0, st, la);
    std::vector<uint32_t> ntok(N), outb(N);
    uint32_t fl[2] = {0, 0};
    HIP_OK(hipMemcpyAsync(ntok.data(), d_ntok, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(outb.data(), d_outb, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(fl, flags, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (fl[1]) { timing_end(st); return 1; }  // an out-of-bounds offset: the exact decoder reports where
    // Extended format: the tokens that can write fewer bytes to the window than they produce are listed (a second parse), one
    // workgroup walks the list for window_pos at each of them, and what comes back per chunk is the lag behind it and the number
    // of lagging tokens in it (tamp_long_wp_kernel).
    std::vector<uint32_t> chunk_lag;  // per chunk: cumulative lag behind it, lagging tokens in it
    uint32_t* d_lag = nullptr;        // the lag lists
    if (extended) {
        std::vector<uint32_t> nspec(N), specbase(N);
        HIP_OK(hipMemcpyAsync(nspec.data(), d_nspec, (size_t)N * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        uint64_t n_entries = 0;
        for (uint32_t i = 0; i < N; i++) specbase[i] = (uint32_t)n_entries, n_entries += (uint64_t)nspec[i] + 1;
        if (n_entries > 0xFFFFFFF0ull) { timing_end(st); return 1; }
        uint32_t* d_spec = nullptr;
        const size_t n_wpblocks = (size_t)((n_entries + kLongWpBlock - 1) / kLongWpBlock);
        {
            std::lock_guard<std::mutex> lock(ctx->long_mu);
            DeviceCtx::HostPipe::Grow& sb = ctx->long_spec[st];
            // gap, token, bytes written per list entry; behind them the lag lists (at most one entry per listed token) and the
            // window_pos tables of the list's blocks (tamp_long_wp_kernel: 8 bytes per block and start value, 20 per block)
            if (sb.need((size_t)n_entries * (3 + 2) * 4 + n_wpblocks * ((size_t)W * 8 + 20) + 256) != hipSuccess) { (void)hipGetLastError(); timing_end(st); return 1; }
            d_spec = static_cast<uint32_t*>(sb.p);
        }
        la.specbase = d_specbase, la.spec_gap = d_spec, la.spec_kl = d_spec + n_entries, la.spec_written = d_spec + 2 * n_entries;
        d_lag = d_spec + 3 * n_entries;
        la.chunk_lag = d_chunk_lag, la.lag = d_lag;
        HIP_OK(hipMemcpyAsync(d_specbase, specbase.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
        la.write = 2;
        hipLaunchKernelGGL(tamp_long_parse_kernel, dim3(lg), dim3(64), 0, st, la);
        {
            LongWpArgs wa;
            wa.a = la, wa.n_entries = (uint32_t)n_entries, wa.n_blocks = (uint32_t)n_wpblocks;
            wa.f_cum = d_spec + 5 * n_entries;
            wa.markers = wa.f_cum + n_wpblocks * (size_t)W;
            wa.state = wa.markers + n_wpblocks;
            wa.f_wp = reinterpret_cast<uint16_t*>(wa.state + 4 * n_wpblocks);
            wa.f_nl = wa.f_wp + n_wpblocks * (size_t)W;
            hipLaunchKernelGGL(tamp_long_wp_kernel<0>, dim3((uint32_t)n_wpblocks), dim3(1024), 0, st, wa);
            hipLaunchKernelGGL(tamp_long_wp_kernel<1>, dim3(1), dim3(64), 0, st, wa);
            hipLaunchKernelGGL(tamp_long_wp_kernel<2>, dim3((uint32_t)n_wpblocks), dim3(256), 0, st, wa);
        }
        chunk_lag.resize(2 * (size_t)N);
        HIP_OK(hipMemcpyAsync(chunk_lag.data(), d_chunk_lag, 2 * (size_t)N * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));  // (also: specbase goes out of scope)
        for (uint32_t i = 0; i < N; i++)
            if (chunk_lag[2 * (size_t)i + 1] > kLongLagCap) { timing_end(st); return 1; }  // (more lagging tokens in one chunk than a group lists)
    }
    // groups of whole chunks: at most kSplitMaxOut output bytes, 2^20 - 1 records and kLongLagCap lagging tokens each
    const char* chain_env = getenv("TAMP_AMD_LONGDEC_CHAIN");
    const bool chain = extended || !chain_env || atoi(chain_env) != 0;
    const uint32_t group_out = chain ? kLongGroupOut : kSplitMaxOut;
    struct Group { uint64_t v0; uint32_t tok0, ntok, nout, lag0, nlag; uint64_t lagv0; };  // lagv0: lag of the stream in front of the group
    std::vector<Group> groups;
    std::vector<uint32_t> tokbase(N), rot(N), chunk_o0(N), chunk_lag0(N), lagbase(N);
    uint64_t v = 0, tk = 0, lags = 0, cum = 0;  // output bytes, records, lagging tokens, lag so far
    {
        Group gcur{0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = 0; i < N; i++) {
            const uint32_t nl = extended ? chunk_lag[2 * (size_t)i + 1] : 0u;
            if (gcur.nout + outb[i] > group_out || gcur.ntok + ntok[i] > 0xFFFFFu || gcur.nlag + nl > kLongLagCap) {
                groups.push_back(gcur);
                gcur = Group{v, (uint32_t)tk, 0, 0, (uint32_t)lags, 0, cum};
            }
            // a group's window cursor starts at (bytes WRITTEN in front of it) mod W: the output position less the lag so far
            tokbase[i] = (uint32_t)tk, rot[i] = (uint32_t)((gcur.v0 - gcur.lagv0) & (W - 1));
            chunk_o0[i] = (uint32_t)(v - gcur.v0), chunk_lag0[i] = (uint32_t)(cum - gcur.lagv0), lagbase[i] = (uint32_t)lags;
            gcur.ntok += ntok[i], gcur.nout += outb[i], gcur.nlag += nl;
            tk += ntok[i], v += outb[i], lags += nl;
            if (extended) cum = chunk_lag[2 * (size_t)i];
        }
        groups.push_back(gcur);
    }
    // (room that is used up -- even exactly -- is TAMP_OUTPUT_FULL in the reference when padding bits are left, decompressor.c:431-436,
    // and a partial last token when it is not enough: the exact decoder's)
    if (v >= cap || tk > 0xFFFFFFFFull - 4096) { timing_end(st); return 1; }
    HIP_OK(hipMemcpyAsync(d_tokbase, tokbase.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_rot, rot.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
    if (extended) {
        HIP_OK(hipMemcpyAsync(d_chunk_o0, chunk_o0.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_chunk_lag0, chunk_lag0.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_lagbase, lagbase.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
        la.chunk_o0 = d_chunk_o0, la.chunk_lag0 = d_chunk_lag0, la.lagbase = d_lagbase;
    }
    la.write = 1;
    hipLaunchKernelGGL(tamp_long_parse_kernel, dim3(lg), dim3(64), 0, st, la);
    // per group: meta word, output offset and size, in tables behind the records
    const size_t G = groups.size();
    uint8_t* const gtab = reinterpret_cast<uint8_t*>(recs) + (((size_t)tk + 4096) * 4 + 255 & ~(size_t)255);
    uint64_t* const d_goff = reinterpret_cast<uint64_t*>(gtab);
    uint32_t* const d_glen = reinterpret_cast<uint32_t*>(d_goff + G);
    uint32_t* const d_gmeta = d_glen + G;
    uint8_t* const d_win = reinterpret_cast<uint8_t*>(d_gmeta + G + 16);  // windows of the groups that start inside the first W bytes
    {
        std::vector<uint64_t> goff(G);
        std::vector<uint32_t> glen(G), gmeta(G);
        for (size_t k = 0; k < G; k++) {
            goff[k] = out_off + groups[k].v0, glen[k] = groups[k].nout;
            gmeta[k] = (groups[k].ntok & 0xFFFFFu) | ((wbits - 8) << 20) | (3u << 23);
        }
        HIP_OK(hipMemcpyAsync(d_goff, goff.data(), G * 8, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_glen, glen.data(), G * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_gmeta, gmeta.data(), G * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));  // (the vectors go out of scope)
    }
    if (dbg_long) fprintf(stderr, "[tamp_amd long decode] %zu groups, %llu tokens, %llu bytes out\n", G, (unsigned long long)tk, (unsigned long long)v);
    if (chain) {
        // every group by a workgroup of its own, no workgroup waiting for another (tamp_decompress_long_kernel.hpp, step 3):
        // tail maps, their composition by one workgroup, finish.  The group table sits behind the tables above, the maps
        // (G x W x 2 bytes) in a buffer of their own.
        uint8_t* const ctab = d_win + 4 * (size_t)(1u << 15);
        LongGroup* const d_groups = reinterpret_cast<LongGroup*>(ctab);
        uint16_t* d_maps = nullptr;
        const size_t n_blocks = (G + kLongScanBlock - 1) / kLongScanBlock;
        {
            std::lock_guard<std::mutex> lock(ctx->long_mu);
            DeviceCtx::HostPipe::Grow& mb = ctx->long_maps[st];
            // (behind the groups' maps: the blocks' maps and the window in front of every block)
            // (... and, extended format, in front of every group: with lags the window is not "the last W output bytes")
            if (mb.need((G + n_blocks) * (size_t)W * 2 + (n_blocks + (extended ? G : 0)) * (size_t)W + 256) != hipSuccess) { (void)hipGetLastError(); timing_end(st); return 1; }
            d_maps = static_cast<uint16_t*>(mb.p);
        }
        uint16_t* const d_blockmap = d_maps + G * (size_t)W;
        uint8_t* const d_blockwin = reinterpret_cast<uint8_t*>(d_blockmap + n_blocks * (size_t)W);
        uint8_t* const d_groupwin = extended ? d_blockwin + n_blocks * (size_t)W : nullptr;
        {
            std::vector<LongGroup> tabv(G);
            for (size_t k = 0; k < G; k++)
                tabv[k] = LongGroup{groups[k].v0, groups[k].tok0, groups[k].ntok, groups[k].nout, groups[k].lag0, groups[k].nlag, 0};
            HIP_OK(hipMemcpyAsync(d_groups, tabv.data(), G * sizeof(LongGroup), hipMemcpyHostToDevice, st));
            HIP_OK(hipStreamSynchronize(st));
        }
        LongResolveArgs ra;
        ra.recs = recs, ra.groups = d_groups, ra.out = out, ra.dict0 = dict0, ra.tailmap = d_maps, ra.wbits = wbits;
        ra.lag = d_lag, ra.groupwin = d_groupwin;
        ra.n_groups = (uint32_t)G;
        auto k_tails = extended ? tamp_long_resolve_kernel<1, true> : tamp_long_resolve_kernel<1, false>;
        auto k_finish = extended ? tamp_long_resolve_kernel<2, true> : tamp_long_resolve_kernel<2, false>;
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tails), hipFuncAttributeMaxDynamicSharedMemorySize, (int)long_resolve_lds()));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_finish), hipFuncAttributeMaxDynamicSharedMemorySize, (int)long_resolve_lds()));
        LongScanArgs sc;
        sc.r = ra, sc.blockmap = d_blockmap, sc.blockwin = d_blockwin, sc.n_blocks = (uint32_t)n_blocks;
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_long_tail_scan_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * W)));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_long_tail_scan_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * W)));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_long_tail_scan_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * W)));
        hipLaunchKernelGGL(k_tails, dim3((uint32_t)G), dim3(256), long_resolve_lds(), st, ra);
        hipLaunchKernelGGL(tamp_long_tail_scan_kernel<0>, dim3((uint32_t)n_blocks), dim3(kLongScanThreads), 4 * W, st, sc);
        hipLaunchKernelGGL(tamp_long_tail_scan_kernel<1>, dim3(1), dim3(kLongScanThreads), 2 * W, st, sc);
        hipLaunchKernelGGL(tamp_long_tail_scan_kernel<2>, dim3((uint32_t)n_blocks), dim3(kLongScanThreads), 2 * W, st, sc);
        hipLaunchKernelGGL(k_finish, dim3((uint32_t)G), dim3(256), long_resolve_lds(), st, ra);
        hipLaunchKernelGGL(tamp_long_finish_kernel, dim3(1), dim3(1), 0, st, d_out_len, d_status, d_consumed, (uint32_t)v, n);
        timing_end(st);
        HIP_OK(hipGetLastError());
        return TAMP_OK;
    }
    SplitArgs sa;
    DecompressArgs& a = sa.d;
    a.in = d_in, a.in_off = d_in_off, a.in_len = d_in_len, a.out = d_out, a.out_cap = d_out_cap, a.status = d_status;
    a.in_consumed = nullptr, a.dict_len = W, a.seed_dicts = ctx->seed_dicts, a.scratch = nullptr, a.only_flagged = nullptr;
    a.flagged_count = nullptr, a.n_streams = 1, a.lds_row = 0, a.max_wbits = (uint8_t)wbits;
    sa.lag = nullptr, sa.flagged = nullptr, sa.flagged_count = nullptr, sa.maxcap = kSplitMaxOut, sa.first = 0, sa.count = 1, sa.spw = 64;
    const uint32_t lds = split_resolve_lds(kSplitMaxOut);
    auto resolve_kernel = tamp_decode_resolve_kernel<256, 4>;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t early = 0;
    for (size_t k = 0; k < G; k++) {
        const Group& gr = groups[k];
        if (gr.nout == 0) continue;
        if (gr.v0 < W) {
            if (early >= 4) { timing_end(st); return TAMP_ERROR; }  // (cannot happen: groups hold >= 11 KiB unless they are the last)
            uint8_t* const wbuf = d_win + (size_t)early * (1u << 15);
            early++;
            hipLaunchKernelGGL(tamp_long_window_kernel, dim3((W + 255) / 256), dim3(256), 0, st, wbuf, out, dict0, (uint32_t)gr.v0, W);
            a.dict = wbuf;
        } else {
            a.dict = out + gr.v0 - W;
        }
        a.out_off = d_goff + k, a.out_len = d_glen + k;
        sa.recs = recs + gr.tok0, sa.meta = d_gmeta + k;
        sa.tokcap = gr.ntok > 2048 ? gr.ntok : 2048;
        hipLaunchKernelGGL(resolve_kernel, dim3(1), dim3(256), lds, st, sa);
    }
    hipLaunchKernelGGL(tamp_long_finish_kernel, dim3(1), dim3(1), 0, st, d_out_len, d_status, d_consumed, (uint32_t)v, n);
    timing_end(st);
    HIP_OK(hipGetLastError());
    return TAMP_OK;
}

int launch_decompress(DeviceCtx* ctx, const uint8_t* d_dict, size_t dict_len, uint8_t max_wbits, const uint8_t* d_in,
                      const uint64_t* d_in_off, const uint32_t* d_in_len, uint8_t* d_out, const uint64_t* d_out_off,
                      const uint32_t* d_out_cap, uint32_t* d_out_len, int8_t* d_status, uint32_t* d_consumed,
                      size_t n_streams, hipStream_t st) {
    if (n_streams == 0) return TAMP_OK;
    DecompressArgs a;
    a.in = d_in, a.in_off = d_in_off, a.in_len = d_in_len;
    a.out = d_out, a.out_off = d_out_off, a.out_cap = d_out_cap, a.out_len = d_out_len, a.status = d_status;
    a.in_consumed = d_consumed;
    a.dict = d_dict, a.dict_len = (uint32_t)(dict_len > 0xFFFFFFFFu ? 0xFFFFFFFFu : dict_len);
    a.seed_dicts = ctx->seed_dicts;
    a.scratch = nullptr;
    a.only_flagged = nullptr;
    a.flagged_count = nullptr;
    a.n_streams = (uint32_t)n_streams;
    DeviceCtx::Slab* call_slab = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        call_slab = &ctx->slabs[st];  // (map nodes do not move)
    }
    std::lock_guard<std::mutex> call_lock(call_slab->launch_mu);
    if (n_streams <= 16 && !(max_wbits & TAMP_AMD_WINDOW_BITS_EXACT)) {
        // one long v1 stream -- or a handful, one after the other: the whole device each (tamp_decompress_long_kernel.hpp).  A stream
        // that is not one (too short, extended, ...) sends the whole call to the decoders below, which write every stream again.
        // (one event pair around all of them: kernel_ms of a call with several long streams is the sum, not the last stream's)
        size_t done = 0;
        timing_begin(st);
        t_timing_outer = true;
        int rc = TAMP_OK;
        for (; done < n_streams; done++) {
            rc = launch_decompress_long(ctx, d_dict, dict_len, max_wbits, d_in, d_in_off + done, d_in_len + done, d_out,
                                        d_out_off + done, d_out_cap + done, d_out_len + done, d_status + done,
                                        d_consumed ? d_consumed + done : nullptr, st);
            if (rc != TAMP_OK) break;
        }
        t_timing_outer = false;
        if (rc != TAMP_OK && rc != 1) return rc;
        if (done == n_streams) return TAMP_OK;
    }
    const bool exact = (max_wbits & TAMP_AMD_WINDOW_BITS_EXACT) != 0;
    uint32_t longest_in = 0xFFFFFFFFu;  // longest compressed stream of the batch (unknown without the pre-pass)
    uint64_t window_bytes = 0;          // sum of the streams' window sizes (0 = unknown)
    uint32_t max_out_cap = 0;           // largest out_cap of the batch (0 = unknown)
    max_wbits &= 0x7F;
    const char* force = getenv("TAMP_AMD_DECODER");  // "wave" | "lane" | "global" | "split" (tuning / tests)
    const bool force_split = force && force[0] == 's';
    if (!exact && max_wbits >= 8 && max_wbits <= 15 && ((max_wbits > 8 && n_streams >= 256) || force_split)) {
        uint32_t* hdr_scan = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            DeviceCtx::Slab& slab = ctx->slabs[st];
            if (!slab.scan) HIP_OK(hipMalloc(&slab.scan, 32));
            hdr_scan = slab.scan;
        }
        uint32_t scan[4] = {0, 0, 0, 0};
        uint32_t& found = scan[0];
        HIP_OK(hipMemsetAsync(hdr_scan, 0, 16, st));
        const uint32_t sg = (uint32_t)std::min<size_t>((n_streams + 255) / 256, (size_t)ctx->cu_count * 4);
        hipLaunchKernelGGL(tamp_header_scan_kernel, dim3(sg), dim3(256), 0, st, d_in, d_in_off, d_in_len, d_out_cap,
                           (uint32_t)n_streams, (uint32_t)max_wbits, hdr_scan);
        HIP_OK(hipMemcpyAsync(scan, hdr_scan, 16, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        // streams above the limit fail with TAMP_INVALID_CONF under either value; nothing valid exceeds `found`
        if (found >= 8 && found < max_wbits) max_wbits = (uint8_t)found;
        if (found == 0) max_wbits = 8;
        longest_in = scan[1];
        window_bytes = (uint64_t)scan[2] << 8;
        max_out_cap = scan[3];
    }
    a.max_wbits = max_wbits;
    a.lds_row = 0;
    const bool valid_bits = max_wbits >= 8 && max_wbits <= 15;
    // Split decoder (tamp_decompress_split_kernel.hpp): parse one lane per stream without any window, resolve one
    // workgroup (out_cap above 2 KiB) or one wavefront per stream by pointer jumping; what it flags is decoded by the wave
    // decoder afterwards.  Needs the pre-pass (longest stream and largest out_cap size its scratch and LDS).
    // Taken for batches of streams of 512 compressed bytes and more, and for most batches of short messages (below), whose
    // output slabs fit RESOLVE's LDS.
    const bool split_fits = valid_bits && max_out_cap && max_out_cap <= kSplitMaxOut && longest_in != 0xFFFFFFFFu;
    // Short messages (round 4, with RESOLVE's wavefront-per-stream build and the parse's whole-stream ring): the split decoder
    // has no window to set up, the lane decoders fill one per message -- from the caller's dictionary, or 2^9 bytes and more
    // of the seeded one.  1 Mi x 256 B: custom dictionary at w = 8 1.30 against 2.18 ms, default window 2^10 2.01 against
    // 6.13 ms, 1 Mi x 512 B at w = 9 3.2 against 20.4 ms; only w = 8 without a dictionary stays with the LDS lanes (1.65
    // against 1.98 ms).  tools/dec_short.py.
    const bool short_split = longest_in < 512 && (d_dict != nullptr || max_wbits >= 9);
    const bool want_split = force ? force_split : ((longest_in >= 512 || short_split) && n_streams >= 256);
    if (want_split && split_fits) {
        SplitArgs sa;
        sa.maxcap = max_out_cap;
        sa.tokcap = std::max<uint32_t>(16, (uint32_t)std::min<uint64_t>(max_out_cap, (uint64_t)longest_in * 8 / 6 + 8));
        sa.tokcap = (sa.tokcap + 15u) & ~15u;  // whole 64-byte groups of records per stream
        // Streams per slice: 256 Ki = 4 parse waves per SIMD (measured on configs[3], 1 Mi streams: 2^17 29.4 ms, 2^18 25.4 ms,
        // 2^19 26.4 ms; TAMP_AMD_SPLIT_SLICE_LOG2 overrides), less when the scratch budget says so (records dominate:
        // tokcap x 4 B per stream; TAMP_AMD_SPLIT_SCRATCH_MB, default 8 GiB of the 288 GB).
        size_t slice_log2 = 18;
        if (const char* e = getenv("TAMP_AMD_SPLIT_SLICE_LOG2")) { const int v = atoi(e); if (v >= 12 && v <= 22) slice_log2 = (size_t)v; }
        size_t slice = std::min<size_t>(n_streams, (size_t)1 << slice_log2);
        const size_t per = (size_t)sa.tokcap * 4 + 4 + kSplitMaxLag * 8;
        {
            // scratch budget: a quarter of what the device has free right now, 8 GiB at most (callers that fill HBM with
            // their own batches keep most of it; a slice of 2^18 long streams needs ~3.7 GiB, and configs[3] cut into
            // uneven slices by a 4 GiB budget ran 6.5 instead of 5.1 ms); TAMP_AMD_SPLIT_SCRATCH_MB overrides
            size_t budget = (size_t)8 << 30, free_b = 0, total_b = 0;
            // (the slab this stream already holds is part of what the call may use: without it the budget -- and with it
            // the slice size, hence the decode time -- of the second call on a shape differed from the first's)
            size_t held = 0;
            {
                std::lock_guard<std::mutex> lock(g_mu);
                held = ctx->slabs[st].split_bytes;
            }
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, std::max((free_b + held) / 4, held));
            else (void)hipGetLastError();
            if (const char* e = getenv("TAMP_AMD_SPLIT_SCRATCH_MB")) { const long v = atol(e); if (v > 0) budget = (size_t)v << 20; }
            slice = std::min(slice, std::max<size_t>(budget / per, 4096));
        }
        // The slab is kept per HIP stream between calls (tamp_amd_trim() releases it).  If the device cannot supply it the
        // slice is halved down to 4,096 streams, and below that the batch goes to the lane / wave decoders, which need
        // little or no scratch: an allocation failure here must not fail a call that another decoder can serve.
        uint8_t* base = nullptr;
        size_t b_recs = 0, b_meta = 0, b_lag = 0;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            DeviceCtx::Slab& slab = ctx->slabs[st];
            for (;;) {
                b_recs = slice * sa.tokcap * 4, b_meta = slice * 4, b_lag = slice * kSplitMaxLag * 8;
                const size_t need = b_recs + b_meta + b_lag + n_streams + 64;
                if (slab.split_bytes >= need) break;
                if (slab.split) {
                    HIP_OK(hipStreamSynchronize(st));
                    HIP_OK(hipFree(slab.split));
                    slab.split = nullptr, slab.split_bytes = 0;
                }
                const bool deny = getenv("TAMP_AMD_SPLIT_FAIL_ABOVE") && need > (size_t)atol(getenv("TAMP_AMD_SPLIT_FAIL_ABOVE"));  // (tests)
                if (!deny && hipMalloc(&slab.split, need) == hipSuccess) {
                    slab.split_bytes = need;
                    break;
                }
                (void)hipGetLastError();  // clear the sticky out-of-memory error
                slab.split = nullptr;
                if (slice <= 4096) { slice = 0; break; }
                slice = std::max<size_t>(slice / 2, 4096);
            }
            base = slab.split;
        }
        if (slice) {
        sa.recs = reinterpret_cast<uint32_t*>(base);
        sa.meta = reinterpret_cast<uint32_t*>(base + b_recs);
        sa.lag = reinterpret_cast<uint32_t*>(base + b_recs + b_meta);
        sa.flagged = base + b_recs + b_meta + b_lag;
        sa.flagged_count = reinterpret_cast<uint32_t*>(sa.flagged + ((n_streams + 3) & ~(size_t)3));  // (inside the 64 bytes of slack)
        HIP_OK(hipMemsetAsync(sa.flagged_count, 0, 4, st));
        sa.d = a;
        // resolve: a workgroup per stream, or -- short messages, out_cap up to 1 KiB -- a wavefront per stream, four per workgroup
        bool wave_resolve = max_out_cap <= kSplitWaveMaxOut;
        if (const char* e = getenv("TAMP_AMD_SPLIT_WAVE_MAX")) {  // (tuning; the one-wavefront RESOLVE covers 4 x 16 x 64 = 4,096 positions)
            const int v = atoi(e);
            wave_resolve = max_out_cap <= (uint32_t)(v < 0 ? 0 : (v > 4096 ? 4096 : v));
        }
        const uint32_t lds = split_resolve_lds(max_out_cap) * (wave_resolve ? 4u : 1u);
        auto resolve_kernel = wave_resolve ? tamp_decode_resolve_kernel<64, 4> : tamp_decode_resolve_kernel<256, 4>;
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(resolve_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        timing_begin(st);
        for (size_t first = 0; first < n_streams; first += slice) {
            sa.first = (uint32_t)first;
            sa.count = (uint32_t)std::min(slice, n_streams - first);
            // streams per wave: enough waves for ~4 per SIMD (tools/dec_split_pmc.sh: the parse runs at one wave's latency)
            const size_t want_waves = (size_t)ctx->cu_count * 16;
            sa.spw = sa.count / 16 < want_waves ? 16 : (sa.count / 32 < want_waves ? 32 : 64);
            if (const char* e = getenv("TAMP_AMD_SPLIT_SPW")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) sa.spw = (uint32_t)v; }
            const uint32_t pwaves = (sa.count + sa.spw - 1) / sa.spw;
            hipLaunchKernelGGL(tamp_decode_parse_kernel, dim3((pwaves + 3) / 4), dim3(256), split_parse_lds(256), st, sa);
            hipLaunchKernelGGL(resolve_kernel, dim3(wave_resolve ? (sa.count + 3) / 4 : sa.count), dim3(256), lds, st, sa);
        }
        // leftovers: the wave decoder over the flagged streams only
        a.only_flagged = sa.flagged;
        a.flagged_count = sa.flagged_count;
        const uint32_t waves = max_wbits <= 12 ? 4 : 1;
        const uint32_t wlds = decode_wave_lds(max_wbits, waves);
        size_t groups = (n_streams + waves - 1) / waves;
        groups = std::min(groups, (size_t)ctx->cu_count * 64);
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_decompress_wave_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        hipLaunchKernelGGL(tamp_decompress_wave_kernel, dim3((uint32_t)groups), dim3(waves * kWave), wlds, st, a);
        timing_end(st);
        HIP_OK(hipGetLastError());
        return TAMP_OK;
        }  // (no scratch to be had: fall through to the lane / wave decoders)
    }
    // Decoder choice: one wavefront per stream (scalar token loop, window in LDS, 64-lane copies) unless the batch is
    // a very large number of streams, where one lane per stream fills the chip and avoids per-stream set-up.
    // Three decoders.  Wave per stream: time follows the total bytes, needs few streams.  Lane per
    // stream with the windows in LDS: rounds of `capacity` streams (the rows limit the resident lanes), a round lasts as
    // long as its longest stream, about twice as fast per byte -- taken when a single round is reasonably full, and for
    // batches of short messages.  Lane per stream with the windows in a global scratch slab: no capacity limit, every
    // wave resident at once and the memory latency hidden by the other waves of the SIMD -- taken for large batches of
    // long streams, whatever their windows (mixed-window batches included).
    const bool bulk = longest_in >= 512;  // short messages: the lean lane build (no bulk path, smaller rows)
    const bool force_global = force && force[0] == 'g';
    bool lds_lanes = false, global_lanes = false;
    if (valid_bits) {
        size_t capacity = 0;
        if (max_wbits <= kLdsWinBits) {
            const uint32_t lds = bulk ? lane_decoder_lds(max_wbits) : kWave * ((1u << max_wbits) + 4);
            capacity = (size_t)ctx->cu_count * std::min<size_t>(160 * 1024 / lds, 16) * kWave;
        }
        if (!bulk) {
            const size_t rounds = capacity ? (n_streams + capacity - 1) / capacity : 1;
            lds_lanes = capacity && n_streams * 10 >= rounds * capacity * 2;
        } else if (capacity && max_wbits <= 9) {
            // small windows: four and more waves of rows fit a CU's LDS, nothing beats that
            const size_t rounds = (n_streams + capacity - 1) / capacity;
            lds_lanes = n_streams * 10 >= rounds * capacity * 6;
        } else {
            lds_lanes = capacity && n_streams * 10 >= capacity * 6 && n_streams * 4 <= capacity * 5;
            global_lanes = !lds_lanes && n_streams >= (size_t)ctx->cu_count * 192;  // ~3/4 wave per SIMD and up
        }
    }
    if (force) lds_lanes = force[0] == 'l', global_lanes = force_global;
    const bool use_wave = force ? (force[0] == 'w') : !(lds_lanes || global_lanes);
    if (valid_bits && use_wave) {
        const uint32_t waves = max_wbits <= 12 ? 4 : 1;
        const uint32_t lds = decode_wave_lds(max_wbits, waves);
        size_t groups = (n_streams + waves - 1) / waves;
        const size_t resident = (size_t)ctx->cu_count * 64;
        if (groups > resident) groups = resident;  // grid-stride beyond that
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_decompress_wave_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        timing_begin(st);
        hipLaunchKernelGGL(tamp_decompress_wave_kernel, dim3((uint32_t)groups), dim3(waves * kWave), lds, st, a);
        timing_end(st);
        HIP_OK(hipGetLastError());
        return TAMP_OK;
    }
    if (valid_bits && max_wbits <= kLdsWinBits && !global_lanes) {
        // windows in LDS: one 64-lane workgroup per 64 streams, one padded row per lane
        a.lds_row = (1u << max_wbits) + (bulk ? kLaneRowPad : 4u);
        const uint32_t lds = bulk ? lane_decoder_lds(max_wbits) : kWave * a.lds_row;
        const uint32_t per_cu = (uint32_t)(160 * 1024 / lds) < 16 ? (uint32_t)(160 * 1024 / lds) : 16;
        size_t groups = (n_streams + kWave - 1) / kWave;
        const size_t resident = (size_t)ctx->cu_count * per_cu;
        if (groups > resident * 4) groups = resident * 4;  // grid-stride beyond a few waves of workgroups
        auto lane_kernel = bulk ? tamp_decompress_kernel<true, true> : tamp_decompress_kernel<true, false>;
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(lane_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        timing_begin(st);
        hipLaunchKernelGGL(lane_kernel, dim3((uint32_t)groups), dim3(kWave), lds, st, a);
        timing_end(st);
        HIP_OK(hipGetLastError());
        return TAMP_OK;
    }
    const uint32_t threads = 256;
    const uint8_t slot_bits = valid_bits ? max_wbits : 8;
    const bool gbulk = valid_bits && bulk;  // bulk path with the windows in the scratch slab (slots padded like LDS rows)
    const size_t slot = ((size_t)1 << slot_bits) + (gbulk ? 64 : 0);
    a.lds_row = (uint32_t)slot;
    // Resident lanes.  Every step touches the lane's window at random: the kernel runs at the speed of the Infinity
    // Cache (256 MB) as long as the windows in flight fit into it, and of HBM sector traffic beyond.  So: as many lanes
    // as ~144 MB of live windows allow, but at least one wave per SIMD (and no more than eight).
    size_t lanes = (size_t)ctx->cu_count * 2048;
    {
        const size_t avg_window = window_bytes ? std::max<size_t>(256, (size_t)(window_bytes / n_streams)) : ((size_t)1 << slot_bits);
        size_t fit = ((size_t)144 << 20) / avg_window;
        if (const char* e = getenv("TAMP_AMD_SCRATCH_MB")) fit = ((size_t)atoi(e) << 20) / avg_window;  // tuning
        lanes = std::min(lanes, std::max(fit, (size_t)ctx->cu_count * 256));
    }
    const size_t budget = (size_t)4 << 30;  // hard bound of the slab
    if (lanes * slot > budget) lanes = budget / slot;
    if (lanes > n_streams) lanes = n_streams;
    const uint32_t grid = (uint32_t)((lanes + threads - 1) / threads);
    const size_t need = (size_t)grid * threads * slot;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        DeviceCtx::Slab& slab = ctx->slabs[st];
        if (slab.bytes < need) {
            if (slab.p) {
                HIP_OK(hipStreamSynchronize(st));  // earlier launches on this stream still use the old slab
                HIP_OK(hipFree(slab.p));
                slab.p = nullptr;
                slab.bytes = 0;
            }
            HIP_OK(hipMalloc(&slab.p, need));
            slab.bytes = need;
        }
        a.scratch = slab.p;
    }
    timing_begin(st);
    if (gbulk)
        hipLaunchKernelGGL((tamp_decompress_kernel<false, true>), dim3(grid), dim3(threads), 128 + threads * kLaneStagePad, st, a);
    else
        hipLaunchKernelGGL((tamp_decompress_kernel<false, false>), dim3(grid), dim3(threads), 0, st, a);
    timing_end(st);
    HIP_OK(hipGetLastError());
    return TAMP_OK;
}

// Resumable decoding: one wavefront per decoder object (tamp_decompress_resume_kernel.hpp).
int launch_decompress_resume(DeviceCtx* ctx, uint8_t* d_states, size_t stride, uint8_t bits_max, const uint8_t* d_in,
                             const uint64_t* d_in_off, const uint32_t* d_in_len, uint8_t* d_out,
                             const uint64_t* d_out_off, const uint32_t* d_out_cap, uint32_t* d_out_len, int8_t* d_status,
                             uint32_t* d_consumed, size_t n_streams, hipStream_t st) {
    if (n_streams == 0) return TAMP_OK;
    ResumeArgs ra;
    DecompressArgs& a = ra.d;
    a.in = d_in, a.in_off = d_in_off, a.in_len = d_in_len;
    a.out = d_out, a.out_off = d_out_off, a.out_cap = d_out_cap, a.out_len = d_out_len, a.status = d_status;
    a.in_consumed = d_consumed;
    a.dict = nullptr, a.dict_len = 0;  // a custom dictionary is the initial content of the object's window
    a.seed_dicts = ctx->seed_dicts;
    a.scratch = nullptr;
    a.only_flagged = nullptr;
    a.flagged_count = nullptr;
    a.n_streams = (uint32_t)n_streams;
    a.lds_row = 0;
    a.max_wbits = bits_max;
    ra.states = d_states, ra.state_stride = stride;
    const uint32_t waves = bits_max <= 12 ? 4 : 1;
    const uint32_t lds = decode_wave_lds(bits_max, waves);
    size_t groups = (n_streams + waves - 1) / waves;
    groups = std::min(groups, (size_t)ctx->cu_count * 64);  // grid-stride beyond that
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_decompress_resume_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    timing_begin(st);
    hipLaunchKernelGGL(tamp_decompress_resume_kernel, dim3((uint32_t)groups), dim3(waves * kWave), lds, st, ra);
    timing_end(st);
    HIP_OK(hipGetLastError());
    return TAMP_OK;
}

// Compressor objects below flush granularity: one wavefront per object (tamp_compress_resume_kernel.hpp).
int launch_compress_resume(DeviceCtx* ctx, uint8_t* d_states, size_t stride, uint8_t bits_max, int op, int write_token,
                           const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint8_t* d_out,
                           const uint64_t* d_out_off, const uint32_t* d_out_cap, uint32_t* d_out_len, int8_t* d_status,
                           uint32_t* d_consumed, size_t n, hipStream_t st) {
    if (n == 0) return TAMP_OK;
    EncodeResumeArgs a;
    a.states = d_states, a.state_stride = stride;
    a.in = d_in, a.in_off = d_in_off, a.in_len = d_in_len;
    a.out = d_out, a.out_off = d_out_off, a.out_cap = d_out_cap, a.out_len = d_out_len, a.status = d_status;
    a.in_consumed = d_consumed;
    a.n_objects = (uint32_t)n, a.op = (uint32_t)op, a.write_token = write_token ? 1u : 0u;
    a.max_wbits = bits_max;
    const uint32_t waves = bits_max <= 12 ? 4 : 1;
    const uint32_t lds = encode_resume_lds(bits_max, waves);
    size_t groups = (n + waves - 1) / waves;
    groups = std::min(groups, (size_t)ctx->cu_count * 64);
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(tamp_compress_resume_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    timing_begin(st);
    hipLaunchKernelGGL(tamp_compress_resume_kernel, dim3((uint32_t)groups), dim3(waves * kWave), lds, st, a);
    timing_end(st);
    HIP_OK(hipGetLastError());
    return TAMP_OK;
}

bool conf_valid(const TampAmdConf* c) {
    return c && c->window >= 8 && c->window <= 15 && c->literal >= 5 && c->literal <= 8;  // compressor.c:208-209
}

// ---------------------------------------------------------------------------------------------
// Host-memory batches.  The caller's arrays live in host memory (pageable or pinned); the batch is cut into chunks of
// consecutive streams and each chunk goes copy-in -> kernel -> copy-out on one of three library streams, so the PCIe
// transfers of neighbouring chunks (both directions) overlap the kernels.  A feeder thread issues copy-in + launch,
// the calling thread issues copy-out: with pageable memory hipMemcpyAsync blocks its caller, and two issuers keep
// both directions busy anyway.  Device staging is kept between calls.
// ---------------------------------------------------------------------------------------------
struct HostBatch {
    const uint8_t* in;
    const uint64_t* in_off;
    const uint32_t* in_len;
    uint8_t* out;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    uint32_t* out_len;
    int8_t* status;
    uint32_t* in_consumed;  // decompress only, may be null
    size_t n;
};

struct HostChunk {
    size_t i0, i1;
    uint64_t in_lo, in_hi, out_lo, out_hi;
    // Output slabs tile [out_lo, out_hi) exactly, in stream order: the copy-back may be ONE transfer of the whole
    // extent (bytes of a slab behind out_len[i] are unspecified afterwards, include/tamp_amd.h).  Otherwise -- gaps
    // between slabs, permuted or overlapping extents -- only the out_len[i] bytes of each stream are copied, so
    // nothing outside the produced bytes is ever written in the caller's buffer.
    bool out_packed;
};

// Consecutive streams: a chunk closes once it holds `min_streams` streams and `min_bytes` of data (the larger of its
// input and output extents), or earlier at `max_bytes`.  Needs both offset tables ascending and disjoint, the layout
// every packed batch has; otherwise one chunk covers the whole batch.
void plan_host_chunks(const HostBatch& b, size_t min_streams, uint64_t min_bytes, uint64_t max_bytes,
                      std::vector<HostChunk>& chunks) {
    bool ordered = true;
    uint64_t in_end = 0, out_end = 0, in_max = 0, out_max = 0, in_min = ~0ull, out_min = ~0ull;
    for (size_t i = 0; i < b.n; i++) {
        ordered = ordered && b.in_off[i] >= in_end && b.out_off[i] >= out_end;
        in_end = b.in_off[i] + b.in_len[i], out_end = b.out_off[i] + b.out_cap[i];
        in_max = std::max(in_max, in_end), out_max = std::max(out_max, out_end);
        in_min = std::min(in_min, b.in_off[i]), out_min = std::min(out_min, b.out_off[i]);
    }
    if (!ordered) {
        chunks.push_back({0, b.n, in_min, in_max, out_min, out_max, false});
        return;
    }
    size_t i0 = 0;
    while (i0 < b.n) {
        size_t i1 = i0 + 1;
        for (; i1 < b.n; i1++) {
            const uint64_t have = std::max(b.in_off[i1 - 1] + b.in_len[i1 - 1] - b.in_off[i0],
                                           b.out_off[i1 - 1] + b.out_cap[i1 - 1] - b.out_off[i0]);
            const uint64_t with = std::max(b.in_off[i1] + b.in_len[i1] - b.in_off[i0],
                                           b.out_off[i1] + b.out_cap[i1] - b.out_off[i0]);
            if ((i1 - i0 >= min_streams && have >= min_bytes) || with > max_bytes) break;
        }
        bool packed = true;
        for (size_t i = i0 + 1; i < i1 && packed; i++) packed = b.out_off[i] == b.out_off[i - 1] + b.out_cap[i - 1];
        chunks.push_back({i0, i1, b.in_off[i0], b.in_off[i1 - 1] + b.in_len[i1 - 1], b.out_off[i0],
                          b.out_off[i1 - 1] + b.out_cap[i1 - 1], packed});
        i0 = i1;
    }
    // a short last chunk is a badly filled launch: give it to its neighbour
    if (chunks.size() >= 2 && chunks.back().i1 - chunks.back().i0 < min_streams / 2) {
        const HostChunk last = chunks.back();
        chunks.pop_back();
        HostChunk& prev = chunks.back();
        prev.out_packed = prev.out_packed && last.out_packed && b.out_off[last.i0] == prev.out_hi;
        prev.i1 = last.i1, prev.in_hi = last.in_hi, prev.out_hi = last.out_hi;
    }
}

struct HostSlot {  // device views of one chunk: offsets stay absolute, the data pointers are shifted instead
    const uint8_t* in;
    uint8_t* out;
    const uint64_t *in_off, *out_off;
    const uint32_t *in_len, *out_cap;
    uint32_t *out_len, *in_consumed;
    int8_t* status;
};
using HostLaunch = std::function<int(const HostSlot&, size_t count, const uint8_t* d_dict, hipStream_t)>;

int run_host_batch(DeviceCtx* ctx, int device, const HostBatch& b, const std::vector<HostChunk>& chunks,
                   const uint8_t* dictionary, size_t dictionary_len, const HostLaunch& launch) {
    using Pipe = DeviceCtx::HostPipe;
    Pipe& P = ctx->pipe;
    std::lock_guard<std::mutex> call_lock(P.mu);
    const uint8_t* d_dict = nullptr;
    if (dictionary && dictionary_len) {
        HIP_OK(P.dict.need(kSeedTable));
        HIP_OK(hipMemcpy(P.dict.p, dictionary, std::min(dictionary_len, kSeedTable), hipMemcpyHostToDevice));
        d_dict = static_cast<const uint8_t*>(P.dict.p);
    }
    size_t max_in = 0, max_out = 0, max_cnt = 0;
    for (const HostChunk& ch : chunks) {
        max_in = std::max<size_t>(max_in, ch.in_hi - ch.in_lo);
        max_out = std::max<size_t>(max_out, ch.out_hi - ch.out_lo);
        max_cnt = std::max(max_cnt, ch.i1 - ch.i0);
    }
    const int depth = (int)std::min<size_t>(Pipe::kDepth, chunks.size());
    const size_t meta_bytes = max_cnt * (8 + 8 + 4 + 4 + 4 + 4 + 1) + 64;
    for (int j = 0; j < depth; j++) {
        if (!P.s[j]) HIP_OK(hipStreamCreateWithFlags(&P.s[j], hipStreamNonBlocking));
        HIP_OK(P.in[j].need(max_in + 64));  // the kernels' vector loads may run past the last byte
        HIP_OK(P.out[j].need(max_out + 1));
        HIP_OK(P.meta[j].need(meta_bytes));
    }
    auto slot_of = [&](int j, const HostChunk& ch) {
        HostSlot s;
        uint8_t* m = static_cast<uint8_t*>(P.meta[j].p);
        s.in_off = reinterpret_cast<uint64_t*>(m), m += max_cnt * 8;
        s.out_off = reinterpret_cast<uint64_t*>(m), m += max_cnt * 8;
        s.in_len = reinterpret_cast<uint32_t*>(m), m += max_cnt * 4;
        s.out_cap = reinterpret_cast<uint32_t*>(m), m += max_cnt * 4;
        s.out_len = reinterpret_cast<uint32_t*>(m), m += max_cnt * 4;
        s.in_consumed = reinterpret_cast<uint32_t*>(m), m += max_cnt * 4;
        s.status = reinterpret_cast<int8_t*>(m);
        s.in = static_cast<const uint8_t*>(P.in[j].p) - ch.in_lo;
        s.out = static_cast<uint8_t*>(P.out[j].p) - ch.out_lo;
        return s;
    };
    auto feed = [&](size_t k) -> int {  // copy-in + launch of chunk k
        const HostChunk& ch = chunks[k];
        const int j = (int)(k % Pipe::kDepth);
        const HostSlot s = slot_of(j, ch);
        const size_t cnt = ch.i1 - ch.i0;
        hipStream_t st = P.s[j];
        if (ch.in_hi > ch.in_lo)
            HIP_OK(hipMemcpyAsync(P.in[j].p, b.in + ch.in_lo, ch.in_hi - ch.in_lo, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(const_cast<uint64_t*>(s.in_off), b.in_off + ch.i0, cnt * 8, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(const_cast<uint64_t*>(s.out_off), b.out_off + ch.i0, cnt * 8, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(const_cast<uint32_t*>(s.in_len), b.in_len + ch.i0, cnt * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(const_cast<uint32_t*>(s.out_cap), b.out_cap + ch.i0, cnt * 4, hipMemcpyHostToDevice, st));
        return launch(s, cnt, d_dict, st);
    };
    auto drain = [&](size_t k) -> int {  // copy-out of chunk k, complete on return
        const HostChunk& ch = chunks[k];
        const int j = (int)(k % Pipe::kDepth);
        const HostSlot s = slot_of(j, ch);
        const size_t cnt = ch.i1 - ch.i0;
        hipStream_t st = P.s[j];
        // (a handful of streams -- ONE long stream decoded with room for its worst case: tamp.decompress(100 MB) offers 325 MB --
        // first learn what was produced and copy exactly that: the packed path below would move the whole extent)
        const bool out_packed = ch.out_packed && cnt > 16;
        if (out_packed && ch.out_hi > ch.out_lo)
            HIP_OK(hipMemcpyAsync(b.out + ch.out_lo, P.out[j].p, ch.out_hi - ch.out_lo, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(b.out_len + ch.i0, s.out_len, cnt * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(b.status + ch.i0, s.status, cnt, hipMemcpyDeviceToHost, st));
        if (b.in_consumed) HIP_OK(hipMemcpyAsync(b.in_consumed + ch.i0, s.in_consumed, cnt * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        // (staging pays when the extent is mostly produced bytes; a sparse or permuted batch can span gigabytes for a few
        // megabytes of output -- those, and extents above 512 MiB of pinned memory per slot, take the merged copies below)
        size_t produced = 0;
        if (!out_packed)
            for (size_t i = ch.i0; i < ch.i1; i++) produced += b.out_len[i];
        const size_t extent = ch.out_hi - ch.out_lo;
        // (... and a handful of streams with room to spare -- one long stream decoded into 8 x its compressed size -- copy their
        // produced bytes directly: the extent of tamp.decompress(100 MB) is 325 MB for 100 MB of output)
        const bool stage_ok = extent <= ((size_t)512 << 20) && extent <= 8 * produced + ((size_t)1 << 20) &&
                              (cnt > 16 || extent <= produced + produced / 4 + ((size_t)1 << 20));
        if (!out_packed && ch.out_hi > ch.out_lo && stage_ok && !getenv("TAMP_AMD_NO_STAGED_COPYBACK") &&
            P.stage[j].need(ch.out_hi - ch.out_lo) == hipSuccess) {
            // One device-to-pinned transfer of the chunk's whole extent, then exactly the produced bytes of every stream
            // placed by the host: bytes between and behind the slabs are never written.  (One hipMemcpyAsync per stream,
            // the form this replaces and the fallback below, is orders of magnitude slower for 10^5+ padded slabs.)
            uint8_t* stg = static_cast<uint8_t*>(P.stage[j].p);
            HIP_OK(hipMemcpyAsync(stg, P.out[j].p, ch.out_hi - ch.out_lo, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            for (size_t i = ch.i0; i < ch.i1; i++)
                if (b.out_len[i]) memcpy(b.out + b.out_off[i], stg + (b.out_off[i] - ch.out_lo), b.out_len[i]);
        } else if (!out_packed) {
            (void)hipGetLastError();  // (a failed pinned allocation must not poison later calls)
            // exactly the produced bytes of every stream, runs of touching full slabs merged into one transfer
            const uint8_t* dev_out = static_cast<const uint8_t*>(P.out[j].p);
            size_t i = ch.i0;
            while (i < ch.i1) {
                const uint64_t lo = b.out_off[i];
                uint64_t hi = lo + b.out_len[i];
                size_t k = i + 1;
                while (k < ch.i1 && b.out_len[k - 1] == b.out_cap[k - 1] && b.out_off[k] == hi) hi += b.out_len[k], k++;
                if (hi > lo) HIP_OK(hipMemcpyAsync(b.out + lo, dev_out + (lo - ch.out_lo), hi - lo, hipMemcpyDeviceToHost, st));
                i = k;
            }
            HIP_OK(hipStreamSynchronize(st));
        }
        return TAMP_OK;
    };
    if (chunks.size() == 1) {
        int rc = feed(0);
        return rc != TAMP_OK ? rc : drain(0);
    }
    std::mutex m;
    std::condition_variable cv;
    size_t fed = 0, drained = 0;
    int feed_rc = TAMP_OK;
    bool stop = false;
    std::string feed_msg;
    std::thread feeder([&] {
        int rc = hipSetDevice(device) == hipSuccess ? TAMP_OK : TAMP_AMD_NO_DEVICE;
        for (size_t k = 0; k < chunks.size() && rc == TAMP_OK; k++) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || k < drained + Pipe::kDepth; });  // slot k % kDepth is free again
                if (stop) return;
            }
            rc = feed(k);
            std::lock_guard<std::mutex> lk(m);
            if (rc == TAMP_OK) fed = k + 1;
            else feed_rc = rc, feed_msg = t_last_error;
            cv.notify_all();
        }
        if (rc != TAMP_OK) {
            std::lock_guard<std::mutex> lk(m);
            if (feed_rc == TAMP_OK) feed_rc = rc;
            cv.notify_all();
        }
    });
    int rc = TAMP_OK;
    for (size_t k = 0; k < chunks.size(); k++) {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return fed > k || feed_rc != TAMP_OK; });
            if (fed <= k) {
                rc = feed_rc;
                snprintf(t_last_error, sizeof t_last_error, "%s", feed_msg.c_str());
                break;
            }
        }
        rc = drain(k);
        if (rc != TAMP_OK) break;
        std::lock_guard<std::mutex> lk(m);
        drained = k + 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
        cv.notify_all();
    }
    feeder.join();
    if (rc != TAMP_OK)
        for (int j = 0; j < depth; j++) (void)hipStreamSynchronize(P.s[j]);  // nothing of this call left in flight
    return rc;
}

// tamp_compressor_init (compressor.c:191-244) on a state + a window buffer that may live apart (the reference-named
// object keeps the window in the caller's buffer)
tamp_res encoder_state_fill(TampAmdEncoderState* s, unsigned char* window, const TampAmdConf* conf, int append,
                            uint8_t window_bits_max) {
    TampAmdConf dflt;
    std::memset(&dflt, 0, sizeof dflt);
    dflt.window = 10, dflt.literal = 8, dflt.extended = 1;  // compressor.c:193-203
    if (!conf) conf = &dflt;
    if (!conf_valid(conf) || conf->window > window_bits_max) return TAMP_INVALID_CONF;
    if (append && (!conf->dictionary_reset || conf->use_custom_dictionary)) return TAMP_INVALID_CONF;  // :210
    std::memset(s, 0, sizeof *s);
    s->window = conf->window, s->literal = conf->literal;
    s->flags = (uint8_t)((conf->use_custom_dictionary ? 1 : 0) | (conf->extended ? 2 : 0) | (conf->dictionary_reset ? 4 : 0) |
                         (append ? 8 : 0) | (conf->lazy_matching ? 16 : 0));
    s->cached_match_index = -1;
    if (!conf->use_custom_dictionary)
        seed_dictionary_host(window, (size_t)1 << conf->window, conf->extended ? conf->literal : 8);
    if (append) {  // FLUSH padded to 16 bits: with the previous stream's trailing FLUSH a dictionary reset (:227-235)
        s->bit_buffer = 0xABu << 23, s->bit_buffer_pos = 16, s->last_was_flush = 1;
    } else {  // header byte (+ a zero byte when dictionary_reset), compressor.c:236-241
        const uint32_t header = ((conf->window - 8u) << 5) | ((conf->literal - 5u) << 3) |
                                ((conf->use_custom_dictionary ? 1u : 0u) << 2) | ((conf->extended ? 1u : 0u) << 1) |
                                (conf->dictionary_reset ? 1u : 0u);
        s->bit_buffer = header << 24, s->bit_buffer_pos = conf->dictionary_reset ? 16 : 8;
    }
    return TAMP_OK;
}

size_t env_or(const char* name, size_t dflt) {  // tuning knobs of the host-memory path
    const char* e = getenv(name);
    const long v = e ? atol(e) : 0;
    return v > 0 ? (size_t)v : dflt;
}

// TAMP_AMD_ALL_DEVICES with host memory: the streams are independent, so the batch is cut into one contiguous range
// per visible device, balanced by input bytes, and each range runs the host-memory path of
// its device on its own host thread.  No data moves between devices; results land in the caller's arrays directly.
int device_fanout_count() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return 0;
    const char* e = getenv("TAMP_AMD_FANOUT");  // tests: more shards than devices (they share device i % count)
    const int forced = e ? atoi(e) : 0;
    return forced > 0 ? forced : count;
}

template <class Call>  // Call(device, first_stream, count) -> library-level return code
int fan_out_over_devices(const uint32_t* in_len, size_t n_streams, const Call& call) {
    const int shards = device_fanout_count();
    int devices = 0;
    (void)hipGetDeviceCount(&devices);
    if (shards < 1 || devices < 1) {
        snprintf(t_last_error, sizeof t_last_error, "no HIP device visible");
        return TAMP_AMD_NO_DEVICE;
    }
    uint64_t total = 0;
    for (size_t i = 0; i < n_streams; i++) total += in_len[i];
    std::vector<size_t> cut(shards + 1, n_streams);
    cut[0] = 0;
    {
        uint64_t run = 0;
        size_t i = 0;
        for (int r = 1; r < shards; r++) {
            const uint64_t target = total / (uint64_t)shards * (uint64_t)r;
            while (i < n_streams && (total ? run < target : i < n_streams * (size_t)r / (size_t)shards)) run += in_len[i++];
            cut[r] = i;
        }
    }
    std::vector<int> rcs(shards, TAMP_OK);
    std::vector<std::string> msgs(shards);
    std::vector<std::thread> workers;
    for (int r = 0; r < shards; r++) {
        if (cut[r + 1] == cut[r]) continue;
        workers.emplace_back([&, r] {
            rcs[r] = call(r % devices, cut[r], cut[r + 1] - cut[r]);
            if (rcs[r] != TAMP_OK) msgs[r] = t_last_error;
        });
    }
    for (std::thread& w : workers) w.join();
    for (int r = 0; r < shards; r++)
        if (rcs[r] != TAMP_OK) {
            snprintf(t_last_error, sizeof t_last_error, "shard %d: %s", r, msgs[r].c_str());
            return rcs[r];
        }
    return TAMP_OK;
}

}  // namespace

extern "C" {

void tamp_initialize_dictionary(unsigned char* buffer, size_t size, uint8_t literal) {
    seed_dictionary_host(buffer, size, literal);
}

int8_t tamp_compute_min_pattern_size(uint8_t window, uint8_t literal) {
    return (int8_t)min_pattern_size(window, literal);
}

// common.h:424 / common.c:58-86: ring[pos..pos+n) (destination wraps) <- ring[off..off+n) (source does not), with the
// result of reading every source byte before any is overwritten.  A host-side buffer utility of the reference's ABI
// (the device decoders have their own copies of this rule); no codec work.
void tamp_window_copy(unsigned char* window, uint16_t* window_pos, uint16_t window_offset, uint8_t match_size,
                      uint16_t window_mask) {
    unsigned char tmp[256];
    for (unsigned i = 0; i < match_size; i++) tmp[i] = window[(size_t)window_offset + i];
    uint16_t p = *window_pos;
    for (unsigned i = 0; i < match_size; i++) {
        window[p] = tmp[i];
        p = (uint16_t)((p + 1) & window_mask);
    }
    *window_pos = p;
}

size_t tamp_amd_compress_bound(size_t n, uint8_t literal, int dictionary_reset) {
    return 1 + (dictionary_reset ? 1 : 0) + (n * ((size_t)literal + 1) + 7) / 8;
}

int tamp_amd_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char* tamp_amd_version(void) { return "tamp_amd 0.1 (gfx950)"; }

int tamp_amd_compress_plan(uint8_t window_bits, uint32_t max_in_len, int lazy_matching, uint32_t* block_positions,
                           uint32_t* lds_bytes, uint32_t* threads, uint32_t* workgroups_per_cu) {
    if (window_bits < 8 || window_bits > 15) return TAMP_AMD_BAD_ARGUMENT;
    // (the same decisions as launch_compress: build, block, workgroup size)
    const uint32_t W = 1u << window_bits;
    const bool packed = window_bits <= 14, lazy = lazy_matching != 0;
    const bool long_streams = max_in_len == 0 || align_up(max_in_len, 64) >= 1024;
    const bool runlist = packed && !lazy && long_streams;
    const uint32_t hb = runlist && window_bits == 10 ? kHb1024 : kHashBits;
    const uint32_t blk = pick_block(W, max_in_len, packed, lazy, runlist, hb);
    const CompressLds L(W, blk, packed, lazy, runlist, hb);
    const uint32_t reg_cap = lazy ? (uint32_t)TAMP_LAZY_PER_CU : (runlist ? (uint32_t)TAMP_WG_PER_CU : (uint32_t)TAMP_LEAN_PER_CU);
    const uint32_t by_lds = 160u * 1024u / align_up(L.total, 2048u);
    if (block_positions) *block_positions = blk;
    if (lds_bytes) *lds_bytes = L.total;
    if (threads) *threads = blk >= 1024 ? 256u : 64u;
    if (workgroups_per_cu) *workgroups_per_cu = by_lds < reg_cap ? by_lds : reg_cap;
    return TAMP_OK;
}

const char* tamp_amd_last_error(void) { return t_last_error; }

#if defined(TAMP_PROF)
// debug-only: per-phase cycle counters (not part of the public header)
int tamp_amd_prof_read(unsigned long long* out6) {
    if (!g_prof) {
        if (hipMalloc(&g_prof, 128) != hipSuccess) return -1;
        (void)hipMemset(g_prof, 0, 128);
        for (int i = 0; i < 12; i++) out6[i] = 0;
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out6, g_prof, 128, hipMemcpyDeviceToHost);
    (void)hipMemset(g_prof, 0, 128);
    return 0;
}
#endif

void* tamp_amd_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void tamp_amd_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

void tamp_amd_set_timing(int enabled) { t_timing = enabled != 0; }


// Release the scratch the library keeps between calls on `device` (decoder window slabs, split-decoder records, header
// pre-pass words: one set per HIP stream that ever decoded) and, since round 4, the staging of the host-memory pipeline
// (pinned host buffers and device chunk buffers).  Every stream that owns a slab is synchronised first.  Returns the number
// of bytes released, or a negative TAMP_AMD_* code.
long long tamp_amd_trim(int device) {
    DeviceCtx* ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc != TAMP_OK) return rc;
    long long freed = 0;
    // lock order of the decode launches: a slab's launch_mu first, g_mu inside it -- so the slabs are listed under g_mu
    // (map nodes do not move) and released one by one in that order
    std::vector<std::pair<hipStream_t, DeviceCtx::Slab*>> slabs;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (auto& kv : ctx->slabs) slabs.emplace_back(kv.first, &kv.second);
    }
    for (auto& ps : slabs) {
        DeviceCtx::Slab& slab = *ps.second;
        std::lock_guard<std::mutex> call_lock(slab.launch_mu);
        if (hipStreamSynchronize(ps.first) != hipSuccess) (void)hipGetLastError();
        std::lock_guard<std::mutex> lock(g_mu);
        if (slab.p) { (void)hipFree(slab.p); freed += (long long)slab.bytes; slab.p = nullptr, slab.bytes = 0; }
        if (slab.split) { (void)hipFree(slab.split); freed += (long long)slab.split_bytes; slab.split = nullptr, slab.split_bytes = 0; }
    }
    {   // block-mode tables and the expensive-first ordering's gathered tables (round 5)
        std::lock_guard<std::mutex> blk_lock(ctx->blk_mu);
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        for (auto& kv : ctx->blk_scratch)
            if (kv.second.p) { (void)hipFree(kv.second.p); freed += (long long)kv.second.bytes; kv.second.p = nullptr, kv.second.bytes = 0; }
        {   // expensive-first tables: a stream's enqueue lock first (a call in flight on it finishes its launches), then the map's
            std::vector<hipStream_t> streams;
            {
                std::lock_guard<std::mutex> lpt_lock(ctx->lpt_mu);
                for (auto& kv : ctx->lpt_scratch) streams.push_back(kv.first);
            }
            for (hipStream_t s2 : streams) {
                std::lock_guard<std::mutex> enqueue(ctx->enqueue_lock(s2));
                if (hipStreamSynchronize(s2) != hipSuccess) (void)hipGetLastError();
                std::lock_guard<std::mutex> lpt_lock(ctx->lpt_mu);
                auto& g = ctx->lpt_scratch[s2];
                if (g.p) { (void)hipFree(g.p); freed += (long long)g.bytes; g.p = nullptr, g.bytes = 0; }
            }
        }
        std::lock_guard<std::mutex> long_lock(ctx->long_mu);  // (the long-stream decoder's chunk tables and records)
        for (auto& kv : ctx->long_scratch)
            if (kv.second.p) { (void)hipFree(kv.second.p); freed += (long long)kv.second.bytes; kv.second.p = nullptr, kv.second.bytes = 0; }
        for (auto& kv : ctx->long_maps)
            if (kv.second.p) { (void)hipFree(kv.second.p); freed += (long long)kv.second.bytes; kv.second.p = nullptr, kv.second.bytes = 0; }
        for (auto& kv : ctx->long_spec)
            if (kv.second.p) { (void)hipFree(kv.second.p); freed += (long long)kv.second.bytes; kv.second.p = nullptr, kv.second.bytes = 0; }
    }
    {   // the host-memory pipeline: pinned staging of non-tiling output slabs (it grows with the largest extent ever staged,
        // possibly gigabytes of pinned RAM) and the device-side chunk buffers; both grow again on demand
        std::lock_guard<std::mutex> pipe_lock(ctx->pipe.mu);
        for (int i = 0; i < DeviceCtx::HostPipe::kDepth; i++) {
            if (ctx->pipe.s[i] && hipStreamSynchronize(ctx->pipe.s[i]) != hipSuccess) (void)hipGetLastError();
            auto& st = ctx->pipe.stage[i];
            if (st.p) { (void)hipHostFree(st.p); freed += (long long)st.bytes; st.p = nullptr, st.bytes = 0; }
            for (auto* g : {&ctx->pipe.in[i], &ctx->pipe.out[i], &ctx->pipe.meta[i]})
                if (g->p) { (void)hipFree(g->p); freed += (long long)g->bytes; g->p = nullptr, g->bytes = 0; }
        }
    }
    return freed;
}

float tamp_amd_last_kernel_ms(void) {
    if (!t_ev_valid) return -1.0f;
    if (hipEventSynchronize(t_ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, t_ev0, t_ev1) != hipSuccess) return -1.0f;
    return ms;
}

int tamp_batch_compress(const TampAmdConf* conf, const uint8_t* dictionary, const uint8_t* in, const uint64_t* in_off,
                        const uint32_t* in_len, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                        uint32_t* out_len, int8_t* status, size_t n_streams, uint32_t max_in_len, int mem, int device,
                        void* stream) {
    if (!conf || (n_streams && (!in_off || !in_len || !out_off || !out_cap || !out_len || !status)))
        return TAMP_AMD_BAD_ARGUMENT;
    if (n_streams > 0xFFFFFFFFull) return TAMP_AMD_BAD_ARGUMENT;
    if (mem != TAMP_AMD_MEM_HOST && mem != TAMP_AMD_MEM_DEVICE) return TAMP_AMD_BAD_ARGUMENT;
    if (device == TAMP_AMD_ALL_DEVICES) {
        if (mem != TAMP_AMD_MEM_HOST) return TAMP_AMD_BAD_ARGUMENT;  // device pointers belong to one device
        if (!max_in_len)
            for (size_t i = 0; i < n_streams; i++) max_in_len = std::max(max_in_len, in_len[i]);
        return fan_out_over_devices(in_len, n_streams, [&](int dev, size_t i0, size_t cnt) {
            return tamp_batch_compress(conf, dictionary, in, in_off + i0, in_len + i0, out, out_off + i0, out_cap + i0,
                                       out_len + i0, status + i0, cnt, max_in_len, mem, dev, nullptr);
        });
    }
    DeviceCtx* ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc != TAMP_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool bad_conf = !conf_valid(conf) || (conf->use_custom_dictionary && !dictionary);

    if (mem == TAMP_AMD_MEM_DEVICE) {
        if (bad_conf) {  // tamp_compressor_init would have returned TAMP_INVALID_CONF for every stream
            HIP_OK(hipMemsetAsync(status, (uint8_t)(int8_t)TAMP_INVALID_CONF, n_streams, st));
            HIP_OK(hipMemsetAsync(out_len, 0, n_streams * sizeof(uint32_t), st));
            return TAMP_OK;
        }
        return launch_compress(ctx, conf, dictionary, in, in_off, in_len, out, out_off, out_cap, out_len, status,
                               n_streams, max_in_len, st);
    }

    // ---- host memory: stage, run, copy back ----
    if (bad_conf) {
        for (size_t i = 0; i < n_streams; i++) status[i] = TAMP_INVALID_CONF, out_len[i] = 0;
        return TAMP_OK;
    }
    if (n_streams == 0) return TAMP_OK;
    if (!max_in_len) {
        uint32_t maxlen = 0;
        for (size_t i = 0; i < n_streams; i++) maxlen = std::max(maxlen, in_len[i]);
        max_in_len = maxlen ? maxlen : 16;
    }
    const HostBatch b = {in, in_off, in_len, out, out_off, out_cap, out_len, status, nullptr, n_streams};
    std::vector<HostChunk> chunks;
    // a chunk fills the device three times over (256 CUs x 6 workgroups = 1,536 streams at once); measured best for
    // 4 KiB streams (tools/host_path_bench.py), and at least 16 MiB so that short messages do not drown in call overhead
    plan_host_chunks(b, env_or("TAMP_AMD_HOST_CHUNK_STREAMS", (size_t)ctx->cu_count * 18),
                     (uint64_t)env_or("TAMP_AMD_HOST_CHUNK_MB", 16) << 20, 1ull << 30, chunks);
    return run_host_batch(ctx, device, b, chunks, conf->use_custom_dictionary ? dictionary : nullptr,
                          (size_t)1 << conf->window,
                          [&](const HostSlot& s, size_t count, const uint8_t* d_dict, hipStream_t cs) {
        return launch_compress(ctx, conf, d_dict, s.in, s.in_off, s.in_len, s.out, s.out_off, s.out_cap, s.out_len,
                               s.status, count, max_in_len, cs);
    });
}

int tamp_batch_decompress(const uint8_t* dictionary, size_t dictionary_len, uint8_t max_window_bits, const uint8_t* in,
                          const uint64_t* in_off, const uint32_t* in_len, uint8_t* out, const uint64_t* out_off,
                          const uint32_t* out_cap, uint32_t* out_len, int8_t* status, uint32_t* in_consumed,
                          size_t n_streams, int mem, int device, void* stream) {
    if (n_streams && (!in_off || !in_len || !out_off || !out_cap || !out_len || !status)) return TAMP_AMD_BAD_ARGUMENT;
    if (n_streams > 0xFFFFFFFFull) return TAMP_AMD_BAD_ARGUMENT;
    if (mem != TAMP_AMD_MEM_HOST && mem != TAMP_AMD_MEM_DEVICE) return TAMP_AMD_BAD_ARGUMENT;
    if (device == TAMP_AMD_ALL_DEVICES) {
        if (mem != TAMP_AMD_MEM_HOST) return TAMP_AMD_BAD_ARGUMENT;
        return fan_out_over_devices(in_len, n_streams, [&](int dev, size_t i0, size_t cnt) {
            return tamp_batch_decompress(dictionary, dictionary_len, max_window_bits, in, in_off + i0, in_len + i0, out,
                                         out_off + i0, out_cap + i0, out_len + i0, status + i0,
                                         in_consumed ? in_consumed + i0 : nullptr, cnt, mem, dev, nullptr);
        });
    }
    DeviceCtx* ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc != TAMP_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!dictionary) dictionary_len = 0;

    if (mem == TAMP_AMD_MEM_DEVICE)
        return launch_decompress(ctx, dictionary, dictionary_len, max_window_bits, in, in_off, in_len, out, out_off,
                                 out_cap, out_len, status, in_consumed, n_streams, st);

    if (n_streams == 0) return TAMP_OK;
    dictionary_len = std::min(dictionary_len, kSeedTable);
    const HostBatch b = {in, in_off, in_len, out, out_off, out_cap, out_len, status, in_consumed, n_streams};
    std::vector<HostChunk> chunks;
    // the lane-per-stream decoders want tens of thousands of streams per launch (launch_decompress)
    plan_host_chunks(b, env_or("TAMP_AMD_HOST_CHUNK_STREAMS", (size_t)ctx->cu_count * 128),
                     (uint64_t)env_or("TAMP_AMD_HOST_CHUNK_MB", 32) << 20, 1ull << 30, chunks);
    uint32_t* no_consumed = nullptr;
    return run_host_batch(ctx, device, b, chunks, dictionary, dictionary_len,
                          [&](const HostSlot& s, size_t count, const uint8_t* d_dict, hipStream_t cs) {
        return launch_decompress(ctx, d_dict, d_dict ? dictionary_len : 0, max_window_bits, s.in, s.in_off, s.in_len, s.out,
                                 s.out_off, s.out_cap, s.out_len, s.status, in_consumed ? s.in_consumed : no_consumed,
                                 count, cs);
    });
}

size_t tamp_amd_decoder_state_size(uint8_t window_bits_max) {
    return sizeof(TampAmdDecoderState) + ((size_t)1 << (window_bits_max & 15));
}

tamp_res tamp_amd_decoder_state_init(void* state, const TampAmdConf* conf, uint8_t window_bits_max) {
    if (!state) return TAMP_AMD_BAD_ARGUMENT;
    if (window_bits_max < 8 || window_bits_max > 15) return TAMP_INVALID_CONF;  // decompressor.c:336
    TampAmdDecoderState* s = static_cast<TampAmdDecoderState*>(state);
    std::memset(s, 0, sizeof *s);
    s->window_bits_max = window_bits_max;
    if (!conf) return TAMP_OK;
    // tamp_decompressor_populate_from_conf, decompressor.c:304-329
    if (!conf_valid(conf) || conf->window > window_bits_max) return TAMP_INVALID_CONF;
    if (!conf->use_custom_dictionary)
        seed_dictionary_host(reinterpret_cast<unsigned char*>(s + 1), (size_t)1 << conf->window,
                             conf->extended ? conf->literal : 8);
    s->conf = (uint8_t)(((conf->window - 8) << 5) | ((conf->literal - 5) << 3) | ((conf->use_custom_dictionary ? 1 : 0) << 2) |
                        ((conf->extended ? 1 : 0) << 1) | (conf->dictionary_reset ? 1 : 0));
    s->flags = 1;
    return TAMP_OK;
}

int tamp_batch_decompress_resume(void* states, size_t state_stride, uint8_t window_bits_max, const uint8_t* in,
                                 const uint64_t* in_off, const uint32_t* in_len, uint8_t* out, const uint64_t* out_off,
                                 const uint32_t* out_cap, uint32_t* out_len, int8_t* status, uint32_t* in_consumed,
                                 size_t n_streams, int mem, int device, void* stream) {
    if (n_streams && (!states || !in_off || !in_len || !out_off || !out_cap || !out_len || !status))
        return TAMP_AMD_BAD_ARGUMENT;
    if (n_streams > 0xFFFFFFFFull) return TAMP_AMD_BAD_ARGUMENT;
    if (mem != TAMP_AMD_MEM_HOST && mem != TAMP_AMD_MEM_DEVICE) return TAMP_AMD_BAD_ARGUMENT;
    if (window_bits_max < 8 || window_bits_max > 15 || (state_stride & 15) ||
        state_stride < tamp_amd_decoder_state_size(window_bits_max) || (reinterpret_cast<uintptr_t>(states) & 3))
        return TAMP_AMD_BAD_ARGUMENT;
    DeviceCtx* ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc != TAMP_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mem == TAMP_AMD_MEM_DEVICE)
        return launch_decompress_resume(ctx, static_cast<uint8_t*>(states), state_stride, window_bits_max, in, in_off,
                                        in_len, out, out_off, out_cap, out_len, status, in_consumed, n_streams, st);
    if (n_streams == 0) return TAMP_OK;
    uint64_t in_end = 0, out_end = 0;
    for (size_t i = 0; i < n_streams; i++) {
        in_end = std::max(in_end, in_off[i] + in_len[i]);
        out_end = std::max(out_end, out_off[i] + out_cap[i]);
    }
    DevBuf d_sta, d_in, d_out, d_io, d_il, d_oo, d_oc, d_ol, d_st, d_ic;
    HIP_OK(d_sta.alloc(n_streams * state_stride));
    HIP_OK(d_in.alloc(in_end + 64));
    HIP_OK(d_out.alloc(out_end));
    HIP_OK(d_io.alloc(n_streams * 8));
    HIP_OK(d_il.alloc(n_streams * 4));
    HIP_OK(d_oo.alloc(n_streams * 8));
    HIP_OK(d_oc.alloc(n_streams * 4));
    HIP_OK(d_ol.alloc(n_streams * 4));
    HIP_OK(d_ic.alloc(n_streams * 4));
    HIP_OK(d_st.alloc(n_streams));
    HIP_OK(hipMemcpyAsync(d_sta.p, states, n_streams * state_stride, hipMemcpyHostToDevice, st));
    if (in_end) HIP_OK(hipMemcpyAsync(d_in.p, in, in_end, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_io.p, in_off, n_streams * 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_il.p, in_len, n_streams * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_oo.p, out_off, n_streams * 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_oc.p, out_cap, n_streams * 4, hipMemcpyHostToDevice, st));
    rc = launch_decompress_resume(ctx, d_sta.as<uint8_t>(), state_stride, window_bits_max, d_in.as<uint8_t>(),
                                  d_io.as<uint64_t>(), d_il.as<uint32_t>(), d_out.as<uint8_t>(), d_oo.as<uint64_t>(),
                                  d_oc.as<uint32_t>(), d_ol.as<uint32_t>(), d_st.as<int8_t>(), d_ic.as<uint32_t>(),
                                  n_streams, st);
    if (rc != TAMP_OK) return rc;
    HIP_OK(hipMemcpyAsync(states, d_sta.p, n_streams * state_stride, hipMemcpyDeviceToHost, st));
    // only what was written goes back: a caller's output buffer is not touched beyond out_len
    HIP_OK(hipMemcpyAsync(out_len, d_ol.p, n_streams * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(status, d_st.p, n_streams, hipMemcpyDeviceToHost, st));
    if (in_consumed) HIP_OK(hipMemcpyAsync(in_consumed, d_ic.p, n_streams * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (size_t i = 0; i < n_streams; i++)
        if (out_len[i])
            HIP_OK(hipMemcpyAsync(out + out_off[i], d_out.as<uint8_t>() + out_off[i], out_len[i], hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return TAMP_OK;
}

size_t tamp_amd_encoder_state_size(uint8_t window_bits_max) {
    return sizeof(TampAmdEncoderState) + ((size_t)1 << (window_bits_max & 15));
}

tamp_res tamp_amd_encoder_state_init(void* state, const TampAmdConf* conf, int append, uint8_t window_bits_max) {
    if (!state) return TAMP_AMD_BAD_ARGUMENT;
    if (window_bits_max > 15) return TAMP_INVALID_CONF;
    TampAmdEncoderState* s = static_cast<TampAmdEncoderState*>(state);
    return encoder_state_fill(s, reinterpret_cast<unsigned char*>(s + 1), conf, append, window_bits_max);
}

int tamp_batch_compress_resume(void* states, size_t state_stride, uint8_t window_bits_max, int op, int write_token,
                               const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint8_t* out,
                               const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, int8_t* status,
                               uint32_t* in_consumed, size_t n_objects, int mem, int device, void* stream) {
    if (n_objects && (!states || !in_off || !in_len || !out_off || !out_cap || !out_len || !status))
        return TAMP_AMD_BAD_ARGUMENT;
    if (n_objects > 0xFFFFFFFFull || op < TAMP_AMD_OP_POLL || op > TAMP_AMD_OP_COMPRESS_AND_FLUSH) return TAMP_AMD_BAD_ARGUMENT;
    if (mem != TAMP_AMD_MEM_HOST && mem != TAMP_AMD_MEM_DEVICE) return TAMP_AMD_BAD_ARGUMENT;
    if (window_bits_max < 8 || window_bits_max > 15 || (state_stride & 15) ||
        state_stride < tamp_amd_encoder_state_size(window_bits_max) || (reinterpret_cast<uintptr_t>(states) & 3))
        return TAMP_AMD_BAD_ARGUMENT;
    DeviceCtx* ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc != TAMP_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mem == TAMP_AMD_MEM_DEVICE)
        return launch_compress_resume(ctx, static_cast<uint8_t*>(states), state_stride, window_bits_max, op, write_token,
                                      in, in_off, in_len, out, out_off, out_cap, out_len, status, in_consumed, n_objects,
                                      st);
    if (n_objects == 0) return TAMP_OK;
    uint64_t in_end = 0, out_end = 0;
    for (size_t i = 0; i < n_objects; i++) {
        in_end = std::max(in_end, in_off[i] + in_len[i]);
        out_end = std::max(out_end, out_off[i] + out_cap[i]);
    }
    DevBuf d_sta, d_in, d_out, d_io, d_il, d_oo, d_oc, d_ol, d_st, d_ic;
    HIP_OK(d_sta.alloc(n_objects * state_stride));
    HIP_OK(d_in.alloc(in_end + 64));
    HIP_OK(d_out.alloc(out_end));
    HIP_OK(d_io.alloc(n_objects * 8));
    HIP_OK(d_il.alloc(n_objects * 4));
    HIP_OK(d_oo.alloc(n_objects * 8));
    HIP_OK(d_oc.alloc(n_objects * 4));
    HIP_OK(d_ol.alloc(n_objects * 4));
    HIP_OK(d_ic.alloc(n_objects * 4));
    HIP_OK(d_st.alloc(n_objects));
    HIP_OK(hipMemcpyAsync(d_sta.p, states, n_objects * state_stride, hipMemcpyHostToDevice, st));
    if (in_end) HIP_OK(hipMemcpyAsync(d_in.p, in, in_end, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_io.p, in_off, n_objects * 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_il.p, in_len, n_objects * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_oo.p, out_off, n_objects * 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_oc.p, out_cap, n_objects * 4, hipMemcpyHostToDevice, st));
    rc = launch_compress_resume(ctx, d_sta.as<uint8_t>(), state_stride, window_bits_max, op, write_token,
                                d_in.as<uint8_t>(), d_io.as<uint64_t>(), d_il.as<uint32_t>(), d_out.as<uint8_t>(),
                                d_oo.as<uint64_t>(), d_oc.as<uint32_t>(), d_ol.as<uint32_t>(), d_st.as<int8_t>(),
                                d_ic.as<uint32_t>(), n_objects, st);
    if (rc != TAMP_OK) return rc;
    HIP_OK(hipMemcpyAsync(states, d_sta.p, n_objects * state_stride, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(out_len, d_ol.p, n_objects * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(status, d_st.p, n_objects, hipMemcpyDeviceToHost, st));
    if (in_consumed) HIP_OK(hipMemcpyAsync(in_consumed, d_ic.p, n_objects * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (size_t i = 0; i < n_objects; i++)  // only what was written goes back
        if (out_len[i])
            HIP_OK(hipMemcpyAsync(out + out_off[i], d_out.as<uint8_t>() + out_off[i], out_len[i], hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return TAMP_OK;
}

tamp_res tamp_amd_compress(const TampAmdConf* conf, const unsigned char* dictionary, unsigned char* output,
                           size_t output_size, size_t* output_written_size, const unsigned char* input,
                           size_t input_size, int device) {
    if (output_written_size) *output_written_size = 0;
    if (input_size > 0xFFFFFFFFull) return TAMP_AMD_BAD_ARGUMENT;
    const uint64_t zero = 0;
    const uint32_t ilen = (uint32_t)input_size;
    const uint32_t ocap = (uint32_t)(output_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : output_size);
    uint32_t olen = 0;
    int8_t st = TAMP_ERROR;
    static const unsigned char empty = 0;
    int rc = tamp_batch_compress(conf, dictionary, input ? input : &empty, &zero, &ilen, output, &zero, &ocap, &olen,
                                 &st, 1, ilen, TAMP_AMD_MEM_HOST, device, nullptr);
    if (rc != TAMP_OK) return (tamp_res)rc;
    if (output_written_size) *output_written_size = olen;
    return st;
}

tamp_res tamp_amd_decompress(const unsigned char* dictionary, size_t dictionary_len, unsigned char* output,
                             size_t output_size, size_t* output_written_size, const unsigned char* input,
                             size_t input_size, size_t* input_consumed_size, int device) {
    if (output_written_size) *output_written_size = 0;
    if (input_consumed_size) *input_consumed_size = 0;
    if (input_size > 0xFFFFFFFFull) return TAMP_AMD_BAD_ARGUMENT;
    const uint64_t zero = 0;
    const uint32_t ilen = (uint32_t)input_size;
    const uint32_t ocap = (uint32_t)(output_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : output_size);
    uint32_t olen = 0, consumed = 0;
    int8_t st = TAMP_ERROR;
    static const unsigned char empty = 0;
    int rc = tamp_batch_decompress(dictionary, dictionary_len, 15, input ? input : &empty, &zero, &ilen, output, &zero,
                                   &ocap, &olen, &st, &consumed, 1, TAMP_AMD_MEM_HOST, device, nullptr);
    if (rc != TAMP_OK) return (tamp_res)rc;
    if (output_written_size) *output_written_size = olen;
    if (input_consumed_size) *input_consumed_size = consumed;
    return st;
}

tamp_res tamp_amd_read_header(TampAmdConf* conf, const unsigned char* input, size_t input_size,
                              size_t* input_consumed_size) {
    // decompressor.c:276-297
    if (input_consumed_size) *input_consumed_size = 0;
    if (input_size == 0) return TAMP_INPUT_EXHAUSTED;
    const size_t hs = 1 + (input[0] & 1);
    if (input_size < hs) return TAMP_INPUT_EXHAUSTED;
    if (hs >= 2 && input[1]) return TAMP_INVALID_CONF;
    std::memset(conf, 0, sizeof(*conf));
    conf->window = (uint8_t)(((input[0] >> 5) & 7) + 8);
    conf->literal = (uint8_t)(((input[0] >> 3) & 3) + 5);
    conf->use_custom_dictionary = (input[0] >> 2) & 1;
    conf->extended = (input[0] >> 1) & 1;
    conf->dictionary_reset = input[0] & 1;
    if (input_consumed_size) *input_consumed_size = hs;
    return TAMP_OK;
}


// ---------------------------------------------------------------------------------------------
// The reference's own symbol names for the one-shot path (include/tamp_compat.h)
// ---------------------------------------------------------------------------------------------
namespace {
// TampCompressor::private_ (40 bytes) holds a TampAmdEncoderState, TampDecompressor::private_ (16 bytes) a
// TampAmdDecoderState: the same fields the reference keeps in those objects; the windows are the callers' buffers.
static_assert(sizeof(TampAmdEncoderState) == 40, "encoder state layout (tamp_compress_resume_kernel.hpp reads it as 10 dwords)");
static_assert(sizeof(TampConf) == 2, "TampConf must match the reference (common.h:170-182)");
static_assert(sizeof(TampCompressor) == 48, "TampCompressor must match the reference (compressor.h:13-66)");
static_assert(sizeof(TampDecompressor) == 24, "TampDecompressor must match the reference (decompressor.h:13-57)");
static_assert(sizeof(TampAmdDecoderState) == 16, "private state must fit");
int compat_device() {
    const char* e = getenv("TAMP_AMD_DEVICE");
    return e ? atoi(e) : 0;
}
inline TampAmdEncoderState* enc_state(TampCompressor* c) { return reinterpret_cast<TampAmdEncoderState*>(c->private_); }
inline bool enc_ready(const TampAmdEncoderState* s) {
    return s->window >= 8 && s->window <= 15 && s->literal >= 5 && s->literal <= 8;
}

// One of the reference's calls on one object, exact at any granularity: the resumable device kernel
// (tamp_compress_resume_kernel.hpp) on [state | window], both written back.
tamp_res compat_encoder_call(TampCompressor* compressor, int op, bool write_token, unsigned char* output,
                             size_t output_size, size_t* output_written_size, const unsigned char* input,
                             size_t input_size, size_t* input_consumed_size) {
    if (output_written_size) *output_written_size = 0;
    if (input_consumed_size) *input_consumed_size = 0;
    TampAmdEncoderState* s = enc_state(compressor);
    if (!enc_ready(s) || !compressor->window) return TAMP_ERROR;  // not initialised
    const size_t W = (size_t)1 << s->window;
    const size_t stride = (sizeof *s + W + 15) & ~(size_t)15;
    std::vector<unsigned char> slot(stride);
    std::memcpy(slot.data(), s, sizeof *s);
    std::memcpy(slot.data() + sizeof *s, compressor->window, W);
    const uint64_t zero = 0;
    static unsigned char empty = 0;
    const uint32_t ilen = (uint32_t)std::min<size_t>(input_size, 0xFFFFFFFFu);
    const uint32_t ocap = (uint32_t)std::min<size_t>(output_size, 0xFFFFFFFFu);
    uint32_t olen = 0, icons = 0;
    int8_t st = TAMP_ERROR;
    const int rc = tamp_batch_compress_resume(slot.data(), stride, s->window, op, write_token, ilen ? input : &empty, &zero,
                                              &ilen, ocap ? output : &empty, &zero, &ocap, &olen, &st, &icons, 1,
                                              TAMP_AMD_MEM_HOST, compat_device(), nullptr);
    if (rc != TAMP_OK) return (tamp_res)rc;
    std::memcpy(s, slot.data(), sizeof *s);
    std::memcpy(compressor->window, slot.data() + sizeof *s, W);
    if (output_written_size) *output_written_size = olen;
    if (input_consumed_size) *input_consumed_size = icons;
    return st;
}

// A whole segment on an object that is between segments (ring empty, nothing pending, output bits byte aligned) with
// ample output room: the batch kernel's segment mode instead of token-by-token parsing.  Same bytes, same state after.
constexpr size_t kCompatPiece = (size_t)1 << 30;  // input bytes one object-level device call looks at
bool compat_segment_applies(const TampAmdEncoderState* s, size_t input_size, size_t output_size) {
    if (input_size < 2048 || input_size > 0xFFFFFFFFull) return false;
    if (s->input_size || s->rle_count || s->extended_match_count || s->cached_match_index >= 0) return false;
    if (s->bit_buffer_pos & 7) return false;
    return output_size >= (size_t)(s->bit_buffer_pos >> 3) + tamp_amd_compress_bound(input_size, s->literal, 0) + 2;
}

tamp_res compat_segment(TampCompressor* compressor, unsigned char* output, size_t output_size,
                        size_t* output_written_size, const unsigned char* input, size_t input_size, bool write_token) {
    TampAmdEncoderState* s = enc_state(compressor);
    const size_t lead = s->bit_buffer_pos >> 3;  // header / append marker still waiting in the bit buffer
    for (size_t k = 0; k < lead; k++) output[k] = (unsigned char)(s->bit_buffer >> (24 - 8 * k));
    TampAmdConf c;
    std::memset(&c, 0, sizeof c);
    c.window = s->window, c.literal = s->literal, c.extended = (s->flags >> 1) & 1;
    c.use_custom_dictionary = s->flags & 1, c.dictionary_reset = (s->flags >> 2) & 1, c.lazy_matching = (s->flags >> 4) & 1;
    size_t written = 0;
    int token = 0;
    uint16_t wp = s->window_pos;
    // data is being processed: the FLUSH at the end of this segment is not "consecutive" (compressor.c:548,784)
    tamp_res r = tamp_amd_compress_segment(&c, 0, 0, 1, write_token, compressor->window, &wp, output + lead,
                                           output_size - lead, &written, input, input_size, &token, compat_device());
    if (r != TAMP_OK) return r;
    s->bit_buffer = 0, s->bit_buffer_pos = 0, s->window_pos = wp, s->last_was_flush = token ? 1 : 0;
    if (output_written_size) *output_written_size = lead + written;
    return TAMP_OK;
}
}  // namespace

namespace {
// tamp_compressor_compress on an object, large input, ample output room: ONE piece for the batch kernel
// (tamp_amd_compress_piece, finish = 0) instead of token-by-token parsing by one wavefront -- the same stream and the
// same object state as far as any later call can tell, except that every whole output byte leaves now (the reference
// would hold the last token's bits back until the next poll: `written` may run up to four bytes ahead of it).
constexpr size_t kCompatPieceMin = 64 << 10;
bool compat_piece_applies(const TampAmdEncoderState* s, size_t input_size, size_t output_size) {
    if (input_size < kCompatPieceMin || input_size > 0xFFFFFF00ull) return false;
    if ((s->flags >> 4) & 1) return false;  // lazy matching: the cached match is not carried
    if (s->cached_match_index >= 0 || s->input_size > 16 || s->bit_buffer_pos > 31) return false;
    return output_size >= tamp_amd_compress_bound(input_size + 300, s->literal, 0) + 8;
}

tamp_res compat_piece(TampCompressor* compressor, unsigned char* output, size_t output_size, size_t* written,
                      const unsigned char* input, size_t input_size) {
    TampAmdEncoderState* s = enc_state(compressor);
    TampAmdConf c;
    std::memset(&c, 0, sizeof c);
    c.window = s->window, c.literal = s->literal, c.extended = (s->flags >> 1) & 1;
    c.use_custom_dictionary = s->flags & 1, c.dictionary_reset = (s->flags >> 2) & 1;
    TampAmdCarry carry;
    std::memset(&carry, 0, sizeof carry);
    carry.rle_count = s->rle_count, carry.ext_count = s->extended_match_count, carry.ext_pos = s->extended_match_position;
    carry.bit_count = s->bit_buffer_pos, carry.bits = s->bit_buffer;
    carry.tail_len = s->input_size;
    for (uint32_t k = 0; k < s->input_size; k++) carry.tail[k] = s->input[(s->input_pos + k) & 15];
    uint16_t wp = s->window_pos;
    int token = 0;
    // (resume = 1: the object's window buffer IS the state -- seeded or custom at init, carried since)
    tamp_res r = tamp_amd_compress_piece(&c, 0, 0, 1, 0, 0, compressor->window, &wp, &carry, output, output_size, written,
                                         input, input_size, &token, compat_device());
    if (r != TAMP_OK) return r;
    s->window_pos = wp;
    s->rle_count = carry.rle_count, s->extended_match_count = carry.ext_count, s->extended_match_position = carry.ext_pos;
    s->bit_buffer_pos = carry.bit_count, s->bit_buffer = carry.bit_count ? carry.bits : 0u;
    s->input_pos = 0, s->input_size = carry.tail_len;
    for (uint32_t k = 0; k < 16; k++) s->input[k] = k < carry.tail_len ? carry.tail[k] : 0;
    s->last_was_flush = 0;  // compressor.c:548
    return TAMP_OK;
}
}  // namespace

tamp_res tamp_compressor_init(TampCompressor* compressor, const TampConf* conf, unsigned char* window) {
    TampAmdConf c;
    std::memset(&c, 0, sizeof c);
    c.window = 10, c.literal = 8, c.extended = 1;  // compressor.c:193-203
    int append = 0;
    if (conf) {
        c.window = conf->window, c.literal = conf->literal, c.extended = conf->extended;
        c.use_custom_dictionary = conf->use_custom_dictionary, c.dictionary_reset = conf->dictionary_reset;
        c.lazy_matching = conf->lazy_matching;
        append = conf->append;
    }
    TampAmdEncoderState fresh;
    unsigned char* seed_into = window;
    const tamp_res r = encoder_state_fill(&fresh, seed_into, &c, append, 15);
    if (r != TAMP_OK) return r;  // compressor.c:208-213: nothing touched on an invalid conf
    std::memset(compressor, 0, sizeof *compressor);
    compressor->window = window;
    std::memcpy(compressor->private_, &fresh, sizeof fresh);
    return TAMP_OK;
}

// compressor.c:665-679: bytes into the 16-byte ring.  A buffer copy on the host; no codec work.
void tamp_compressor_sink(TampCompressor* compressor, const unsigned char* input, size_t input_size,
                          size_t* consumed_size) {
    TampAmdEncoderState* s = enc_state(compressor);
    size_t taken = 0;
    while (taken < input_size && s->input_size < 16) {
        s->input[(s->input_pos + s->input_size) & 15] = input[taken++];
        s->input_size++;
    }
    if (consumed_size) *consumed_size = taken;
}

bool tamp_compressor_full(const TampCompressor* compressor) {
    return reinterpret_cast<const TampAmdEncoderState*>(compressor->private_)->input_size == 16;
}

tamp_res tamp_compressor_poll(TampCompressor* compressor, unsigned char* output, size_t output_size,
                              size_t* output_written_size) {
    return compat_encoder_call(compressor, TAMP_AMD_OP_POLL, false, output, output_size, output_written_size, nullptr, 0,
                               nullptr);
}

tamp_res tamp_compressor_compress_cb(TampCompressor* compressor, unsigned char* output, size_t output_size,
                                     size_t* output_written_size, const unsigned char* input, size_t input_size,
                                     size_t* input_consumed_size, tamp_callback_t callback, void* user_data) {
    // One device call takes up to kCompatPiece bytes (32-bit lengths on the device side); longer inputs go piece by
    // piece on the same object -- the reference's own loop does nothing else (compressor.c:700-719) -- and the progress
    // callback (common.h:184-210: (input consumed, total input)) fires after every piece.
    size_t consumed = 0, written = 0;
    tamp_res r = TAMP_OK;
    if (output_written_size) *output_written_size = 0;
    if (input_consumed_size) *input_consumed_size = 0;
    if (!enc_ready(enc_state(compressor)) || !compressor->window) return TAMP_ERROR;
    // with a progress callback the pieces are smaller: it fires -- and may abort -- after each of them
    const size_t piece_max = callback ? std::min<size_t>(kCompatPiece, env_or("TAMP_AMD_PROGRESS_PIECE_MB", 16) << 20) : kCompatPiece;
    do {
        const size_t piece = std::min(input_size - consumed, piece_max);
        size_t c = 0, w = 0;
        if (compat_piece_applies(enc_state(compressor), piece, output_size - written)) {
            r = compat_piece(compressor, output + written, output_size - written, &w, input + consumed, piece);
            if (r == TAMP_OK) c = piece;  // (every byte is taken: parsed, or waiting in the ring)
        } else {
            r = compat_encoder_call(compressor, TAMP_AMD_OP_COMPRESS, false, output + written, output_size - written, &w,
                                    input + consumed, piece, &c);
        }
        consumed += c, written += w;
        if (r == TAMP_OK && callback) {
            int cb = callback(user_data, consumed, input_size);
            if (cb) r = (tamp_res)cb;
        }
        if (c < piece) break;  // output room ran out (TAMP_OUTPUT_FULL) or an error: the caller sees how far it got
    } while (r == TAMP_OK && consumed < input_size);
    if (input_consumed_size) *input_consumed_size = consumed;
    if (output_written_size) *output_written_size = written;
    return r;
}

tamp_res tamp_compressor_compress(TampCompressor* compressor, unsigned char* output, size_t output_size,
                                  size_t* output_written_size, const unsigned char* input, size_t input_size,
                                  size_t* input_consumed_size) {
    return tamp_compressor_compress_cb(compressor, output, output_size, output_written_size, input, input_size,
                                       input_consumed_size, nullptr, nullptr);
}

tamp_res tamp_compressor_compress_and_flush_cb(TampCompressor* compressor, unsigned char* output, size_t output_size,
                                               size_t* output_written_size, const unsigned char* input,
                                               size_t input_size, size_t* input_consumed_size, bool write_token,
                                               tamp_callback_t callback, void* user_data) {
    if (output_written_size) *output_written_size = 0;
    if (input_consumed_size) *input_consumed_size = 0;
    TampAmdEncoderState* s = enc_state(compressor);
    if (!enc_ready(s) || !compressor->window) return TAMP_ERROR;
    tamp_res r;
    if (compat_segment_applies(s, input_size, output_size)) {
        r = compat_segment(compressor, output, output_size, output_written_size, input, input_size, write_token);
        if (r == TAMP_OK && input_consumed_size) *input_consumed_size = input_size;
    } else if (input_size > kCompatPiece) {  // compressor.c:815-845 as it is written there: compress, then flush
        size_t consumed = 0, written = 0, w2 = 0;
        // (the caller's callback rides along: it sees the progress of every piece and a non-zero return aborts the call
        // there, as in the reference, where it runs per poll)
        r = tamp_compressor_compress_cb(compressor, output, output_size, &written, input, input_size, &consumed, callback, user_data);
        if (r == TAMP_OK && consumed == input_size) {
            r = tamp_compressor_flush(compressor, output + written, output_size - written, &w2, write_token);
            written += w2;
        } else if (r == TAMP_OK) {
            r = TAMP_OUTPUT_FULL;
        }
        if (input_consumed_size) *input_consumed_size = consumed;
        if (output_written_size) *output_written_size = written;
    } else {
        r = compat_encoder_call(compressor, TAMP_AMD_OP_COMPRESS_AND_FLUSH, write_token, output, output_size,
                                output_written_size, input, input_size, input_consumed_size);
    }
    if (r == TAMP_OK && callback) {  // final "100 %" callback, compressor.c:836-842
        int cb = callback(user_data, input_size, input_size);
        if (cb) return (tamp_res)cb;
    }
    return r;
}

tamp_res tamp_compressor_flush(TampCompressor* compressor, unsigned char* output, size_t output_size,
                               size_t* output_written_size, bool write_token) {
    return compat_encoder_call(compressor, TAMP_AMD_OP_FLUSH, write_token, output, output_size, output_written_size,
                               nullptr, 0, nullptr);
}

tamp_res tamp_compressor_reset_dictionary(TampCompressor* compressor, unsigned char* output, size_t output_size,
                                          size_t* output_written_size) {  // compressor.c:845-881
    if (output_written_size) *output_written_size = 0;
    TampAmdEncoderState* s = enc_state(compressor);
    if (!enc_ready(s) || !compressor->window) return TAMP_ERROR;
    if (!(s->flags & 4)) return TAMP_INVALID_CONF;
    for (int i = 0; i < 2; i++) {  // two FLUSH tokens in a row on purpose: the suppression flag is cleared before each
        size_t w = 0;
        s->last_was_flush = 0;
        tamp_res r = tamp_compressor_flush(compressor, output, output_size, &w, true);
        if (output_written_size) *output_written_size += w;
        if (r != TAMP_OK) return r;
        output += w, output_size -= w;
    }
    // re-initialise with the same conf minus the custom dictionary; the header that init writes is discarded
    TampAmdConf c;
    std::memset(&c, 0, sizeof c);
    c.window = s->window, c.literal = s->literal, c.extended = (s->flags >> 1) & 1, c.dictionary_reset = 1;
    c.lazy_matching = (s->flags >> 4) & 1;
    const int append = (s->flags >> 3) & 1;
    const tamp_res r = encoder_state_fill(s, compressor->window, &c, append, 15);
    s->bit_buffer = 0, s->bit_buffer_pos = 0;
    return r;
}

tamp_res tamp_compress_stream(TampCompressor* compressor, tamp_read_t read_cb, void* read_handle,
                              tamp_write_t write_cb, void* write_handle, size_t* input_consumed_size,
                              size_t* output_written_size, tamp_callback_t callback, void* user_data) {
    // compressor.c:891-955: pull, compress, push, until EOF; then flush(write_token=false).  The reference pumps a
    // 32-byte work buffer; here the pull fills a host buffer of at most kStreamBuffer bytes (TAMP_AMD_STREAM_BUFFER_MB,
    // default 64 MiB).  An input that ends inside the first fill is ONE segment for the batch kernel (the fast path:
    // same bytes, tamp_compressor_compress_and_flush).  A longer one is fed buffer by buffer to
    // tamp_compressor_compress on the same object -- exact at any cut, memory bounded -- and flushed at EOF.
    if (input_consumed_size) *input_consumed_size = 0;
    if (output_written_size) *output_written_size = 0;
    TampAmdEncoderState* s = enc_state(compressor);
    if (!enc_ready(s) || !compressor->window) return TAMP_ERROR;
    // (TAMP_AMD_STREAM_BUFFER_BYTES, when set, names the buffer in bytes: measurements at the reference's own 32-byte pump size)
    const size_t kStreamBuffer = env_or("TAMP_AMD_STREAM_BUFFER_BYTES", 0) >= 16 ? env_or("TAMP_AMD_STREAM_BUFFER_BYTES", 0)
                                                                                  : env_or("TAMP_AMD_STREAM_BUFFER_MB", 64) << 20;
    constexpr size_t kChunk = 1 << 16;
    std::vector<unsigned char> in, out;
    size_t total_in = 0, total_out = 0;
    auto push = [&](size_t n) -> tamp_res {
        for (size_t at = 0; at < n;) {
            const size_t k = std::min(n - at, kChunk);
            int w = write_cb(write_handle, out.data() + at, k);
            if (w < 0 || (size_t)w != k) return TAMP_WRITE_ERROR;
            at += k, total_out += k;
            if (output_written_size) *output_written_size = total_out;
        }
        return TAMP_OK;
    };
    bool eof = false, first = true;
    for (;;) {
        in.clear();
        while (!eof && in.size() < kStreamBuffer) {
            const size_t at = in.size();
            in.resize(at + kChunk);
            int got = read_cb(read_handle, in.data() + at, kChunk);
            if (got < 0) return TAMP_READ_ERROR;
            in.resize(at + (size_t)got);
            if (got == 0) eof = true;
        }
        total_in += in.size();
        if (input_consumed_size) *input_consumed_size = total_in;
        if (callback && !in.empty()) {  // once per read chunk, total unknown (common.h:198-200)
            int cb = callback(user_data, total_in, 0);
            if (cb) return (tamp_res)cb;
        }
        out.resize(tamp_amd_compress_bound(in.size() + 320, s->literal, 1) + 64);  // (room that lets a buffer go as ONE piece)
        size_t written = 0, consumed = 0;
        if (eof) {  // last (or only) buffer: compress + flush
            tamp_res r = first ? tamp_compressor_compress_and_flush_cb(compressor, out.data(), out.size(), &written, in.data(),
                                                                       in.size(), &consumed, false, nullptr, nullptr)
                               : tamp_compressor_compress(compressor, out.data(), out.size(), &written, in.data(), in.size(), &consumed);
            if (r != TAMP_OK) return r;
            if (consumed != in.size()) return TAMP_ERROR;
            if ((r = push(written)) != TAMP_OK) return r;
            if (!first) {
                r = tamp_compressor_flush(compressor, out.data(), out.size(), &written, false);
                if (r != TAMP_OK) return r;
                if ((r = push(written)) != TAMP_OK) return r;
            }
            return TAMP_OK;
        }
        tamp_res r = tamp_compressor_compress(compressor, out.data(), out.size(), &written, in.data(), in.size(), &consumed);
        if (r != TAMP_OK) return r;
        if (consumed != in.size()) return TAMP_ERROR;
        if ((r = push(written)) != TAMP_OK) return r;
        first = false;
    }
}

tamp_res tamp_decompress_stream(TampDecompressor* decompressor, tamp_read_t read_cb, void* read_handle,
                                tamp_write_t write_cb, void* write_handle, size_t* input_consumed_size,
                                size_t* output_written_size, tamp_callback_t callback, void* user_data) {
    // decompressor.c:585-640: pull a chunk, decode as far as it goes, push, repeat -- the same loop, with work buffers
    // sized for a device call instead of a microcontroller stack.  Memory stays bounded whatever the stream expands to.
    size_t consumed_proxy, written_proxy;
    if (!input_consumed_size) input_consumed_size = &consumed_proxy;
    if (!output_written_size) output_written_size = &written_proxy;
    *input_consumed_size = 0, *output_written_size = 0;
    constexpr size_t kIn = (size_t)1 << 20, kOut = (size_t)8 << 20;
    std::vector<unsigned char> in(kIn), out(kOut);
    size_t pos = 0, avail = 0;
    bool eof = false;
    for (;;) {
        if (avail == 0 && !eof) {
            const int got = read_cb(read_handle, in.data(), (int)std::min<size_t>(kIn, INT_MAX));
            if (got < 0) return TAMP_READ_ERROR;
            eof = got == 0;
            pos = 0, avail = (size_t)got;
            *input_consumed_size += (size_t)got;
        }
        size_t chunk_consumed = 0, chunk_written = 0;
        const tamp_res res = tamp_decompressor_decompress_cb(decompressor, out.data(), kOut, &chunk_written,
                                                             in.data() + pos, avail, &chunk_consumed, nullptr, nullptr);
        if (res < TAMP_OK) return res;
        pos += chunk_consumed, avail -= chunk_consumed;
        for (size_t at = 0; at < chunk_written;) {
            const size_t n = std::min(chunk_written - at, (size_t)1 << 30);
            const int w = write_cb(write_handle, out.data() + at, n);
            if (w < 0 || (size_t)w != n) return TAMP_WRITE_ERROR;
            at += n;
        }
        *output_written_size += chunk_written;
        if (res == TAMP_INPUT_EXHAUSTED && eof) break;
        if (callback) {
            const int cb = callback(user_data, *input_consumed_size, 0);
            if (cb) return (tamp_res)cb;
        }
    }
    return TAMP_OK;
}

// Built-in I/O handlers (common.c:92-132): plain host adaptors, no codec work.
int tamp_stream_mem_read(void* handle, unsigned char* buffer, size_t size) {
    TampMemReader* r = static_cast<TampMemReader*>(handle);
    const size_t n = std::min(std::min(size, r->size - r->pos), (size_t)INT_MAX);
    std::memcpy(buffer, r->data + r->pos, n);
    r->pos += n;
    return (int)n;
}

int tamp_stream_mem_write(void* handle, const unsigned char* buffer, size_t size) {
    TampMemWriter* w = static_cast<TampMemWriter*>(handle);
    if (size > w->capacity - w->pos || size > (size_t)INT_MAX) return -1;
    std::memcpy(w->data + w->pos, buffer, size);
    w->pos += size;
    return (int)size;
}

int tamp_stream_stdio_read(void* handle, unsigned char* buffer, size_t size) {
    FILE* f = static_cast<FILE*>(handle);
    const size_t n = fread(buffer, 1, std::min(size, (size_t)INT_MAX), f);
    return n == 0 && ferror(f) ? -1 : (int)n;
}

int tamp_stream_stdio_write(void* handle, const unsigned char* buffer, size_t size) {
    FILE* f = static_cast<FILE*>(handle);
    const size_t n = fwrite(buffer, 1, size, f);
    return n < size && ferror(f) ? -1 : (int)n;
}

tamp_res tamp_compressor_compress_and_flush(TampCompressor* compressor, unsigned char* output, size_t output_size,
                                            size_t* output_written_size, const unsigned char* input, size_t input_size,
                                            size_t* input_consumed_size, bool write_token) {
    return tamp_compressor_compress_and_flush_cb(compressor, output, output_size, output_written_size, input,
                                                 input_size, input_consumed_size, write_token, nullptr, nullptr);
}

tamp_res tamp_decompressor_read_header(TampConf* conf, const unsigned char* input, size_t input_size,
                                       size_t* input_consumed_size) {
    TampAmdConf c;
    tamp_res r = tamp_amd_read_header(&c, input, input_size, input_consumed_size);
    if (r != TAMP_OK) return r;
    conf->window = c.window, conf->literal = c.literal, conf->use_custom_dictionary = c.use_custom_dictionary;
    conf->extended = c.extended, conf->dictionary_reset = c.dictionary_reset;
    return TAMP_OK;
}

tamp_res tamp_decompressor_init(TampDecompressor* decompressor, const TampConf* conf, unsigned char* window,
                                uint8_t window_bits) {
    if (window_bits < 8 || window_bits > 15) return TAMP_INVALID_CONF;  // decompressor.c:336
    std::memset(decompressor, 0, sizeof *decompressor);
    decompressor->window = window;
    TampAmdDecoderState* s = reinterpret_cast<TampAmdDecoderState*>(decompressor->private_);
    s->window_bits_max = window_bits;
    if (!conf) return TAMP_OK;
    // tamp_decompressor_populate_from_conf, decompressor.c:304-329
    if (conf->window < 8 || conf->window > 15 || conf->literal < 5 || conf->literal > 8) return TAMP_INVALID_CONF;
    if (conf->window > window_bits) return TAMP_INVALID_CONF;
    if (!conf->use_custom_dictionary)
        seed_dictionary_host(window, (size_t)1 << conf->window, conf->extended ? conf->literal : 8);
    s->conf = (uint8_t)(((conf->window - 8) << 5) | ((conf->literal - 5) << 3) | (conf->use_custom_dictionary << 2) |
                        (conf->extended << 1) | conf->dictionary_reset);
    s->flags = 1;
    return TAMP_OK;
}

tamp_res tamp_decompressor_decompress_cb(TampDecompressor* decompressor, unsigned char* output, size_t output_size,
                                         size_t* output_written_size, const unsigned char* input, size_t input_size,
                                         size_t* input_consumed_size, tamp_callback_t callback, void* user_data) {
    // One call of the reference's (decompressor.c:371-578) = one step of the resumable device decoder on this object:
    // state from private_, window from the caller's buffer, both written back afterwards.
    if (output_written_size) *output_written_size = 0;
    if (input_consumed_size) *input_consumed_size = 0;
    TampAmdDecoderState* s = reinterpret_cast<TampAmdDecoderState*>(decompressor->private_);
    const uint8_t bits_max = s->window_bits_max;
    if (bits_max < 8 || bits_max > 15 || !decompressor->window) return TAMP_ERROR;  // not initialised
    const size_t wcap = (size_t)1 << bits_max;
    // before the header is known the whole buffer may hold a custom dictionary; afterwards 1 << window bytes are live
    const size_t wlive = (s->flags & 1) ? (size_t)1 << (((s->conf >> 5) & 7) + 8) : wcap;
    std::vector<unsigned char> slot(sizeof(TampAmdDecoderState) + wcap);
    std::memcpy(slot.data(), s, sizeof *s);
    std::memcpy(slot.data() + sizeof *s, decompressor->window, wlive);
    const uint64_t zero = 0;
    static unsigned char empty = 0;
    size_t written = 0, consumed = 0;
    int8_t st = TAMP_ERROR;
    for (;;) {  // one device step per 256 MiB of input / 1 GiB of output room (the kernel's counters are 32 bits wide)
        const uint32_t ilen = (uint32_t)std::min<size_t>(input_size - consumed, 0x10000000u);
        const uint32_t ocap = (uint32_t)std::min<size_t>(output_size - written, 0x40000000u);
        uint32_t olen = 0, icons = 0;
        const int rc = tamp_batch_decompress_resume(slot.data(), slot.size(), bits_max, ilen ? input + consumed : &empty,
                                                    &zero, &ilen, ocap ? output + written : &empty, &zero, &ocap, &olen,
                                                    &st, &icons, 1, TAMP_AMD_MEM_HOST, compat_device(), nullptr);
        if (rc != TAMP_OK) return (tamp_res)rc;
        written += olen, consumed += icons;
        const bool more_in = st == TAMP_INPUT_EXHAUSTED && icons == ilen && consumed < input_size;
        const bool more_out = st == TAMP_OUTPUT_FULL && olen == ocap && written < output_size;
        if (!more_in && !more_out) break;
    }
    std::memcpy(s, slot.data(), sizeof *s);
    const size_t wnow = (s->flags & 1) ? (size_t)1 << (((s->conf >> 5) & 7) + 8) : 0;
    if (wnow) std::memcpy(decompressor->window, slot.data() + sizeof *s, wnow);
    if (output_written_size) *output_written_size = written;
    if (input_consumed_size) *input_consumed_size = consumed;
    if (st >= 0 && callback) {
        const int cb = callback(user_data, consumed, input_size);
        if (cb) return (tamp_res)cb;
    }
    return st;
}

tamp_res tamp_decompressor_decompress(TampDecompressor* decompressor, unsigned char* output, size_t output_size,
                                      size_t* output_written_size, const unsigned char* input, size_t input_size,
                                      size_t* input_consumed_size) {
    return tamp_decompressor_decompress_cb(decompressor, output, output_size, output_written_size, input, input_size,
                                           input_consumed_size, nullptr, nullptr);
}


// ---------------------------------------------------------------------------------------------
// Segment call: one piece of a stream between two flush points, with the window carried in and out.
// This is what tamp.Compressor.write()/flush()/reset_dictionary() need (compressor.c:227-241,728-881).
// ---------------------------------------------------------------------------------------------
namespace {
// One piece of a stream on the batch kernel.  finish = 1: the piece ends with tamp_compressor_flush(flush_token) -- a
// SEGMENT; finish = 0: it ends the way tamp_compressor_compress ends a call (compressor.c:681-722) and `carry` takes what
// the reference's object would still hold.  A carry that comes in is continued from (its run / extended match bytes and
// its unparsed tail are put back in front of the input: tamp_compress_kernel.hpp, kSegStateExtra).
tamp_res segment_core(const TampAmdConf* conf, int emit_header, int append_marker, int resume, int finish, int flush_token,
                      unsigned char* window_state, uint16_t* window_pos, TampAmdCarry* carry, unsigned char* output,
                      size_t output_size, size_t* output_written_size, const unsigned char* input, size_t input_size,
                      int* token_written, int device) {
    if (output_written_size) *output_written_size = 0;
    if (token_written) *token_written = 0;
    if (!conf_valid(conf) || conf->lazy_matching > 1 || !window_state || !window_pos) return TAMP_INVALID_CONF;
    if (!finish && (!carry || conf->lazy_matching)) return TAMP_AMD_BAD_ARGUMENT;  // (lazy: the cached match is not carried)
    if (input_size > 0xFFFFFF00ull) return TAMP_AMD_BAD_ARGUMENT;
    DeviceCtx* ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc != TAMP_OK) return (tamp_res)rc;
    const size_t W = (size_t)1 << conf->window;
    const size_t SS = W + kSegStateExtra;
    SegmentSpec seg;
    seg.flags = kSegSave | (resume ? kSegResume : 0) | ((finish && flush_token) ? kSegFlushToken : 0) | (finish ? 0 : kSegPartial);
    if (append_marker) {  // compressor.c:227-235: FLUSH (9 bits) padded to 16 bits instead of a header
        seg.nlead = 2, seg.lead = (uint16_t)(0xABu << 7);
    } else if (emit_header) {
        const uint8_t header = (uint8_t)(((conf->window - 8) << 5) | ((conf->literal - 5) << 3) |
                                         ((conf->use_custom_dictionary != 0) << 2) | ((conf->extended != 0) << 1) |
                                         (conf->dictionary_reset != 0));
        seg.nlead = conf->dictionary_reset ? 2 : 1, seg.lead = (uint16_t)(header << 8);
    } else {
        seg.nlead = 0, seg.lead = 0;
    }
    // what leads the input: the bytes a carried run / extended match has consumed (they are window bytes: the last byte
    // written, resp. window[pos .. pos + count)), then the carried tail of the 16-byte ring
    std::vector<unsigned char> prefix;
    std::vector<unsigned char> stbuf(SS, 0);
    std::memcpy(stbuf.data(), window_state, W);
    stbuf[W] = (unsigned char)(*window_pos & 0xFF), stbuf[W + 1] = (unsigned char)(*window_pos >> 8);
    if (carry && resume) {
        if (carry->tail_len > 16 || carry->bit_count > 31 || (carry->rle_count && carry->ext_count) ||
            (size_t)carry->ext_pos + carry->ext_count > W || ((emit_header || append_marker) && carry->bit_count))
            return TAMP_AMD_BAD_ARGUMENT;
        const unsigned char last = window_state[(*window_pos - 1) & (W - 1)];
        prefix.insert(prefix.end(), carry->rle_count, last);
        prefix.insert(prefix.end(), window_state + carry->ext_pos, window_state + carry->ext_pos + carry->ext_count);
        prefix.insert(prefix.end(), carry->tail, carry->tail + carry->tail_len);
        stbuf[W + 3] = carry->rle_count, stbuf[W + 4] = carry->ext_count, stbuf[W + 5] = carry->bit_count;
        stbuf[W + 6] = (unsigned char)(carry->ext_pos & 0xFF), stbuf[W + 7] = (unsigned char)(carry->ext_pos >> 8);
        for (int k = 0; k < 4; k++) stbuf[W + 16 + k] = (unsigned char)(carry->bits >> (8 * k));
    }
    const uint64_t zero = 0;
    if (finish && !flush_token && !resume && emit_header && !append_marker && prefix.empty() && !conf->extended &&
        !conf->lazy_matching && conf->literal == 8 && conf->window <= 14 && input_size >= ((size_t)256 << 10) &&
        input_size <= 0xFFFFFF00ull) {
        // A whole fresh v1 stream in one finishing call: the batch call takes it as ONE stream and spreads its blocks over
        // all workgroups (launch_compress_blocks).  The object afterwards holds what the reference's would: every consumed
        // byte was written (compressor.c:651-657), so the window is the stream's last W bytes at their ring positions.
        const uint32_t ilen1 = (uint32_t)input_size;
        const uint32_t ocap1 = (uint32_t)(output_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : output_size);
        uint32_t olen1 = 0;
        int8_t st1 = TAMP_ERROR;
        rc = tamp_batch_compress(conf, conf->use_custom_dictionary ? window_state : nullptr, input, &zero, &ilen1, output, &zero,
                                 &ocap1, &olen1, &st1, 1, ilen1, TAMP_AMD_MEM_HOST, device, nullptr);
        if (rc != TAMP_OK) return (tamp_res)rc;
        if (output_written_size) *output_written_size = olen1;
        if (st1 == TAMP_OK) {
            const size_t first = input_size > W ? input_size - W : 0;
            for (size_t p2 = first; p2 < input_size; p2++) window_state[p2 & (W - 1)] = input[p2];
            *window_pos = (uint16_t)(input_size & (W - 1));
            if (carry) std::memset(carry, 0, sizeof *carry);
        }
        return st1;
    }
    DevBuf d_in, d_out, d_io, d_il, d_oo, d_oc, d_ol, d_st, d_state, d_dict;
    if ((uint64_t)prefix.size() + (uint64_t)input_size > 0xFFFFFFFFull) return TAMP_AMD_BAD_ARGUMENT;  // (32-bit stream lengths)
    const uint32_t ilen = (uint32_t)(prefix.size() + input_size);
    const uint32_t ocap = (uint32_t)(output_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : output_size);
    HIP_OK(d_in.alloc((size_t)ilen + 64));
    HIP_OK(d_out.alloc(output_size));
    HIP_OK(d_io.alloc(8));
    HIP_OK(d_il.alloc(4));
    HIP_OK(d_oo.alloc(8));
    HIP_OK(d_oc.alloc(4));
    HIP_OK(d_ol.alloc(4));
    HIP_OK(d_st.alloc(1));
    HIP_OK(d_state.alloc(SS));
    hipStream_t st = nullptr;
    if (!prefix.empty()) HIP_OK(hipMemcpyAsync(d_in.p, prefix.data(), prefix.size(), hipMemcpyHostToDevice, st));
    if (input_size)
        HIP_OK(hipMemcpyAsync(d_in.as<uint8_t>() + prefix.size(), input, input_size, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_io.p, &zero, 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_il.p, &ilen, 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_oo.p, &zero, 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_oc.p, &ocap, 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_state.p, stbuf.data(), SS, hipMemcpyHostToDevice, st));
    const uint8_t* dict = nullptr;
    if (!resume && conf->use_custom_dictionary) {  // a fresh stream with a custom dictionary: the state buffer holds it
        HIP_OK(d_dict.alloc(W));
        HIP_OK(hipMemcpyAsync(d_dict.p, window_state, W, hipMemcpyHostToDevice, st));
        dict = d_dict.as<uint8_t>();
    }
    rc = launch_compress(ctx, conf, dict, d_in.as<uint8_t>(), d_io.as<uint64_t>(), d_il.as<uint32_t>(),
                         d_out.as<uint8_t>(), d_oo.as<uint64_t>(), d_oc.as<uint32_t>(), d_ol.as<uint32_t>(),
                         d_st.as<int8_t>(), 1, ilen ? ilen : 16, st, &seg, d_state.as<uint8_t>());
    if (rc != TAMP_OK) return (tamp_res)rc;
    uint32_t olen = 0;
    int8_t status = TAMP_ERROR;
    HIP_OK(hipMemcpyAsync(&olen, d_ol.p, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&status, d_st.p, 1, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(stbuf.data(), d_state.p, SS, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    // A piece that did not complete leaves window and carry as they came in, so its bytes must not count either: a caller
    // that consumed them and offered the piece again would emit them twice.  (A finishing call keeps the reference's
    // contract -- what fitted is delivered with TAMP_OUTPUT_FULL, compressor.c:65-75.)
    if (!finish && status != TAMP_OK) olen = 0;
    if (olen) HIP_OK(hipMemcpy(output, d_out.p, olen, hipMemcpyDeviceToHost));
    if (output_written_size) *output_written_size = olen;
    if (status == TAMP_OK) {
        std::memcpy(window_state, stbuf.data(), W);
        *window_pos = (uint16_t)(stbuf[W] | (stbuf[W + 1] << 8));
        if (token_written) *token_written = stbuf[W + 2];
        if (carry) {
            std::memset(carry, 0, sizeof *carry);
            if (!finish) {
                carry->rle_count = stbuf[W + 3], carry->ext_count = stbuf[W + 4], carry->bit_count = stbuf[W + 5];
                carry->ext_pos = (uint16_t)(stbuf[W + 6] | (stbuf[W + 7] << 8));
                carry->bits = (uint32_t)stbuf[W + 16] | ((uint32_t)stbuf[W + 17] << 8) | ((uint32_t)stbuf[W + 18] << 16) |
                              ((uint32_t)stbuf[W + 19] << 24);
                const uint32_t parsed = (uint32_t)stbuf[W + 9] | ((uint32_t)stbuf[W + 10] << 8) |
                                        ((uint32_t)stbuf[W + 11] << 16) | ((uint32_t)stbuf[W + 12] << 24);
                const uint32_t left = ilen - parsed;  // < 16: the ring never stays full (compressor.c:704-718)
                if (parsed > ilen || left > 15) return TAMP_ERROR;
                carry->tail_len = (uint8_t)left;
                for (uint32_t j = 0; j < left; j++) {
                    const size_t at = (size_t)parsed + j;
                    carry->tail[j] = at < prefix.size() ? prefix[at] : input[at - prefix.size()];
                }
            }
        }
    } else if (!finish && status == TAMP_OUTPUT_FULL) {
        return TAMP_OUTPUT_FULL;  // (the carry is left as it came in: give the piece tamp_amd_compress_bound(input_size + 271) of room)
    }
    return status;
}
}  // namespace

tamp_res tamp_amd_compress_segment(const TampAmdConf* conf, int emit_header, int append_marker, int resume,
                                   int flush_token, unsigned char* window_state, uint16_t* window_pos,
                                   unsigned char* output, size_t output_size, size_t* output_written_size,
                                   const unsigned char* input, size_t input_size, int* token_written, int device) {
    return segment_core(conf, emit_header, append_marker, resume, 1, flush_token, window_state, window_pos, nullptr, output,
                        output_size, output_written_size, input, input_size, token_written, device);
}

tamp_res tamp_amd_compress_piece(const TampAmdConf* conf, int emit_header, int append_marker, int resume, int finish,
                                 int flush_token, unsigned char* window_state, uint16_t* window_pos, TampAmdCarry* carry,
                                 unsigned char* output, size_t output_size, size_t* output_written_size,
                                 const unsigned char* input, size_t input_size, int* token_written, int device) {
    if (!carry) return TAMP_AMD_BAD_ARGUMENT;
    return segment_core(conf, emit_header, append_marker, resume, finish, flush_token, window_state, window_pos, carry, output,
                        output_size, output_written_size, input, input_size, token_written, device);
}

}  // extern "C"
