// libtoricenv: C-ABI over the HIP kernels (see include/toricenv.h for the contract and the
// reference interfaces each entry point replaces).  gfx950 only; no CPU path: every entry
// point needs a HIP device and reports TQ_E_HIP otherwise.
#include "toricenv.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "kernels.hpp"
#include "stream_write.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// A failed HIP call is reported through the return code; the runtime's sticky "last error" is cleared so that the
// caller's next launch check (PyTorch's, say) does not trip over it.
#define HIPCHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return fail(TQ_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                      \
    } while (0)

#define KCHECK() HIPCHECK(hipGetLastError())

constexpr int MAX_DEVICES = 16;
constexpr int kSizes[] = {3, 5, 7, 9, 11, 13, 15, 17, 19, 21};

bool size_ok(int d) {
    for (int s : kSizes) if (s == d) return true;
    return false;
}
int size_slot(int d) { return (d - 3) / 2; }

// per-device caches shared by handles and the stateless entry points
struct DeviceCtx {
    std::mutex mu;
    uint16_t* lut[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int* err = nullptr;             // error latch of the stateless entry points (tq_states_check)
    void* ws = nullptr;             // scratch of the tq_states_persp_* entry points (tq_states_reserve)
    size_t ws_bytes = 0;
    int num_cus = 0;
};
constexpr int SPLIT_MAX = 256;       // persistent workgroups of the stack write: one per CU of an MI355X
constexpr int SPLIT_LG = 13;         // the scan's table cuts the stack into 1 << SPLIT_LG fine parts: 32 per workgroup
constexpr int SPLIT_ENTRIES = (1 << SPLIT_LG) + 1;
constexpr int N_SLOT_SETS = 8;       // sets of slot counters of the stack write, used in turn
// Fine parts (of 32) that the workgroup of an odd XCD hands to its even neighbour (stream_write.hpp: the odd XCDs of an
// MI355X store ~20 % slower; sweep in profiles/r04_xcd_bias_sweep.txt).  tq_set_xcd_bias / TORICENV_XCD_BIAS = 0..16.
// Used for d >= 7 and 32- / 16-bit stacks: smaller lattices and the u8 stack are bound by the producers, where unequal
// shares only cost (u8, d=7, in the two-stream loop: 0.0931 ms with equal shares, 0.1000 ms with 37 : 27).
constexpr int XCD_BIAS_DEFAULT = 5;
static std::atomic<int> g_xcd_bias{-1};
static int xcd_bias() {
    int b = g_xcd_bias.load(std::memory_order_relaxed);
    if (b < 0) {
        b = XCD_BIAS_DEFAULT;
        if (const char* e = getenv("TORICENV_XCD_BIAS"); e && *e) { const int v = atoi(e); b = v < 0 ? 0 : (v > 16 ? 16 : v); }
        g_xcd_bias.store(b, std::memory_order_relaxed);
    }
    return b;
}
DeviceCtx g_ctx[MAX_DEVICES];

inline dim3 grid1(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// alignment contract of include/toricenv.h: the kernels use 16-byte vector accesses on these
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
#define REQUIRE_ALIGNED16(p, name)                                                        \
    do {                                                                                  \
        if ((p) && !aligned16(p)) return fail(TQ_E_INVALID, "%s must be 16-byte aligned", name); \
    } while (0)

int decode_latch(int flag) {
    if (flag & tq::ERR_ACTION) return fail(TQ_E_ACTION, "an action outside the lattice or with op not in 1..3 was applied");
    if (flag & tq::ERR_CAPACITY) return fail(TQ_E_CAPACITY, "perspective stack capacity exceeded");
    if (flag & tq::ERR_INTERNAL)
        return fail(TQ_E_INVALID, "stack write refused: the offsets are not the scan of these lattices' perspective counts "
                                  "(stale, shifted or from another batch), or a wave gave up waiting");
    if (flag & tq::ERR_INDEX) return fail(TQ_E_INDEX, "tq_reset_idx: an index outside [0, n_envs) was given");
    if (flag & tq::ERR_RESET_DUP) return fail(TQ_E_INDEX, "tq_reset_idx: an index was listed more than once");
    if (flag & tq::ERR_RESET_ROUNDS)
        return fail(TQ_E_RESET, "a reset drew %d rounds without producing a defect (p_error too small)", tq::MAX_RESET_ROUNDS);
    return TQ_OK;
}

#define DISPATCH_D(d, CALL)          \
    switch (d) {                     \
        case 3: CALL(3); break;      \
        case 5: CALL(5); break;      \
        case 7: CALL(7); break;      \
        case 9: CALL(9); break;      \
        case 11: CALL(11); break;    \
        case 13: CALL(13); break;    \
        case 15: CALL(15); break;    \
        case 17: CALL(17); break;    \
        case 19: CALL(19); break;    \
        case 21: CALL(21); break;    \
        default: return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d); \
    }

int current_device(int* dev) {
    HIPCHECK(hipGetDevice(dev));
    if (*dev < 0 || *dev >= MAX_DEVICES) return fail(TQ_E_INVALID, "device %d out of range", *dev);
    return TQ_OK;
}

int get_lut(int dev, int d, hipStream_t stream, const uint16_t** out) {
    DeviceCtx& c = g_ctx[dev];
    std::lock_guard<std::mutex> lock(c.mu);
    const int slot = size_slot(d);
    if (!c.lut[slot]) {
        const int nq = 2 * d * d;
        const size_t bytes = (2 * (size_t)nq * nq + 15) & ~(size_t)15;
        uint16_t* p = nullptr;
        HIPCHECK(hipMalloc(&p, bytes));
        HIPCHECK(hipMemsetAsync(p, 0, bytes, stream));
#define CALL(D) hipLaunchKernelGGL(tq::k_build_lut<D>, grid1((int64_t)nq * nq, 256), dim3(256), 0, stream, p)
        DISPATCH_D(d, CALL)
#undef CALL
        KCHECK();
        HIPCHECK(hipStreamSynchronize(stream));     // once per (device, d)
        c.lut[slot] = p;
    }
    if (!c.err) {
        HIPCHECK(hipMalloc((void**)&c.err, sizeof(int)));
        HIPCHECK(hipMemset(c.err, 0, sizeof(int)));
    }
    if (!c.num_cus) {
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, dev));
        c.num_cus = prop.multiProcessorCount;
    }
    *out = c.lut[slot];
    return TQ_OK;
}

int launch_scan(const int32_t* counts, int64_t* partial, bool partial_valid, int64_t* offsets, int32_t* counts_out,
                int64_t n, hipStream_t stream, int32_t* split) {
    const unsigned blocks = (unsigned)((n + tq::SCAN_CHUNK - 1) / tq::SCAN_CHUNK);
    if (!partial_valid)
        hipLaunchKernelGGL(tq::k_scan_partials, dim3((unsigned)((n + tq::PART_BLOCK - 1) / tq::PART_BLOCK)), dim3(256), 0,
                           stream, counts, partial, n);
    hipLaunchKernelGGL(tq::k_scan_final, dim3(blocks), dim3(256), 0, stream, counts, (const int64_t*)partial, offsets,
                       counts_out, n, split, SPLIT_LG);
    KCHECK();
    return TQ_OK;
}

// The stack write (stream_write.hpp): SPLIT_MAX persistent workgroups, one per CU, each with its own contiguous
// part of the stack.  Waves per workgroup by role, from the sweeps of tools/stream_tune.hip (profiles/r04_stream_tune_*):
//   * storers: 4 saturate a CU's store path (2-3 for d <= 5, whose short rows leave the producers more to do);
//   * positions waves: 1 for a 4-byte stack; 2 for 16- and 8-bit stacks, which carry 2-4 times the perspectives per
//     byte stored (u8, d=7: 5.3 -> 6.0-6.2 TB/s; d=9: 6.3 -> 6.5-6.6);
//   * producers: the rest.  d >= 7 is bound by the store path whatever the mix (d >= 13: 6.9-7.0 TB/s for f32, bf16
//     and u8 with 2 to 11 producers -- since Bits::get stopped pinning bitsets in scratch / LDS; before that fix 3
//     producers beat 7 there).  Small lattices are producer-bound (a d=3 lattice is 1.3 KB of output against ~3000
//     cycles of set-up), so d <= 5 gets every wave that is left.
template <int D, int ES>
struct StreamCfg {
    static constexpr int NS = D <= 5 ? (D == 5 && ES == 2 ? 3 : 2) : 4;
    static constexpr int NPW = (ES < 4 && D >= 5) ? 2 : 1;
    static constexpr int NP = D >= 17 ? 3 : (D >= 13 ? 8 - NPW : 16 - NS - NPW);   // d >= 17: 5-7 words per plane, 8 waves = 256 VGPRs each
    static constexpr int CPW = 8, RB = 14, RP = 12;          // 8 KiB windows, 64 KB bit ring, 16 KB position ring
};
// Tried and not adopted for the producer-bound small lattices: two workgroups per CU (512 workgroups, half-size rings):
// 39 -> 45 us at d=3, no gain at d=5 -- the producers are bound by the CU's instruction issue, not by latency
// (profiles/r04_stream_tune_small_two_wgs_per_cu.txt).
template <int D, typename OutT>
int launch_persp_write_t(const uint64_t* vp, int64_t n, const int64_t* offsets, void* out, int32_t* pos,
                         int64_t capacity, int* err, hipStream_t stream, int64_t first, int64_t count,
                         const int32_t* split, unsigned int* slots, int bias) {
    using C = StreamCfg<D, (int)sizeof(OutT)>;
    // a workgroup's part of the stack is addressed with 32-bit element offsets
    if ((double)count * (2.0 * D * D) * (2.0 * D * D) / SPLIT_MAX * 1.5 > 2.0e9)    // (the largest share is 1.5 of the mean)
        return fail(TQ_E_INVALID, "lattice range too large for one stack write (%lld lattices of d=%d)", (long long)count, D);
    hipLaunchKernelGGL((tq::k_persp_stream<D, OutT, C::NS, C::NP, C::CPW, C::RB, C::RP, false, C::NPW>), dim3(SPLIT_MAX),
                       dim3(64 * (C::NS + C::NPW + C::NP)), 0, stream, vp, n, offsets, (OutT*)out, pos, capacity, err, first, first + count, split,
                       SPLIT_LG, (D >= 7 && sizeof(OutT) >= 2) ? bias : 0, slots, (unsigned long long*)nullptr);
    KCHECK();
    return TQ_OK;
}

// split == nullptr: the cut points did not come with the scan of these offsets (a lattice sub-range, or offsets from
// elsewhere): every workgroup of the write finds its own two (find_cut)
template <int D>
int launch_persp_write(const uint64_t* vp, int64_t n, const int64_t* offsets, void* out, int32_t* pos,
                       int64_t capacity, int dtype, int* err, hipStream_t stream, int64_t first, int64_t count,
                       const int32_t* split, unsigned int* slots = nullptr, int bias = 0) {
    if (count == 0) return TQ_OK;
    switch (dtype) {
        case TQ_F32: return launch_persp_write_t<D, float>(vp, n, offsets, out, pos, capacity, err, stream, first, count, split, slots, bias);
        case TQ_F16: return launch_persp_write_t<D, __half>(vp, n, offsets, out, pos, capacity, err, stream, first, count, split, slots, bias);
        case TQ_BF16: return launch_persp_write_t<D, tq::bf16_t>(vp, n, offsets, out, pos, capacity, err, stream, first, count, split, slots, bias);
        case TQ_U8: return launch_persp_write_t<D, uint8_t>(vp, n, offsets, out, pos, capacity, err, stream, first, count, split, slots, bias);
        default: return fail(TQ_E_INVALID, "unknown dtype %d", dtype);
    }
}

}  // namespace

struct tq_env {
    int n, d, w, device;
    uint64_t seed;
    int64_t first_env;
    double terminal_reward;
    int max_steps;
    int min_err;           // config "min_qubit_errors": 0 = depolarizing sampler, n > 0 = exactly n errors per reset
    tq::PerrSchedule sched;
    uint64_t* planes;      // [6][W][N]: the lattices
    uint64_t* planes_alt;  // the second buffer: tq_actor_step reads `planes`, writes this one, then the two swap (so the
                           // stack write of the pre-step lattices can run beside the step on another stream)
    uint64_t* prev;        // [2][W][N]
    uint32_t* episodes;
    uint32_t* steps;
    int32_t* counts;
    int64_t* partial;      // level-1 sums of the scan: one per 256 counts
    bool partial_valid;    // left current by the last all-lattice kernel (false after tq_reset_idx)
    double* p_roof;
    int* err;              // device error latch
    uint32_t* mark;        // [N] epoch of the last indexed reset that touched the lattice (duplicate detection)
    uint32_t reset_epoch;
    void* tblock;          // packed block of N slots: scratch of tq_transition_write
    const uint16_t* lut;
    int num_cus;
    int32_t* split[2];     // cut points of the stack write, written by the scan (tq_persp_count); two tables take turns, so
    const int64_t* split_for[2];   // the scan of the next step does not overwrite what a running write reads; the offsets
    int split_last;        // array each belongs to, and which one was written last
    unsigned int* slots;   // N_SLOT_SETS sets of STREAM_SLOT_WORDS counters: the workgroups of a stack write take their shares by XCD (stream_write.hpp)
    unsigned write_seq;    // and leave them zero; write i uses set i % N, so N writes of one handle may be in flight
    int xcd_bias;          // this handle's share setting (tq_env_set_xcd_bias), or -1: the process-wide one
};

namespace {
// Makes the handle's device current for the duration of one entry point and restores the caller's
// device on the way out (PyTorch callers already run under torch.cuda.device(...); C callers must
// not find their current device changed behind their back).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(const tq_env* h) {
        if (!h) return fail(TQ_E_INVALID, "NULL handle");
        return enter_device(h->device);
    }
    int enter_device(int device) {
        if (int rc = current_device(&prev)) return rc;
        if (prev != device) {
            HIPCHECK(hipSetDevice(device));
            switched = true;
        }
        return TQ_OK;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define HANDLE(h)                                  \
    DeviceGuard _guard;                            \
    if (int _rc = _guard.enter(h)) return _rc;     \
    hipStream_t stream = (hipStream_t)stream_
}  // namespace

extern "C" {

int tq_version(void) { return TQ_VERSION; }
int tq_set_xcd_bias(int bias) {
    if (bias < 0 || bias > 16) return fail(TQ_E_INVALID, "xcd bias %d outside 0..16", bias);
    g_xcd_bias.store(bias, std::memory_order_relaxed);
    return TQ_OK;
}
int tq_get_xcd_bias(void) { return xcd_bias(); }
int tq_env_set_xcd_bias(tq_env* h, int bias) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    if (bias < -1 || bias > 16) return fail(TQ_E_INVALID, "xcd bias %d outside -1..16", bias);
    h->xcd_bias = bias;
    return TQ_OK;
}
int tq_env_get_xcd_bias(const tq_env* h) { return !h ? TQ_E_INVALID : (h->xcd_bias >= 0 ? h->xcd_bias : xcd_bias()); }

// ---- stack buffers backed by 2 MiB physical chunks (HIP virtual memory API)
namespace {
// A tq_stack_alloc buffer: 2 MiB physical chunks (HIP virtual memory API), each mapped once behind one virtual range.
//
// Two hazards of this API on this stack (ROCm 7.2, MI355X), both met in round 3 and both guarded against here:
//  * hipMemUnmap + hipMemMap of a different chunk at the same address leaves STALE TRANSLATIONS behind: kernels went on
//    reading and writing the previous chunk through that address (78 % of a 600 MiB buffer, indefinitely: a
//    hipDeviceSynchronize, a second of sleep, a 1 MiB hipMalloc + hipFree changed nothing; a 64 MiB hipMalloc + hipFree
//    or a stream creation did).  Writes through such aliased translations LOOK fast -- 7.0 TB/s for a stack that is
//    wrong in 70 % of its elements -- which cost this round an afternoon (profiles/r03_stack_write_ab.txt section 12).
//    Nothing is ever re-mapped here.
//  * the same happens across buffers when hipMemAddressFree gives an address range back and a later reservation
//    receives it again: the new buffer reads and writes the previous tenant's pages.  Address ranges are therefore
//    never given back (2 MiB-rounded buffer sizes out of a 128 TiB address space).
// And every buffer is CHECKED before it is handed out: the driver is made to invalidate the device's translations
// (a 64 MiB hipMalloc + hipFree), every 2 MiB page gets a tag of its own through its address, and every workgroup of a
// grid that covers all CUs reads every page's tag back.
struct ChunkedAlloc {
    char* va = nullptr;
    size_t bytes = 0, chunk = 0, mapped = 0;
    int device = 0;
};
std::mutex g_alloc_mu;
std::vector<ChunkedAlloc> g_allocs;

__global__ void k_page_tag(char* base, size_t chunk, size_t n, int write) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *reinterpret_cast<unsigned long long*>(base + i * chunk) = write ? (0x7a6b5c4d3e2f1001ull ^ (unsigned long long)i) : 0ull;
}
__global__ void k_page_check(const char* base, size_t chunk, size_t n, int* bad) {
    int mine = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x)
        mine += *reinterpret_cast<const volatile unsigned long long*>(base + i * chunk) != (0x7a6b5c4d3e2f1001ull ^ (unsigned long long)i);
    if (mine) atomicAdd(bad, mine);
}
// 0 = every page of the buffer is reached through its own address from everywhere; > 0 = pages that are not; < 0 = HIP error
long long translation_check(char* va, size_t chunk, size_t n) {
    void* flush = nullptr;                                    // a mapping of its own, made and torn down: the tear-down
    if (hipMalloc(&flush, (size_t)64 << 20) != hipSuccess) return -1;      // invalidates the process's translations
    if (hipFree(flush) != hipSuccess) return -1;
    int* scratch = nullptr;
    if (hipMalloc((void**)&scratch, sizeof(int)) != hipSuccess) return -1;
    long long result = -1;
    if (hipMemsetAsync(scratch, 0, sizeof(int), nullptr) == hipSuccess) {
        hipLaunchKernelGGL(k_page_tag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, va, chunk, n, 1);
        hipLaunchKernelGGL(k_page_check, dim3(2048), dim3(64), 0, nullptr, (const char*)va, chunk, n, scratch);
        hipLaunchKernelGGL(k_page_check, dim3(2048), dim3(64), 0, nullptr, (const char*)va, chunk, n, scratch);
        hipLaunchKernelGGL(k_page_tag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, va, chunk, n, 0);
        int bad = 0;
        if (hipMemcpy(&bad, scratch, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess) result = bad;
    }
    (void)hipFree(scratch);
    return result;
}
void chunked_unmap(char* va, size_t chunk, size_t count) {    // every chunk was mapped by a call of its own and is unmapped the same way
    for (size_t i = 0; i < count; ++i) (void)hipMemUnmap(va + i * chunk, chunk);
}
}  // namespace

int tq_stack_alloc(int device, uint64_t bytes, void** out) {
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
    *out = nullptr;
    if (bytes == 0) return fail(TQ_E_INVALID, "bytes must be > 0");
    DeviceGuard guard;
    if (int rc = guard.enter_device(device)) return rc;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    HIPCHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t chunk = (size_t)2 << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = ((size_t)bytes + chunk - 1) / chunk;
    if (n > ((size_t)1 << 40) / chunk) return fail(TQ_E_INVALID, "%llu bytes is more than a device holds", (unsigned long long)bytes);
    void* va = nullptr;
    HIPCHECK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
    size_t mapped = 0;
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        hipMemGenericAllocationHandle_t hnd;
        e = hipMemCreate(&hnd, chunk, &prop, 0);
        if (e != hipSuccess) break;
        e = hipMemMap((char*)va + i * chunk, chunk, 0, hnd, 0);
        (void)hipMemRelease(hnd);                            // the mapping keeps the memory alive
        if (e == hipSuccess) ++mapped;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, n * chunk, &acc, 1);
    }
    if (e == hipSuccess) e = hipMemsetAsync(va, 0, n * chunk, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        chunked_unmap((char*)va, chunk, mapped);             // the address range stays reserved (see above)
        (void)hipGetLastError();
        return fail(TQ_E_HIP, "chunked allocation of %llu bytes failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    const long long bad = translation_check((char*)va, chunk, n);
    if (bad != 0) {
        chunked_unmap((char*)va, chunk, mapped);
        (void)hipGetLastError();
        if (bad < 0) return fail(TQ_E_HIP, "the check of the new buffer's address translations could not run (a HIP call failed)");
        return fail(TQ_E_HIP, "%lld pages of the new buffer are not reached through their own addresses (stale address translations): "
                              "not handing it out", bad);
    }
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    g_allocs.push_back(ChunkedAlloc{(char*)va, n * chunk, chunk, mapped, device});
    *out = va;
    return TQ_OK;
}

int tq_stack_free(void* ptr) {
    if (!ptr) return TQ_OK;
    ChunkedAlloc a;
    {
        std::lock_guard<std::mutex> lock(g_alloc_mu);
        for (size_t i = 0; i < g_allocs.size(); ++i)
            if (g_allocs[i].va == ptr) { a = g_allocs[i]; g_allocs[i] = g_allocs.back(); g_allocs.pop_back(); break; }
    }
    if (!a.va) return fail(TQ_E_INVALID, "pointer did not come from tq_stack_alloc");
    DeviceGuard guard;
    if (int rc = guard.enter_device(a.device)) { std::lock_guard<std::mutex> lock(g_alloc_mu); g_allocs.push_back(a); return rc; }
    (void)hipDeviceSynchronize();
    chunked_unmap(a.va, a.chunk, a.mapped);                  // the physical chunks go back; the address range is not given back
    (void)hipGetLastError();
    return TQ_OK;
}
const char* tq_last_error(void) { return g_err; }

int tq_create(tq_env** out, int n_envs, int d, int device, uint64_t seed, int64_t first_env_id) {
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_envs <= 0) return fail(TQ_E_INVALID, "n_envs must be > 0 (got %d)", n_envs);
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (first_env_id < 0 || first_env_id + n_envs > 0xFFFFFFFFll)
        return fail(TQ_E_INVALID, "global env ids must fit in 32 bits");
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= MAX_DEVICES)
        return fail(TQ_E_INVALID, "device %d not available (%d HIP devices)", device, ndev);
    DeviceGuard guard;                                        // the caller's current device is restored on return
    if (int rc = guard.enter_device(device)) return rc;
    tq_env* h = new (std::nothrow) tq_env();
    if (!h) return fail(TQ_E_INVALID, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->n = n_envs; h->d = d; h->w = (d * d + 63) / 64; h->device = device;
    h->seed = seed; h->first_env = first_env_id;
    h->terminal_reward = 100.0; h->max_steps = 75;
    h->xcd_bias = -1;
    h->sched = tq::PerrSchedule{TQ_PERR_FIXED, 0.1, 0.1, 0.1, 0.0};
    const size_t N = (size_t)n_envs, W = (size_t)h->w;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemset(*p, 0, bytes); };
    alloc((void**)&h->planes, 6 * W * N * 8);
    alloc((void**)&h->planes_alt, 6 * W * N * 8);
    alloc((void**)&h->prev, 2 * W * N * 8);
    alloc((void**)&h->episodes, N * 4);
    alloc((void**)&h->steps, N * 4);
    alloc((void**)&h->counts, N * 4 + 32);                      // +32: int4 tail loads of the scan stay in bounds
    alloc((void**)&h->partial, ((N + tq::PART_BLOCK - 1) / tq::PART_BLOCK) * 8);
    alloc((void**)&h->p_roof, N * 8);
    alloc((void**)&h->err, 4);
    alloc((void**)&h->mark, N * 4);
    alloc(&h->tblock, (size_t)tq::block_bytes(h->w, n_envs));
    alloc((void**)&h->split[0], (SPLIT_ENTRIES + 3) * sizeof(int32_t));
    alloc((void**)&h->split[1], (SPLIT_ENTRIES + 3) * sizeof(int32_t));
    alloc((void**)&h->slots, N_SLOT_SETS * tq::STREAM_SLOT_WORDS * sizeof(unsigned int));
    h->reset_epoch = 0;
    if (e != hipSuccess) { tq_destroy(h); return fail(TQ_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }
    if (int rc = get_lut(device, d, nullptr, &h->lut)) { tq_destroy(h); return rc; }
    h->num_cus = g_ctx[device].num_cus;
    // set-up calls allocate AND synchronise (toricenv.h): the memsets above ran on the null stream, and a
    // caller's non-blocking stream does not order itself behind that -- the first kernel of a fresh handle
    // (k_reset reads episodes / mark) must find them zero.
    if (hipError_t se = hipStreamSynchronize(nullptr); se != hipSuccess) {
        tq_destroy(h);
        return fail(TQ_E_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(se));
    }
    *out = h;
    return TQ_OK;
}

int tq_destroy(tq_env* h) {
    if (!h) return TQ_OK;
    DeviceGuard guard;
    (void)guard.enter_device(h->device);
    (void)hipFree(h->mark); (void)hipFree(h->tblock); (void)hipFree(h->split[0]); (void)hipFree(h->split[1]); (void)hipFree(h->slots);
    (void)hipFree(h->planes); (void)hipFree(h->planes_alt); (void)hipFree(h->prev); (void)hipFree(h->episodes); (void)hipFree(h->steps);
    (void)hipFree(h->counts); (void)hipFree(h->partial); (void)hipFree(h->p_roof); (void)hipFree(h->err);
    delete h;
    return TQ_OK;
}

int tq_set_params(tq_env* h, double p_error_default, double terminal_reward, int max_steps_per_episode) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    // p_error = 0 can never produce a defect: every reset would spin through MAX_RESET_ROUNDS rounds.  With the
    // fixed-n sampler (min_qubit_errors > 0, set first) p_error is not used at all and 0 is accepted.
    const bool p_unused = h->min_err > 0;
    if (!((p_error_default > 0.0 || (p_unused && p_error_default == 0.0)) && p_error_default <= 1.0))
        return fail(TQ_E_INVALID, "p_error must be in (0,1] (0 is accepted only with min_qubit_errors > 0)");
    if (max_steps_per_episode < 1) return fail(TQ_E_INVALID, "max_steps_per_episode must be >= 1");
    h->sched.p_default = p_error_default;
    h->terminal_reward = terminal_reward;
    h->max_steps = max_steps_per_episode;
    return TQ_OK;
}

int tq_set_min_qubit_errors(tq_env* h, int n_errors) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    if (n_errors < 0 || n_errors > 2 * h->d * h->d) return fail(TQ_E_INVALID, "min_qubit_errors must be in [0, 2*d*d]");
    if (n_errors == 0 && !(h->sched.p_default > 0.0))
        return fail(TQ_E_INVALID, "min_qubit_errors = 0 selects the depolarizing sampler, which needs p_error in (0,1]");
    h->min_err = n_errors;
    return TQ_OK;
}

int tq_set_perror_schedule(tq_env* h, int strategy, double p_start, double p_final, double p_delta) {
    DeviceGuard guard;
    if (int rc = guard.enter(h)) return rc;
    if (strategy < TQ_PERR_FIXED || strategy > TQ_PERR_RANDOM) return fail(TQ_E_INVALID, "unknown p_error strategy %d", strategy);
    if (!(p_start > 0.0 && p_start <= 1.0 && p_final > 0.0 && p_final <= 1.0 && p_delta >= 0.0))
        return fail(TQ_E_INVALID, "p_error schedule needs 0 < p_start, p_final <= 1 and p_delta >= 0");
    h->sched.strategy = strategy; h->sched.p_start = p_start; h->sched.p_final = p_final; h->sched.p_delta = p_delta;
    // env_p_errors = ones * p_start (Actor_mp.py:46)
    double* host = new (std::nothrow) double[h->n];
    if (!host) return fail(TQ_E_INVALID, "out of host memory");
    for (int i = 0; i < h->n; ++i) host[i] = p_start;
    hipError_t e = hipMemcpy(h->p_roof, host, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice);
    delete[] host;
    HIPCHECK(e);
    return TQ_OK;
}

int tq_num_envs(const tq_env* h) { return h ? h->n : 0; }
int tq_size(const tq_env* h) { return h ? h->d : 0; }

int tq_reset_all(tq_env* h, const double* p_err, void* stream_) {
    HANDLE(h);
#define CALL(D) hipLaunchKernelGGL(tq::k_reset<D>, grid1(h->n, 256), dim3(256), 0, stream, h->planes, h->episodes, \
        h->steps, h->counts, (const int32_t*)nullptr, 0, p_err, h->sched.p_default, h->seed, h->first_env, (int64_t)h->n, \
        h->partial, h->mark, 0u, h->min_err, h->err)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    h->partial_valid = true;
    return TQ_OK;
}

int tq_reset_idx(tq_env* h, const int32_t* idx, int n_idx, const double* p_err, void* stream_) {
    HANDLE(h);
    if (n_idx < 0 || (n_idx > 0 && !idx)) return fail(TQ_E_INVALID, "bad idx / n_idx");
    if (n_idx == 0) return TQ_OK;
    if (++h->reset_epoch == 0) {                             // epoch 0 is the mark array's initial value
        HIPCHECK(hipMemsetAsync(h->mark, 0, 4 * (size_t)h->n, stream));
        h->reset_epoch = 1;
    }
#define CALL(D) hipLaunchKernelGGL(tq::k_reset<D>, grid1(n_idx, 256), dim3(256), 0, stream, h->planes, h->episodes, \
        h->steps, h->counts, idx, n_idx, p_err, h->sched.p_default, h->seed, h->first_env, (int64_t)h->n, (int64_t*)nullptr, \
        h->mark, h->reset_epoch, h->min_err, h->err)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    h->partial_valid = false;
    return TQ_OK;
}

int tq_step(tq_env* h, const int32_t* actions, float* rewards, uint8_t* terminals, void* stream_) {
    HANDLE(h);
    if (!actions) return fail(TQ_E_INVALID, "actions is NULL");
    REQUIRE_ALIGNED16(actions, "actions");
#define CALL(D) hipLaunchKernelGGL(tq::k_step<D>, grid1(h->n, 256), dim3(256), 0, stream, h->planes, h->prev, actions, \
        rewards, terminals, h->steps, h->counts, (float)h->terminal_reward, (int64_t)h->n, h->err, h->partial)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    h->partial_valid = true;
    return TQ_OK;
}

int tq_get_state(tq_env* h, uint8_t* out, void* stream_) {
    HANDLE(h);
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
    const int64_t total = (int64_t)h->n * 2 * h->d * h->d;
#define CALL(D) hipLaunchKernelGGL(tq::k_get_state<D>, grid1(total, 256), dim3(256), 0, stream, h->planes, (int64_t)h->n, \
        (const int32_t*)nullptr, (int64_t)h->n, out)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_get_state_idx(tq_env* h, const int32_t* idx, int n_idx, uint8_t* out, void* stream_) {
    HANDLE(h);
    if (n_idx < 0 || (n_idx > 0 && (!idx || !out))) return fail(TQ_E_INVALID, "bad idx / out");
    if (n_idx == 0) return TQ_OK;
    const int64_t total = (int64_t)n_idx * 2 * h->d * h->d;
#define CALL(D) hipLaunchKernelGGL(tq::k_get_state<D>, grid1(total, 256), dim3(256), 0, stream, h->planes, (int64_t)h->n, \
        idx, (int64_t)n_idx, out)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_get_qubits(tq_env* h, uint8_t* out, void* stream_) {
    HANDLE(h);
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
    const int64_t total = (int64_t)h->n * 2 * h->d * h->d;
#define CALL(D) hipLaunchKernelGGL(tq::k_get_qubits<D>, grid1(total, 256), dim3(256), 0, stream, h->planes, (int64_t)h->n, out)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_set_qubits(tq_env* h, const uint8_t* qubits, void* stream_) {
    HANDLE(h);
    if (!qubits) return fail(TQ_E_INVALID, "qubits is NULL");
#define CALL(D) hipLaunchKernelGGL(tq::k_set_qubits<D>, grid1(h->n, 256), dim3(256), 0, stream, h->planes, h->counts, qubits, (int64_t)h->n, h->partial)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    h->partial_valid = true;
    return TQ_OK;
}

int tq_get_counters(tq_env* h, uint32_t* episodes, uint32_t* steps, void* stream_) {
    HANDLE(h);
    if (episodes) HIPCHECK(hipMemcpyAsync(episodes, h->episodes, 4 * (size_t)h->n, hipMemcpyDeviceToDevice, stream));
    if (steps) HIPCHECK(hipMemcpyAsync(steps, h->steps, 4 * (size_t)h->n, hipMemcpyDeviceToDevice, stream));
    return TQ_OK;
}

int tq_eval_ground_state(tq_env* h, uint8_t* out, void* stream_) {
    HANDLE(h);
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
#define CALL(D) hipLaunchKernelGGL(tq::k_flags<D>, grid1(h->n, 256), dim3(256), 0, stream, h->planes, (int64_t)h->n, out, (uint8_t*)nullptr)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_is_terminal(tq_env* h, uint8_t* out, void* stream_) {
    HANDLE(h);
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
#define CALL(D) hipLaunchKernelGGL(tq::k_flags<D>, grid1(h->n, 256), dim3(256), 0, stream, h->planes, (int64_t)h->n, (uint8_t*)nullptr, out)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_persp_count(tq_env* h, int32_t* counts, int64_t* offsets, void* stream_) {
    HANDLE(h);
    if (!offsets) return fail(TQ_E_INVALID, "offsets is NULL");
    REQUIRE_ALIGNED16(offsets, "offsets");
    REQUIRE_ALIGNED16(counts, "counts");
    const int k = h->split_last ^ 1;                         // not the table the previous scan wrote: a stack write may still read it
    if (int rc = launch_scan(h->counts, h->partial, h->partial_valid, offsets, counts, h->n, stream, h->split[k])) return rc;
    h->partial_valid = true;
    h->split_for[k] = offsets;
    h->split_last = k;
    return TQ_OK;
}

int tq_persp_write_range(tq_env* h, const int64_t* offsets, int first, int count, void* out, int32_t* positions,
                         int64_t capacity, int dtype, void* stream_) {
    HANDLE(h);
    if (!offsets || !out) return fail(TQ_E_INVALID, "offsets / out is NULL");
    if (capacity < 0) return fail(TQ_E_INVALID, "negative capacity");
    if (first < 0 || count < 0 || (int64_t)first + count > h->n) return fail(TQ_E_INVALID, "lattice range outside [0, n_envs)");
    REQUIRE_ALIGNED16(out, "out");
    REQUIRE_ALIGNED16(positions, "positions");
    const uint64_t* vp = h->planes + (size_t)tq::PL_V * h->w * h->n;
    // the cut points of the whole batch came with the scan of these very offsets; for a lattice sub-range, or offsets
    // from elsewhere, the workgroups find theirs themselves
    const int32_t* split = nullptr;
    if (first == 0 && count == h->n) {
        if (offsets == h->split_for[h->split_last]) split = h->split[h->split_last];
        else if (offsets == h->split_for[h->split_last ^ 1]) split = h->split[h->split_last ^ 1];
    }
    unsigned int* slots = h->slots + tq::STREAM_SLOT_WORDS * (h->write_seq++ % N_SLOT_SETS);
#define CALL(D) if (int rc = launch_persp_write<D>(vp, h->n, offsets, out, positions, capacity, dtype, h->err, stream, first, count, \
        split, slots, h->xcd_bias >= 0 ? h->xcd_bias : xcd_bias())) return rc
    DISPATCH_D(h->d, CALL)
#undef CALL
    return TQ_OK;
}

int tq_persp_write(tq_env* h, const int64_t* offsets, void* out, int32_t* positions, int64_t capacity,
                   int dtype, void* stream_) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    return tq_persp_write_range(h, offsets, 0, h->n, out, positions, capacity, dtype, stream_);
}

// ---- stateless variants (states outside a handle) -------------------------------------------
static size_t states_scratch_bytes(int d, int64_t n) {
    const size_t w = (size_t)(d * d + 63) / 64;
    const size_t cnt_bytes = (((size_t)n * 4 + 32 + 15) & ~(size_t)15);
    const size_t part_bytes = (((size_t)n + tq::PART_BLOCK - 1) / tq::PART_BLOCK) * 8;
    return 2 * w * (size_t)n * 8 + cnt_bytes + part_bytes;
}

// set-up call: allocates (and synchronises); the tq_states_persp_* calls themselves never allocate
int tq_states_reserve(int d, int n_max) {
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (n_max <= 0) return fail(TQ_E_INVALID, "n_max must be > 0");
    int dev;
    if (int rc = current_device(&dev)) return rc;
    const uint16_t* lut_unused;
    if (int rc = get_lut(dev, d, nullptr, &lut_unused)) return rc;
    DeviceCtx& c = g_ctx[dev];
    std::lock_guard<std::mutex> lock(c.mu);
    const size_t need = states_scratch_bytes(d, n_max);
    if (c.ws_bytes < need) {
        HIPCHECK(hipDeviceSynchronize());                    // work queued on the old scratch must finish first
        if (c.ws) HIPCHECK(hipFree(c.ws));
        c.ws = nullptr; c.ws_bytes = 0;
        HIPCHECK(hipMalloc(&c.ws, need));
        HIPCHECK(hipMemset(c.ws, 0, need));
        c.ws_bytes = need;
    }
    return TQ_OK;
}

static int states_scratch(int dev, int d, int n, uint64_t** vp, int32_t** counts, int64_t** partial, int** err) {
    DeviceCtx& c = g_ctx[dev];
    std::lock_guard<std::mutex> lock(c.mu);
    const size_t w = (size_t)(d * d + 63) / 64;
    const size_t cnt_bytes = (((size_t)n * 4 + 32 + 15) & ~(size_t)15);
    if (c.ws_bytes < states_scratch_bytes(d, n))
        return fail(TQ_E_CAPACITY, "stateless scratch too small for %d states of d=%d: call tq_states_reserve(d, n_max) first", n, d);
    *vp = (uint64_t*)c.ws;
    *counts = (int32_t*)((char*)c.ws + 2 * w * (size_t)n * 8);
    *partial = (int64_t*)((char*)c.ws + 2 * w * (size_t)n * 8 + cnt_bytes);
    *err = c.err;
    return TQ_OK;
}

int tq_states_persp_count(int d, int n, const uint8_t* states, int32_t* counts, int64_t* offsets, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (n <= 0 || !states || !offsets) return fail(TQ_E_INVALID, "bad n / states / offsets");
    REQUIRE_ALIGNED16(offsets, "offsets");
    REQUIRE_ALIGNED16(counts, "counts");
    int dev;
    if (int rc = current_device(&dev)) return rc;
    const uint16_t* lut_unused;
    if (int rc = get_lut(dev, d, stream, &lut_unused)) return rc;
    uint64_t* vp; int32_t* cnt; int64_t* part; int* err;
    if (int rc = states_scratch(dev, d, n, &vp, &cnt, &part, &err)) return rc;
#define CALL(D) hipLaunchKernelGGL(tq::k_pack_states<D>, grid1(n, 256), dim3(256), 0, stream, states, vp, cnt, (int64_t)n)
    DISPATCH_D(d, CALL)
#undef CALL
    KCHECK();
    // no cut-point table for the stateless path: a per-device table would be shared by every caller and stream of the
    // device; the workgroups of tq_states_persp_write find their cut points themselves (find_cut, a few microseconds)
    if (int rc = launch_scan(cnt, part, false, offsets, counts, n, stream, nullptr)) return rc;
    return TQ_OK;
}

int tq_states_persp_write(int d, int n, const uint8_t* states, const int64_t* offsets, void* out,
                          int32_t* positions, int64_t capacity, int dtype, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (n <= 0 || !states || !offsets || !out || capacity < 0) return fail(TQ_E_INVALID, "bad arguments");
    REQUIRE_ALIGNED16(out, "out");
    REQUIRE_ALIGNED16(positions, "positions");
    int dev;
    if (int rc = current_device(&dev)) return rc;
    const uint16_t* lut;
    if (int rc = get_lut(dev, d, stream, &lut)) return rc;
    uint64_t* vp; int32_t* cnt; int64_t* part; int* err;
    if (int rc = states_scratch(dev, d, n, &vp, &cnt, &part, &err)) return rc;
#define CALL(D) hipLaunchKernelGGL(tq::k_pack_states<D>, grid1(n, 256), dim3(256), 0, stream, states, vp, (int32_t*)nullptr, (int64_t)n)
    DISPATCH_D(d, CALL)
#undef CALL
    KCHECK();
#define CALL(D) if (int rc = launch_persp_write<D>(vp, n, offsets, out, positions, capacity, dtype, err, stream, 0, n, nullptr)) return rc
    DISPATCH_D(d, CALL)
#undef CALL
    return TQ_OK;
}

int tq_states_transition(int d, int n, const uint8_t* states, const uint8_t* next_states, const int32_t* actions,
                         uint8_t* persp, uint8_t* next_persp, int32_t* actions_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (n <= 0 || !actions || (persp && !states) || (next_persp && !next_states)) return fail(TQ_E_INVALID, "bad arguments");
    REQUIRE_ALIGNED16(actions, "actions");
    REQUIRE_ALIGNED16(actions_out, "actions_out");
    int dev;
    if (int rc = current_device(&dev)) return rc;
    const uint16_t* lut;
    if (int rc = get_lut(dev, d, stream, &lut)) return rc;
    int* err = g_ctx[dev].err;
    const int64_t total = (int64_t)n * 2 * d * d;
#define CALL(D) hipLaunchKernelGGL(tq::k_states_transition<D>, grid1(total, 256), dim3(256), 0, stream, states, next_states, \
        actions, persp, next_persp, actions_out, lut, (int64_t)n, err)
    DISPATCH_D(d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_states_check(void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int dev;
    if (int rc = current_device(&dev)) return rc;
    int* latch = g_ctx[dev].err;
    if (!latch) return TQ_OK;                                // no stateless call has run on this device yet
    int flag = 0;
    HIPCHECK(hipMemcpyAsync(&flag, latch, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    if (flag) HIPCHECK(hipMemsetAsync(latch, 0, sizeof(int), stream));
    return decode_latch(flag);
}

int tq_select_action(tq_env* h, const float* q_table, const int64_t* offsets, const int32_t* positions,
                     const double* eps, int32_t* actions, float* q_values, void* stream_) {
    HANDLE(h);
    if (!offsets || !positions || !actions) return fail(TQ_E_INVALID, "offsets / positions / actions is NULL");
    if (q_table && !eps) return fail(TQ_E_INVALID, "eps is NULL");
#define CALL(D) hipLaunchKernelGGL(tq::k_select<D>, grid1((int64_t)h->n * 64, 256), dim3(256), 0, stream, q_table, offsets, \
        positions, eps, h->episodes, h->steps, 0u, 0u, (uint32_t)tq::DOMAIN_SEL, actions, q_values, h->seed, h->first_env, \
        (int64_t)h->n)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_states_select_action(int n, const float* q_table, const int64_t* offsets, const int32_t* positions,
                            const double* eps, uint64_t seed, uint64_t call_counter, int64_t first_id,
                            int32_t* actions, float* q_values, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || !offsets || !positions || !actions) return fail(TQ_E_INVALID, "bad n / offsets / positions / actions");
    if (q_table && !eps) return fail(TQ_E_INVALID, "eps is NULL");
    if (first_id < 0 || first_id + n > 0xFFFFFFFFll) return fail(TQ_E_INVALID, "state ids must fit in 32 bits");
    // the kernel does not depend on the lattice size (positions carry the coordinates)
    hipLaunchKernelGGL(tq::k_select<3>, grid1((int64_t)n * 64, 256), dim3(256), 0, stream, q_table, offsets, positions, eps,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t)call_counter,
                       (uint32_t)(call_counter >> 32), (uint32_t)tq::DOMAIN_SEL_CALL, actions, q_values, seed, first_id,
                       (int64_t)n);
    KCHECK();
    return TQ_OK;
}

int tq_segment_max(const float* q_table, const int64_t* offsets, int n, const int32_t* largest, float* out,
                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || !offsets || !out) return fail(TQ_E_INVALID, "bad n / offsets / out");
    hipLaunchKernelGGL(tq::k_segment_max, grid1((int64_t)n * 64, 256), dim3(256), 0, stream, q_table, offsets, largest,
                       out, (int64_t)n);
    KCHECK();
    return TQ_OK;
}

int64_t tq_transition_block_bytes(int d, int64_t cap) {
    if (!size_ok(d) || cap < 0) return -1;
    return tq::block_bytes((d * d + 63) / 64, cap);
}

// the packed block this goes through is the handle's own scratch (allocated by tq_create)
int tq_transition_write(tq_env* h, const int32_t* actions, uint8_t* persp, uint8_t* next_persp,
                        int32_t* actions_out, void* stream_) {
    HANDLE(h);
    if (!actions) return fail(TQ_E_INVALID, "actions is NULL");
    REQUIRE_ALIGNED16(actions, "actions");
    REQUIRE_ALIGNED16(actions_out, "actions_out");
    tq::BlockView b = tq::block_view(h->tblock, h->w, h->n);
#define CALL(D) hipLaunchKernelGGL(tq::k_transition<D>, grid1(h->n, 256), dim3(256), 0, stream, h->planes, h->prev, actions, \
        b, (int64_t)0, (int64_t)h->n, h->err)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    return tq_transition_unpack(h->d, h->tblock, h->n, 0, h->n, persp, next_persp, actions_out, nullptr, nullptr, nullptr,
                                stream_);
}

int tq_block_priorities(int d, void* block, int64_t cap, int n_envs, int n_steps, const float* q_values,
                        double discount, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (!block || n_envs <= 0 || n_steps <= 0 || (int64_t)n_envs * n_steps > cap)
        return fail(TQ_E_INVALID, "bad block / n_envs / n_steps (n_envs * n_steps must be <= cap)");
    tq::BlockView b = tq::block_view(block, (d * d + 63) / 64, cap);
    hipLaunchKernelGGL(tq::k_block_priorities, grid1((int64_t)n_envs * n_steps, 256), dim3(256), 0, stream, b,
                       (int64_t)n_envs, (int64_t)n_steps, q_values, discount);
    KCHECK();
    return TQ_OK;
}

int tq_transition_unpack(int d, const void* block, int64_t cap, int64_t first, int64_t count, uint8_t* persp,
                         uint8_t* next_persp, int32_t* actions, float* rewards, uint8_t* terminals, float* priorities,
                         void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!size_ok(d)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (!block || first < 0 || count < 0 || first + count > cap) return fail(TQ_E_INVALID, "bad block / slot range");
    if (count == 0) return TQ_OK;
    REQUIRE_ALIGNED16(actions, "actions");
    tq::BlockView b = tq::block_view(const_cast<void*>(block), (d * d + 63) / 64, cap);
    const int64_t total = count * 2 * d * d;
#define CALL(D) hipLaunchKernelGGL(tq::k_block_unpack<D>, grid1(total, 256), dim3(256), 0, stream, b, first, count, persp, \
        next_persp, actions, rewards, terminals, priorities)
    DISPATCH_D(d, CALL)
#undef CALL
    KCHECK();
    return TQ_OK;
}

int tq_actor_step(tq_env* h, const int32_t* actions, int32_t* actions_out, float* rewards, uint8_t* terminals,
                  void* block, int64_t block_cap, int64_t slot_base, void* stream_) {
    HANDLE(h);
    REQUIRE_ALIGNED16(actions, "actions");
    REQUIRE_ALIGNED16(actions_out, "actions_out");
    if (block && (reinterpret_cast<uintptr_t>(block) & 7u)) return fail(TQ_E_INVALID, "block must be 8-byte aligned");
    tq::BlockView b;
    memset(&b, 0, sizeof(b));
    if (block) {
        if (slot_base < 0 || slot_base + h->n > block_cap) return fail(TQ_E_CAPACITY, "transition block too small");
        b = tq::block_view(block, h->w, block_cap);
    }
#define CALL(D) hipLaunchKernelGGL(tq::k_actor_step<D>, grid1(h->n, 256), dim3(256), 0, stream, (const uint64_t*)h->planes, h->planes_alt, h->episodes, \
        h->steps, h->counts, h->p_roof, actions, actions_out, rewards, terminals, b, block ? 1 : 0, slot_base, h->sched, \
        (float)h->terminal_reward, h->max_steps, h->min_err, h->seed, h->first_env, (int64_t)h->n, h->err, h->partial)
    DISPATCH_D(h->d, CALL)
#undef CALL
    KCHECK();
    { uint64_t* t = h->planes; h->planes = h->planes_alt; h->planes_alt = t; }   // the buffer just written holds the lattices now
    h->partial_valid = true;
    return TQ_OK;
}

int tq_check(tq_env* h, void* stream_) {
    HANDLE(h);
    int flag = 0;
    HIPCHECK(hipMemcpyAsync(&flag, h->err, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    if (flag) HIPCHECK(hipMemsetAsync(h->err, 0, sizeof(int), stream));
    return decode_latch(flag);
}

}  // extern "C"
