// common.h -- shared host/device helpers of the MI355X seed-and-extend core (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/nvbio_amd.h"

namespace nvbio_amd {

// ---- error handling: status codes + a thread-local message, never exceptions ----------------
void        set_error(const char* fmt, ...);
const char* get_error();

#define NVB_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            nvbio_amd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return NVBIO_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

#define NVB_CHECK(call)                                                                            \
    do { nvbio_status _s = (call); if (_s != NVBIO_OK) return _s; } while (0)

#define NVB_REQUIRE(cond, msg)                                                                     \
    do { if (!(cond)) { nvbio_amd::set_error("invalid argument: %s", msg); return NVBIO_ERR_INVALID; } } while (0)

// Scratch with stream-ordered semantics WITHOUT the runtime's stream-ordered pool: a block handed out by scratch_alloc may be used by work enqueued on
// `s` after the call; scratch_free gives it back at once, for the next call on the SAME stream (whose work runs behind everything that used the block).
// Blocks are plain hipMalloc allocations cached per (device, stream).  Why not hipMallocAsync / hipFreeAsync: on ROCm 7.2 a block RE-used from the
// default pool did not hold what a kernel had just written into it -- the flags of the banded scorer's first pass read back as zeros in the second
// call of a process (seen with hipMemcpy right behind the kernel; with hipMalloc / hipFree in its place: never), and once in a few fresh-box runs the
// gap chance read zeros there in production (61,818 reads of 10 M scored -20 instead of -18).  core.hip.
hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t s);
void       scratch_free(void* p, hipStream_t s);
void       scratch_release_idle();                        // hipFree every idle block (each behind a synchronisation of its stream)

// select the device and fail loudly if it is not a gfx950: there is no CPU fallback
nvbio_status use_device(int device);

// RAII device switch that restores the caller's current device
struct DeviceGuard
{
    int  prev;
    bool ok;
    explicit DeviceGuard(int device) : prev(-1), ok(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (use_device(device) == NVBIO_OK);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// launch-grid helper: enough 256-thread workgroups for n items, capped so that the grid stays
// a small multiple of the chip (256 CUs x 8 blocks) and the kernels grid-stride the rest
static inline unsigned grid_for(uint64_t n, unsigned block = 256, unsigned cap_blocks = 256u * 32u)
{
    uint64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap_blocks) b = cap_blocks;
    return (unsigned)b;
}

// ---- device-side symbol access --------------------------------------------------------------
// Big-endian packed streams (PackedStream<..,BITS,true>): symbol i of a 2-bit stream sits at bits
// [30-2(i&15), 31-2(i&15)] of word i>>4; of a 4-bit stream at [28-4(i&7), 31-4(i&7)] of word i>>3.
// The reader keeps the last word in a register so that a scan over consecutive symbols issues
// one load per 16 (8) symbols.
template <int BITS>
struct SymbolReader
{
    const uint32_t* words;
    uint32_t        cur_idx;
    uint32_t        cur;
    __device__ __forceinline__ explicit SymbolReader(const void* p) : words((const uint32_t*)p), cur_idx(0xFFFFFFFFu), cur(0) {}
    __device__ __forceinline__ uint32_t get(uint32_t i)
    {
        constexpr uint32_t LOG = (BITS == 2) ? 4 : 3;
        constexpr uint32_t PER = 1u << LOG;
        const uint32_t w = i >> LOG;
        if (w != cur_idx) { cur = words[w]; cur_idx = w; }
        return (cur >> ((32u - BITS) - BITS * (i & (PER - 1u)))) & ((1u << BITS) - 1u);
    }
};
template <>
struct SymbolReader<8>
{
    const uint8_t* bytes;
    __device__ __forceinline__ explicit SymbolReader(const void* p) : bytes((const uint8_t*)p) {}
    __device__ __forceinline__ uint32_t get(uint32_t i) { return bytes[i]; }
};

} // namespace nvbio_amd
