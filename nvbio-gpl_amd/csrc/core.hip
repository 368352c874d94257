// core.hip -- status/error plumbing and device selection of the C ABI.
#include "common.h"
#include <mutex>
#include <vector>

namespace nvbio_amd {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start( ap, fmt );
    vsnprintf( g_error, sizeof(g_error), fmt, ap );
    va_end( ap );
}
const char* get_error() { return g_error; }

// ---- scratch blocks cached per (device, stream): see common.h ----
namespace {
struct ScratchBlock { void* p; size_t cap; bool busy; int device; hipStream_t stream; };
std::mutex                 g_scratch_mutex;
std::vector<ScratchBlock>  g_scratch;
const size_t SCRATCH_KEEP_PER_STREAM = 32ull << 30;             // idle blocks beyond this are released (after the stream has drained)
}

hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t s)
{
    *p = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice( &dev );
    if (e != hipSuccess) return e;
    if (bytes == 0) bytes = 256;
    const size_t want = (bytes + ((2u << 20) - 1u)) & ~(size_t)((2u << 20) - 1u);
    std::lock_guard<std::mutex> lock( g_scratch_mutex );
    // best fit among this stream's idle blocks (any larger block will do: a loop whose requests shrink from pass to pass must not allocate in every pass)
    int best = -1;
    for (size_t i = 0; i < g_scratch.size(); ++i)
    {
        const ScratchBlock& b = g_scratch[i];
        if (!b.busy && b.device == dev && b.stream == s && b.cap >= want && (best < 0 || b.cap < g_scratch[best].cap)) best = (int)i;
    }
    if (best >= 0) { g_scratch[best].busy = true; *p = g_scratch[best].p; return hipSuccess; }
    void* q = nullptr;
    e = hipMalloc( &q, want );
    if (e != hipSuccess)
    {
        // out of memory: give this stream's idle blocks back (its work may still use them: drain it first) and try once more
        (void)hipGetLastError();
        (void)hipStreamSynchronize( s );
        for (size_t i = 0; i < g_scratch.size(); )
        {
            if (!g_scratch[i].busy && g_scratch[i].device == dev) { (void)hipFree( g_scratch[i].p ); g_scratch.erase( g_scratch.begin() + i ); }
            else ++i;
        }
        e = hipMalloc( &q, want );
        if (e != hipSuccess) return e;
    }
    g_scratch.push_back( ScratchBlock{ q, want, true, dev, s } );
    *p = q;
    return hipSuccess;
}

void scratch_release_idle()
{
    std::lock_guard<std::mutex> lock( g_scratch_mutex );
    int prev = -1; (void)hipGetDevice( &prev );
    for (size_t i = 0; i < g_scratch.size(); )
    {
        if (!g_scratch[i].busy)
        {
            (void)hipSetDevice( g_scratch[i].device );
            (void)hipStreamSynchronize( g_scratch[i].stream );   // work behind which the block was given back may still use it
            (void)hipFree( g_scratch[i].p );
            g_scratch.erase( g_scratch.begin() + i );
        }
        else ++i;
    }
    if (prev >= 0) (void)hipSetDevice( prev );
}

void scratch_free(void* p, hipStream_t s)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock( g_scratch_mutex );
    size_t idle = 0; int dev = -1;
    for (ScratchBlock& b : g_scratch)
        if (b.p == p) { b.busy = false; dev = b.device; }
    for (const ScratchBlock& b : g_scratch) if (!b.busy && b.device == dev && b.stream == s) idle += b.cap;
    if (idle > SCRATCH_KEEP_PER_STREAM)
    {
        (void)hipStreamSynchronize( s );                          // work behind which the idle blocks were given back may still read them
        for (size_t i = 0; i < g_scratch.size(); )
        {
            if (!g_scratch[i].busy && g_scratch[i].device == dev && g_scratch[i].stream == s) { (void)hipFree( g_scratch[i].p ); g_scratch.erase( g_scratch.begin() + i ); }
            else ++i;
        }
    }
}

nvbio_status use_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount( &count ) != hipSuccess || count <= 0)
    {
        set_error( "no HIP device is visible: this library has no CPU fallback" );
        return NVBIO_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count)
    {
        set_error( "device %d out of range (%d visible)", device, count );
        return NVBIO_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties( &prop, device ) != hipSuccess)
    {
        set_error( "hipGetDeviceProperties(%d) failed", device );
        return NVBIO_ERR_HIP;
    }
    if (strncmp( prop.gcnArchName, "gfx950", 6 ) != 0)
    {
        set_error( "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName );
        return NVBIO_ERR_NO_DEVICE;
    }
    if (hipSetDevice( device ) != hipSuccess)
    {
        set_error( "hipSetDevice(%d) failed", device );
        return NVBIO_ERR_HIP;
    }
    // (scratch -- boundary columns, direction vectors, scan temporaries -- comes from scratch_alloc above, not from the runtime's stream-ordered pool)
    return NVBIO_OK;
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" {

int         nvbio_amd_version(void)    { return NVBIO_AMD_VERSION; }
nvbio_status nvbio_amd_release_scratch(void) { scratch_release_idle(); return NVBIO_OK; }
const char* nvbio_amd_last_error(void) { return get_error(); }

nvbio_status nvbio_amd_device_count(int* count)
{
    NVB_REQUIRE( count != nullptr, "count is NULL" );
    *count = 0;
    if (hipGetDeviceCount( count ) != hipSuccess) { *count = 0; set_error( "hipGetDeviceCount failed" ); return NVBIO_ERR_NO_DEVICE; }
    return NVBIO_OK;
}

nvbio_status nvbio_amd_device_arch(int device, char* name, uint32_t name_len)
{
    NVB_REQUIRE( name != nullptr && name_len > 0, "name buffer is NULL" );
    hipDeviceProp_t prop;
    NVB_HIP( hipGetDeviceProperties( &prop, device ) );
    strncpy( name, prop.gcnArchName, name_len - 1 );
    name[name_len - 1] = 0;
    return NVBIO_OK;
}

nvbio_status nvbio_amd_stream_synchronize(int device, void* stream)
{
    DeviceGuard g( device ); if (!g.ok) return NVBIO_ERR_NO_DEVICE;
    NVB_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
    return NVBIO_OK;
}

} // extern "C"
