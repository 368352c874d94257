// core.hip -- status/error plumbing and device selection of the C ABI.
#include "common.h"

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
    // Scratch (boundary columns, direction vectors, scan temporaries) comes from the device's stream-ordered
    // pool.  By default the pool hands freed blocks back to the driver at every synchronisation, which makes
    // each call pay for mapping gigabytes again (measured: 1.4 s per 16 GiB); let it keep up to 48 GiB (a paired-end
    // step holds a scoring column, a banded and a full-matrix direction scratch at once).
    static bool pool_ready[64] = { false };
    if (device < 64 && !pool_ready[device])
    {
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool( &pool, device ) == hipSuccess)
        {
            uint64_t keep = 48ull << 30;
            (void)hipMemPoolSetAttribute( pool, hipMemPoolAttrReleaseThreshold, &keep );
        }
        pool_ready[device] = true;
    }
    return NVBIO_OK;
}

} // namespace nvbio_amd

using namespace nvbio_amd;

extern "C" {

int         nvbio_amd_version(void)    { return NVBIO_AMD_VERSION; }
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
