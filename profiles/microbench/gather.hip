// Gather throughput of one MI355X as the traversal kernels see it: every lane of a wave loads from an address of
// its own.  Reports, per load width, number of active lanes per wave and footprint, the time one wave-level load
// instruction occupies a CU (cycles per instruction per CU at full occupancy) - which tells whether the memory
// pipeline charges a load by instruction, by active lane or by byte.
//   hipcc --offload-arch=gfx950 -O3 -o gather gather.hip && ./gather
// With arguments `width lanes footprint` it runs that one case (a warming launch, then the one that counts) and prints
// the 64-byte lines the second launch touched: for a profiler pass that counts what left the L2 (FETCH_SIZE), see
// profiles/calibrate_fetch_size.sh.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK( x ) do { hipError_t e = ( x ); if ( e != hipSuccess ) { std::printf( "%s: %s\n", #x, hipGetErrorString( e ) ); return 1; } } while ( 0 )

typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef uint32_t u32x2 __attribute__( ( ext_vector_type( 2 ) ) );

template < int kWidth >
__device__ __forceinline__ uint32_t Load( __amdgpu_buffer_rsrc_t rsrc, uint32_t offset )
{
    if constexpr ( kWidth == 16 )
    {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128( rsrc, static_cast< int >( offset ), 0, 0 );
        return v.x ^ v.y ^ v.z ^ v.w;
    }
    else if constexpr ( kWidth == 8 )
    {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64( rsrc, static_cast< int >( offset ), 0, 0 );
        return v.x ^ v.y;
    }
    else
        return static_cast< uint32_t >( __builtin_amdgcn_raw_buffer_load_b32( rsrc, static_cast< int >( offset ), 0, 0 ) );
}

// Every active lane walks its own pseudo-random sequence of 16-byte-aligned offsets below `bytes` (a power of two);
// four loads are in flight per lane.  activeLanes of every 64 take part (lanes spread evenly over the wave).
template < int kWidth >
__global__ __launch_bounds__( 1024 ) void Gather( const void* base, uint32_t bytes, int loadsPerLane, int activeLanes, int sharedBy, uint32_t* out )
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc( const_cast< void* >( base ), 0, static_cast< int >( bytes ), 0x00020000 );
    const uint32_t lane = threadIdx.x & 63u;
    const bool active = ( lane * activeLanes ) / 64u != ( ( lane + 1 ) * activeLanes ) / 64u;
    // lanes of a group of `sharedBy` consecutive active lanes read consecutive 16-byte pieces of one line
    const uint32_t id = ( blockIdx.x * 1024u + threadIdx.x ) / static_cast< uint32_t >( sharedBy );
    const uint32_t piece = ( threadIdx.x % static_cast< uint32_t >( sharedBy ) ) * 16u;
    uint32_t state = id * 2654435761u + 12345u, acc = 0;
    if ( active )
    {
        for ( int i = 0; i < loadsPerLane; i += 4 )
        {
            uint32_t o[4];
            for ( int k = 0; k < 4; ++k )
            {
                state = state * 1664525u + 1013904223u;
                o[k]  = ( ( state >> 4 ) * 64u * ( sharedBy > 4 ? 2u : 1u ) + piece ) & ( bytes - 1u ) & ~15u;
            }
            uint32_t v[4];
            for ( int k = 0; k < 4; ++k ) v[k] = Load< kWidth >( rsrc, o[k] );
            for ( int k = 0; k < 4; ++k ) acc ^= v[k];
        }
    }
    if ( acc == 0x12345678u ) out[0] = acc; // never true for the zero-filled buffer; keeps the loads alive
}

int main( int argc, char** argv )
{
    hipDeviceProp_t prop;
    CHECK( hipGetDeviceProperties( &prop, 0 ) );
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    std::printf( "%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz );
    const size_t maxBytes = 1ull << 30;
    void* buffer = nullptr;
    uint32_t* out = nullptr;
    CHECK( hipMalloc( &buffer, maxBytes ) );
    CHECK( hipMemset( buffer, 0, maxBytes ) );
    CHECK( hipMalloc( &out, 64 ) );
    hipEvent_t t0, t1;
    CHECK( hipEventCreate( &t0 ) );
    CHECK( hipEventCreate( &t1 ) );
    const int blocks = 2 * cus, loads = 2048;
    if ( argc == 4 )
    {
        const int width = std::atoi( argv[1] ), lanes = std::atoi( argv[2] );
        const uint32_t footprint = static_cast< uint32_t >( std::strtoul( argv[3], nullptr, 10 ) );
        for ( int rep = 0; rep < 2; ++rep )
        {
            if ( width == 16 ) hipLaunchKernelGGL( Gather< 16 >, dim3( blocks ), dim3( 1024 ), 0, 0, buffer, footprint, loads, lanes, 1, out );
            else if ( width == 8 ) hipLaunchKernelGGL( Gather< 8 >, dim3( blocks ), dim3( 1024 ), 0, 0, buffer, footprint, loads, lanes, 1, out );
            else hipLaunchKernelGGL( Gather< 4 >, dim3( blocks ), dim3( 1024 ), 0, 0, buffer, footprint, loads, lanes, 1, out );
            CHECK( hipDeviceSynchronize() );
        }
        // every load of an active lane touches one 64-byte line of its own choosing
        std::printf( "width %d lanes %d footprint %u: %.0f loads of one 64-byte line each per launch\n", width, lanes, footprint,
                     static_cast< double >( blocks ) * 16.0 * lanes * loads );
        return 0;
    }
    std::printf( "%-6s %-7s %-9s %-10s %12s %14s %12s\n", "width", "lanes", "share", "footprint", "ms", "cyc/instr/CU", "GB/s" );
    for ( uint32_t footprint : { 16u << 10, 2u << 20, 1u << 30 } )
        for ( int sharedBy : { 1, 2, 4 } )
            for ( int width : { 4, 8, 16 } )
                for ( int lanes : { 64, 32, 16, 8 } )
                {
                    if ( sharedBy > 1 && ( width != 16 || lanes != 64 ) ) continue;
                    for ( int rep = 0; rep < 2; ++rep ) // first pass warms the caches
                    {
                        CHECK( hipEventRecord( t0 ) );
                        if ( width == 16 ) hipLaunchKernelGGL( Gather< 16 >, dim3( blocks ), dim3( 1024 ), 0, 0, buffer, footprint, loads, lanes, sharedBy, out );
                        else if ( width == 8 ) hipLaunchKernelGGL( Gather< 8 >, dim3( blocks ), dim3( 1024 ), 0, 0, buffer, footprint, loads, lanes, sharedBy, out );
                        else hipLaunchKernelGGL( Gather< 4 >, dim3( blocks ), dim3( 1024 ), 0, 0, buffer, footprint, loads, lanes, sharedBy, out );
                        CHECK( hipEventRecord( t1 ) );
                        CHECK( hipEventSynchronize( t1 ) );
                    }
                    float ms = 0;
                    CHECK( hipEventElapsedTime( &ms, t0, t1 ) );
                    const double instrPerCu = static_cast< double >( blocks ) / cus * 16.0 * loads; // waves per CU x loads
                    const double cycles = ms * 1e-3 * ghz * 1e9 / instrPerCu;
                    const double gbs = static_cast< double >( blocks ) * 16.0 * lanes * loads * width / ( ms * 1e-3 ) * 1e-9;
                    std::printf( "%-6d %-7d %-9d %-10u %12.3f %14.1f %12.0f\n", width, lanes, sharedBy, footprint, ms, cycles, gbs );
                }
    return 0;
}
