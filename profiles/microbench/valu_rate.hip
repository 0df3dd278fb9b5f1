// What a SIMD of gfx950 issues per cycle, by instruction kind and waves per SIMD: times loops of independent
// v_fma_f32 / v_pk_fma_f32 / v_cvt_f32_ubyte / v_max3_f32 / v_cndmask and prints wave-instructions per cycle per SIMD
// (clock taken from s_memtime).  The traversal kernels' counter summaries divide SQ_ACTIVE_INST_VALU * 4 by the
// SIMD-cycles of the launch; this says what that quotient is at the hardware's limit.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__( ( ext_vector_type( 2 ) ) );

template < int kKind >
__global__ __launch_bounds__( 256 ) void Loop( float* out, int iters, unsigned long long* cycles )
{
    float a[8];
    f2 p[8];
    unsigned u[8];
    for ( int k = 0; k < 8; ++k )
    {
        a[k] = threadIdx.x * 0.001f + k;
        p[k] = f2{ a[k], a[k] + 0.5f };
        u[k] = threadIdx.x * 2654435761u + k;
    }
    const float m = 1.0001f, c = 0.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for ( int i = 0; i < iters; ++i )
    {
#pragma unroll
        for ( int r = 0; r < 4; ++r )
#pragma unroll
            for ( int k = 0; k < 8; ++k )
            {
                if constexpr ( kKind == 0 ) a[k] = __builtin_fmaf( a[k], m, c );
                if constexpr ( kKind == 1 ) p[k] = __builtin_elementwise_fma( p[k], f2{ m, m }, f2{ c, c } );
                if constexpr ( kKind == 2 )
                {
                    a[k] = static_cast< float >( ( u[k] >> 8 ) & 0xFFu );
                    asm volatile( "" : "+v"( a[k] ) );
                }
                if constexpr ( kKind == 3 ) a[k] = __builtin_fmaxf( __builtin_fmaxf( a[k], a[( k + 1 ) & 7] ), c );
                if constexpr ( kKind == 4 )
                {
                    u[k] = a[k] > c ? u[k] : u[( k + 1 ) & 7];
                    asm volatile( "" : "+v"( u[k] ) );
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for ( int k = 0; k < 8; ++k ) s += a[k] + p[k].x + p[k].y + u[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ( threadIdx.x == 0 ) cycles[blockIdx.x] = t1 - t0;
}

template < int kKind >
void Run( const char* name, int wavesPerSimd )
{
    const int blocks = 256 * wavesPerSimd, iters = 20000; // 256-thread blocks: 4 waves, one per SIMD; wavesPerSimd blocks per CU
    float* out;
    unsigned long long* cycles;
    hipMalloc( &out, sizeof( float ) * blocks * 256 );
    hipMalloc( &cycles, 8 * blocks );
    hipLaunchKernelGGL( Loop< kKind >, dim3( blocks ), dim3( 256 ), 0, 0, out, 100, cycles );
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate( &e0 );
    hipEventCreate( &e1 );
    hipEventRecord( e0 );
    hipLaunchKernelGGL( Loop< kKind >, dim3( blocks ), dim3( 256 ), 0, 0, out, iters, cycles );
    hipEventRecord( e1 );
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime( &ms, e0, e1 );
    std::vector< unsigned long long > h( blocks );
    hipMemcpy( h.data(), cycles, 8 * blocks, hipMemcpyDeviceToHost );
    double mean = 0;
    for ( auto v : h ) mean += double( v ) / blocks;
    const double instrPerWave = double( iters ) * 32;
    // s_memtime ticks at 100 MHz on this part; cycles from the elapsed time at the clock the chip reports are less direct,
    // so print both the wall-time rate and the tick count
    printf( "%-18s %d waves/SIMD: %.3f ms, %.2f G wave-instr/s per SIMD (%.3f instr per cycle at 2.4 GHz), memtime ticks per wave %.0f\n", name, wavesPerSimd, ms,
            instrPerWave * wavesPerSimd / ( ms * 1e-3 ) / 1e9, instrPerWave * wavesPerSimd / ( ms * 1e-3 ) / 2.4e9, mean );
    hipFree( out );
    hipFree( cycles );
}

int main()
{
    for ( int w : { 1, 2, 4, 8 } )
    {
        Run< 0 >( "v_fma_f32", w );
        Run< 1 >( "v_pk_fma_f32", w );
        Run< 2 >( "v_cvt_f32_ubyte1", w );
        Run< 3 >( "v_max3_f32", w );
        Run< 4 >( "v_cmp+v_cndmask", w );
    }
    return 0;
}
