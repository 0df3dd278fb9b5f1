// Single-precision arc cosine that returns the same floats as the host C library.
//
// The reference computes the v coordinate of a sphere hit with acosf (src/shapes.cpp:66; the unqualified acos of
// its neighbours resolve to the double functions, this one is the float function).  glibc's acosf
// (sysdeps/ieee754/flt-32/e_acosf.c, the float version of Sun's fdlibm routine) is within one ulp but not correctly
// rounded, so acos in double rounded to float differs from it for a few percent of the arguments - enough to move a
// texel now and then.  This is that routine restated: three ranges, a degree-6 over degree-4 rational approximation
// of ( asin( x ) - x ) / x^3 in single precision, separate multiplies and adds (the library's x86-64 build has no
// fused variant of it).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined( __HIPCC__ ) || defined( __CUDACC__ )
#define PT_ACOSF_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define PT_ACOSF_HD inline
#endif

namespace ptd
{

PT_ACOSF_HD float AcosfRational( float z )
{
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const float p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
    const float q = 1.0f + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
    return p / q;
}

PT_ACOSF_HD float AcosLibm( float x )
{
    const float pi = 3.1415925026e+00f, pio2Hi = 1.5707962513e+00f, pio2Lo = 7.5497894159e-08f;
    uint32_t bits;
    memcpy( &bits, &x, 4 );
    const int32_t hx = static_cast< int32_t >( bits );
    const int32_t ix = hx & 0x7fffffff;
    if ( ix == 0x3f800000 ) return hx > 0 ? 0.0f : pi + 2.0f * pio2Lo; // acos( 1 ) = 0, acos( -1 ) = pi
    if ( ix > 0x3f800000 ) return ( x - x ) / ( x - x );                // |x| > 1 or NaN
    if ( ix < 0x3f000000 )                                              // |x| < 0.5
    {
        if ( ix <= 0x32800000 ) return pio2Hi + pio2Lo; // |x| <= 2^-26
        const float r = AcosfRational( x * x );
        return pio2Hi - ( x - ( pio2Lo - r * x ) );
    }
    if ( hx < 0 ) // x < -0.5
    {
        const float z = ( 1.0f + x ) * 0.5f;
        const float r = AcosfRational( z );
        const float s = sqrtf( z );
        const float w = r * s - pio2Lo;
        return pi - 2.0f * ( s + w );
    }
    // x > 0.5
    const float z = ( 1.0f - x ) * 0.5f;
    const float s = sqrtf( z );
    uint32_t sb;
    memcpy( &sb, &s, 4 );
    sb &= 0xfffff000u;
    float df;
    memcpy( &df, &sb, 4 );
    const float c = ( z - df * df ) / ( s + df );
    const float r = AcosfRational( z );
    const float w = r * s + c;
    return 2.0f * ( df + w );
}

} // namespace ptd
