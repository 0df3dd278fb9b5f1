// Single-precision sine and cosine that return the same floats as the host C library.
//
// The reference obtains the bounce and light-sample directions from std::cos / std::sin of a float
// (src/sampling.cpp:37,45,60), i.e. glibc's sincosf.  That routine is accurate to well under one ulp but
// not correctly rounded, and the device math library rounds differently in roughly one call in 500.  In the
// reference's Cornell scene a one-ulp change of a path vertex flips the outcome of later shadow rays (they
// are aimed exactly at the emitter with tMax = distance, src/lights.cpp:45-49), so a deterministic
// comparison against the CPU needs the very same floats.
//
// This is the algorithm glibc uses since 2.28 (sysdeps/ieee754/flt-32/s_sincosf.c, from ARM's
// optimized-routines, S. Nagy): evaluate in double precision; |x| < pi/4 directly, otherwise reduce by the
// nearest multiple n of pi/2 found with a 2^24-prescaled 2/pi, flip signs by quadrant and swap the two
// polynomials for odd n.  The polynomial coefficients are the published minimax ones.  Whether the double
// operations are fused or not moves the result by about 1e-16 relative, which changes the rounded float in
// about one call per billion.  Arguments of 120 and more (never produced by the sampling warps) fall back to
// the double-precision library functions.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined( __HIPCC__ ) || defined( __CUDACC__ )
#define PT_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define PT_HD inline
#endif

namespace ptd
{

struct SinCosTable
{
    double sign[4];
    double hpiInv; // 2 / pi * 2^24
    double hpi;    // pi / 2
    double c0, c1, c2, c3, c4;
    double s1, s2, s3;
};

PT_HD uint32_t AbsTop12( float x )
{
    uint32_t u;
    memcpy( &u, &x, 4 );
    return ( u >> 20 ) & 0x7ffu;
}

// The host library's variant for CPUs with FMA is built with contraction enabled; allow it here as well.
PT_HD void SinCosPoly( double x, double x2, double cs, int n, float* sinp, float* cosp )
{
#if defined( __clang__ )
#pragma clang fp contract( fast )
#endif
    // cs = +1 selects the table for quadrants 0 and 1, -1 the negated cosine polynomial of quadrants 2 and 3
    const double c0 = cs * 0x1p0, c1 = cs * -0x1.ffffffd0c621cp-2, c2 = cs * 0x1.55553e1068f19p-5,
                 c3 = cs * -0x1.6c087e89a359dp-10, c4 = cs * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    double x4  = x2 * x2;
    double x3  = x2 * x;
    double pc2 = c3 + x2 * c4;
    double ps1 = s2 + x2 * s3;
    double pc1 = c0 + x2 * c1;
    double x5  = x3 * x2;
    double x6  = x4 * x2;
    double s   = x + x3 * s1;
    double c   = pc1 + x4 * c2;
    float sv   = static_cast< float >( s + x5 * ps1 );
    float cv   = static_cast< float >( c + x6 * pc2 );
    if ( n & 1 )
    {
        *sinp = cv;
        *cosp = sv;
    }
    else
    {
        *sinp = sv;
        *cosp = cv;
    }
}

PT_HD void SinCosLibm( float y, float* sinp, float* cosp )
{
#if defined( __clang__ )
#pragma clang fp contract( fast )
#endif
    double x = y;
    if ( AbsTop12( y ) < AbsTop12( 0x1.921FB6p-1f ) ) // |y| < pi / 4
    {
        double x2 = x * x;
        if ( AbsTop12( y ) < AbsTop12( 0x1p-12f ) )
        {
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
        SinCosPoly( x, x2, 1.0, 0, sinp, cosp );
    }
    else if ( AbsTop12( y ) < AbsTop12( 120.0f ) )
    {
        double r  = x * 0x1.45F306DC9C883p+23;
        int n     = ( static_cast< int32_t >( r ) + 0x800000 ) >> 24;
        x         = x - n * 0x1.921FB54442D18p0;
        const double sign = ( ( n & 3 ) == 1 || ( n & 3 ) == 2 ) ? -1.0 : 1.0; // { 1, -1, -1, 1 }
        SinCosPoly( x * sign, x * x, ( n & 2 ) ? -1.0 : 1.0, n, sinp, cosp );
    }
    else
    {
        *sinp = static_cast< float >( sin( x ) );
        *cosp = static_cast< float >( cos( x ) );
    }
}

} // namespace ptd
