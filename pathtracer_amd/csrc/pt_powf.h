// Single-precision pow that returns the same floats as the host C library.
//
// The reference's tonemap pass ends with glm::pow( pixel, 1 / gamma ) on floats (src/tonemap.cpp:6-9), i.e. the C
// library's powf, once per channel of every pixel.  glibc's powf is accurate to 0.82 ulp but not correctly
// rounded: pow evaluated in double and rounded once differs from it in the last bit for about one argument in
// 1,700, which is visible in the 8-bit picture file now and then.  Picture files identical to the reference's need
// the very same floats.
//
// This is the algorithm of glibc 2.28 and later (sysdeps/ieee754/flt-32/e_powf.c, from ARM's optimized-routines,
// Sz. Nagy): log2( x ) in double from a 16-entry table of ( 1 / c, log2( c ) ) and a degree-4 polynomial, times y,
// then 2^( y log2 x ) from a 32-entry table of 2^( i / 32 ) and a degree-3 polynomial, rounded to float once.  The
// tables and coefficients below are the published ones (e_powf_log2_data.c, e_exp2f_data.c), written as their
// bit patterns.  Every multiply-add of the two polynomials is FUSED here, because that is the variant an x86-64
// host with FMA runs (glibc selects __powf_fma there; with separate multiplies and adds the double result moves by
// about 1e-16 relative, which changes the rounded float about once in a billion calls).  Special cases follow the
// C standard's Annex F exactly as that file does; round-to-nearest is assumed.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined( __HIPCC__ ) || defined( __CUDACC__ )
#define PT_POWF_HD __host__ __device__ __forceinline__
#define PT_POWF_TABLE static __device__ const
#else
#include <math.h>
#define PT_POWF_HD inline
#define PT_POWF_TABLE static const
#endif

namespace ptd
{

// { 1 / c, log2( c ) } for c near the centre of each of 16 subintervals of [ 0x1.66p-1, 0x1.66p0 )
PT_POWF_TABLE uint64_t kPowfLog2Table[16][2] = {
    { 0x3ff661ec79f8f3beull, 0xbfdefec65b963019ull }, { 0x3ff571ed4aaf883dull, 0xbfdb0b6832d4fca4ull },
    { 0x3ff49539f0f010b0ull, 0xbfd7418b0a1fb77bull }, { 0x3ff3c995b0b80385ull, 0xbfd39de91a6dcf7bull },
    { 0x3ff30d190c8864a5ull, 0xbfd01d9bf3f2b631ull }, { 0x3ff25e227b0b8ea0ull, 0xbfc97c1d1b3b7af0ull },
    { 0x3ff1bb4a4a1a343full, 0xbfc2f9e393af3c9full }, { 0x3ff12358f08ae5baull, 0xbfb960cbbf788d5cull },
    { 0x3ff0953f419900a7ull, 0xbfaa6f9db6475fceull }, { 0x3ff0000000000000ull, 0x0000000000000000ull },
    { 0x3fee608cfd9a47acull, 0x3fb338ca9f24f53dull }, { 0x3feca4b31f026aa0ull, 0x3fc476a9543891baull },
    { 0x3feb2036576afce6ull, 0x3fce840b4ac4e4d2ull }, { 0x3fe9c2d163a1aa2dull, 0x3fd40645f0c6651cull },
    { 0x3fe886e6037841edull, 0x3fd88e9c2c1b9ff8ull }, { 0x3fe767dcf5534862ull, 0x3fdce0a44eb17bccull },
};
// log1p( r ) / ln 2 ~ r ( A4 + r ( A3 + r ( A2 + r ( A1 + r A0 ) ) ) )
PT_POWF_TABLE uint64_t kPowfLog2Poly[5] = { 0x3fd27616c9496e0bull, 0xbfd71969a075c67aull, 0x3fdec70a6ca7baddull, 0xbfe7154748bef6c8ull,
                                            0x3ff71547652ab82bull };
// 2^( i / 32 ) with i * 2^47 subtracted from the bit pattern, i = 0 .. 31
PT_POWF_TABLE uint64_t kExp2fTable[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
// 2^r ~ 1 + r ( C2 + r ( C1 + r C0 ) ) for |r| <= 1 / 64
PT_POWF_TABLE uint64_t kExp2fPoly[3] = { 0x3fac6af84b912394ull, 0x3fcebfce50fac4f3ull, 0x3fe62e42ff0c52d6ull };

PT_POWF_HD double PowfBitsToDouble( uint64_t b )
{
    double d;
    memcpy( &d, &b, 8 );
    return d;
}

PT_POWF_HD uint32_t PowfFloatBits( float f )
{
    uint32_t u;
    memcpy( &u, &f, 4 );
    return u;
}

PT_POWF_HD float PowfBitsToFloat( uint32_t u )
{
    float f;
    memcpy( &f, &u, 4 );
    return f;
}

// 0: y is not an integer, 1: odd integer, 2: even integer
PT_POWF_HD int PowfCheckInt( uint32_t iy )
{
    const int e = static_cast< int >( ( iy >> 23 ) & 0xffu );
    if ( e < 0x7f ) return 0;
    if ( e > 0x7f + 23 ) return 2;
    if ( iy & ( ( 1u << ( 0x7f + 23 - e ) ) - 1u ) ) return 0;
    if ( iy & ( 1u << ( 0x7f + 23 - e ) ) ) return 1;
    return 2;
}

PT_POWF_HD bool PowfZeroInfNan( uint32_t i ) { return 2u * i - 1u >= 2u * 0x7f800000u - 1u; }

PT_POWF_HD float PowLibm( float x, float y )
{
    uint32_t signBias = 0; // 1 << ( 5 + 11 ) once the result is negative
    uint32_t ix = PowfFloatBits( x );
    const uint32_t iy = PowfFloatBits( y );
    if ( ix - 0x00800000u >= 0x7f800000u - 0x00800000u || PowfZeroInfNan( iy ) )
    {
        // x below 2^-126, infinite or NaN, or y zero, infinite or NaN
        if ( PowfZeroInfNan( iy ) )
        {
            if ( 2u * iy == 0 ) return 1.0f; // x^0 (a signalling NaN x aside)
            if ( ix == 0x3f800000u ) return 1.0f;
            if ( 2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u ) return x + y;
            if ( 2u * ix == 2u * 0x3f800000u ) return 1.0f;
            if ( ( 2u * ix < 2u * 0x3f800000u ) == !( iy & 0x80000000u ) ) return 0.0f; // |x| < 1 and y = inf, or |x| > 1 and y = -inf
            return y * y;
        }
        if ( PowfZeroInfNan( ix ) )
        {
            float x2 = x * x;
            if ( ( ix & 0x80000000u ) && PowfCheckInt( iy ) == 1 )
            {
                x2       = -x2;
                signBias = 1;
            }
            if ( 2u * ix == 0 && ( iy & 0x80000000u ) ) return signBias ? -PowfBitsToFloat( 0x7f800000u ) : PowfBitsToFloat( 0x7f800000u ); // 1 / +-0
            return ( iy & 0x80000000u ) ? 1 / x2 : x2;
        }
        // x and y are finite and not zero
        if ( ix & 0x80000000u )
        {
            const int yint = PowfCheckInt( iy );
            if ( yint == 0 ) // negative base, fractional exponent: NaN
            {
                const float zero = x - x;
                return zero / zero;
            }
            if ( yint == 1 ) signBias = 1u << ( 5 + 11 );
            ix &= 0x7fffffffu;
        }
        if ( ix < 0x00800000u )
        {
            // subnormal x: scale it up and take the 23 off the exponent again
            ix = PowfFloatBits( x * 8388608.0f );
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }

    // log2( x ) = log1p( z / c - 1 ) / ln 2 + log2( c ) + k, x = 2^k z, z in [ 0x1.66p-1, 0x1.66p0 )
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i   = ( tmp >> ( 23 - 4 ) ) % 16u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz  = ix - top;
    const int k        = static_cast< int32_t >( top ) >> 23;
    const double invc = PowfBitsToDouble( kPowfLog2Table[i][0] ), logc = PowfBitsToDouble( kPowfLog2Table[i][1] );
    const double A0 = PowfBitsToDouble( kPowfLog2Poly[0] ), A1 = PowfBitsToDouble( kPowfLog2Poly[1] ), A2 = PowfBitsToDouble( kPowfLog2Poly[2] ),
                 A3 = PowfBitsToDouble( kPowfLog2Poly[3] ), A4 = PowfBitsToDouble( kPowfLog2Poly[4] );
    const double z  = static_cast< double >( PowfBitsToFloat( iz ) );
    const double r  = fma( z, invc, -1.0 );
    const double y0 = logc + static_cast< double >( k );
    const double r2 = r * r;
    double yy       = fma( A0, r, A1 );
    const double p  = fma( A2, r, A3 );
    const double r4 = r2 * r2;
    double q        = fma( A4, r, y0 );
    q               = fma( p, r2, q );
    yy              = fma( yy, r4, q );
    const double ylogx = static_cast< double >( y ) * yy; // cannot overflow: y is a float

    uint64_t ybits;
    memcpy( &ybits, &ylogx, 8 );
    if ( ( ( ybits >> 47 ) & 0xffffu ) >= ( 0x405f800000000000ull >> 47 ) ) // |y log2 x| >= 126
    {
        if ( ylogx > PowfBitsToDouble( 0x405fffffffd1d571ull ) ) return signBias ? -PowfBitsToFloat( 0x7f800000u ) : PowfBitsToFloat( 0x7f800000u ); // overflow
        if ( ylogx <= -150.0 ) return signBias ? -0.0f : 0.0f;                                                                              // underflow
        if ( ylogx < -149.0 ) return signBias ? -PowfBitsToFloat( 1u ) : PowfBitsToFloat( 1u ); // (0x1.4p-75f)^2: the smallest subnormal
    }

    // 2^( y log2 x ): x = k / 32 + r with |r| <= 1 / 64, 2^x = 2^( k / 32 ) 2^r
    const double shift = PowfBitsToDouble( 0x42e8000000000000ull ); // 0x1.8p52 / 32
    double kd = ylogx + shift;
    uint64_t ki;
    memcpy( &ki, &kd, 8 );
    kd -= shift;
    const double rr = ylogx - kd;
    uint64_t t      = kExp2fTable[ki % 32u];
    t += ( ki + signBias ) << ( 52 - 5 );
    const double s  = PowfBitsToDouble( t );
    const double C0 = PowfBitsToDouble( kExp2fPoly[0] ), C1 = PowfBitsToDouble( kExp2fPoly[1] ), C2 = PowfBitsToDouble( kExp2fPoly[2] );
    const double zz = fma( C0, rr, C1 );
    const double rr2 = rr * rr;
    double e         = fma( C2, rr, 1.0 );
    e                = fma( zz, rr2, e );
    e                = e * s;
    return static_cast< float >( e );
}

} // namespace ptd
