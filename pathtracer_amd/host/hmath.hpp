// Host-side float vector / matrix helpers.
//
// The reference does all of its host math with glm 0.9.9.2 (ext/glm).  Scene flattening has to produce
// the same floats the reference's host would hand to the integrator, so every helper here performs the
// same IEEE binary32 operations in the same order as the glm routine it stands in for (cited per
// function).  Build with -ffp-contract=off.
#pragma once

#include <cmath>
#include <cstdint>

namespace pth
{

constexpr float kPi = 3.14159265358979323846f; // src/math.hpp:9-11 (float M_PI, the MSVC semantics)

struct vec2
{
    float x = 0, y = 0;
};

struct vec3
{
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3( float a ) : x( a ), y( a ), z( a ) {}
    vec3( float a, float b, float c ) : x( a ), y( b ), z( c ) {}
    float& operator[]( int i ) { return ( &x )[i]; }
    float operator[]( int i ) const { return ( &x )[i]; }
};

inline vec3 operator+( vec3 a, vec3 b ) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-( vec3 a, vec3 b ) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*( vec3 a, vec3 b ) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3 operator*( vec3 a, float s ) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*( float s, vec3 a ) { return { s * a.x, s * a.y, s * a.z }; }
inline vec3 operator-( vec3 a ) { return { -a.x, -a.y, -a.z }; }
inline vec3& operator+=( vec3& a, vec3 b ) { a = a + b; return a; }
inline bool operator==( vec3 a, vec3 b ) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=( vec3 a, vec3 b ) { return !( a == b ); }

// glm/detail/func_geometric.inl:48-55
inline float dot( vec3 a, vec3 b )
{
    vec3 t = a * b;
    return t.x + t.y + t.z;
}

// glm/detail/func_geometric.inl:68-79
inline vec3 cross( vec3 x, vec3 y )
{
    return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
}

// glm/detail/func_geometric.inl:8-14
inline float length( vec3 v ) { return std::sqrt( dot( v, v ) ); }

// glm/detail/func_geometric.inl:82-90 with func_exponential.inl:44-50: v * (1 / sqrt(dot(v, v)))
inline vec3 normalize( vec3 v ) { return v * ( 1.0f / std::sqrt( dot( v, v ) ) ); }

// glm/detail/func_common.inl:16-20 and the matching max
inline float gmin( float x, float y ) { return ( y < x ) ? y : x; }
inline float gmax( float x, float y ) { return ( x < y ) ? y : x; }
inline vec3 vmin( vec3 a, vec3 b ) { return { gmin( a.x, b.x ), gmin( a.y, b.y ), gmin( a.z, b.z ) }; }
inline vec3 vmax( vec3 a, vec3 b ) { return { gmax( a.x, b.x ), gmax( a.y, b.y ), gmax( a.z, b.z ) }; }

// glm::radians: degrees * 0.01745329251994329576923690768489f
inline float radians( float deg ) { return deg * 0.01745329251994329576923690768489f; }
inline vec3 radians( vec3 d ) { return { radians( d.x ), radians( d.y ), radians( d.z ) }; }

struct vec4
{
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4( float a, float b, float c, float d ) : x( a ), y( b ), z( c ), w( d ) {}
    vec4( vec3 v, float d ) : x( v.x ), y( v.y ), z( v.z ), w( d ) {}
    float& operator[]( int i ) { return ( &x )[i]; }
    float operator[]( int i ) const { return ( &x )[i]; }
};

inline vec4 operator+( vec4 a, vec4 b ) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline vec4 operator-( vec4 a, vec4 b ) { return { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
inline vec4 operator*( vec4 a, vec4 b ) { return { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }
inline vec4 operator*( vec4 a, float s ) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }
inline vec3 xyz( vec4 v ) { return { v.x, v.y, v.z }; }

// Column-major 4x4, m[c] is column c like glm::mat4.
struct mat4
{
    vec4 c[4];
    mat4() : mat4( 1.0f ) {}
    explicit mat4( float d ) { c[0] = { d, 0, 0, 0 }; c[1] = { 0, d, 0, 0 }; c[2] = { 0, 0, d, 0 }; c[3] = { 0, 0, 0, d }; }
    vec4& operator[]( int i ) { return c[i]; }
    const vec4& operator[]( int i ) const { return c[i]; }
};

// glm/detail/type_mat4x4.inl:561-572: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline vec4 operator*( const mat4& m, vec4 v )
{
    vec4 mul0 = m[0] * v.x;
    vec4 mul1 = m[1] * v.y;
    vec4 add0 = mul0 + mul1;
    vec4 mul2 = m[2] * v.z;
    vec4 mul3 = m[3] * v.w;
    vec4 add1 = mul2 + mul3;
    return add0 + add1;
}

// glm/ext/matrix_transform.inl translate
inline mat4 translate( const mat4& m, vec3 v )
{
    mat4 r = m;
    r[3]   = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}

// glm/ext/matrix_transform.inl rotate (axis-angle); cos/sin are the float overloads
inline mat4 rotate( const mat4& m, float angle, vec3 v )
{
    const float c = std::cos( angle );
    const float s = std::sin( angle );
    vec3 axis     = normalize( v );
    vec3 temp     = ( 1.0f - c ) * axis;

    float r00 = c + temp[0] * axis[0];
    float r01 = temp[0] * axis[1] + s * axis[2];
    float r02 = temp[0] * axis[2] - s * axis[1];
    float r10 = temp[1] * axis[0] - s * axis[2];
    float r11 = c + temp[1] * axis[1];
    float r12 = temp[1] * axis[2] + s * axis[0];
    float r20 = temp[2] * axis[0] + s * axis[1];
    float r21 = temp[2] * axis[1] - s * axis[0];
    float r22 = c + temp[2] * axis[2];

    mat4 r;
    r[0] = m[0] * r00 + m[1] * r01 + m[2] * r02;
    r[1] = m[0] * r10 + m[1] * r11 + m[2] * r12;
    r[2] = m[0] * r20 + m[1] * r21 + m[2] * r22;
    r[3] = m[3];
    return r;
}

// glm/ext/matrix_transform.inl scale
inline mat4 scale( const mat4& m, vec3 v )
{
    mat4 r;
    r[0] = m[0] * v.x;
    r[1] = m[1] * v.y;
    r[2] = m[2] * v.z;
    r[3] = m[3];
    return r;
}

inline mat4 transpose( const mat4& m )
{
    mat4 r;
    for ( int i = 0; i < 4; ++i )
        for ( int j = 0; j < 4; ++j )
            r[i][j] = m[j][i];
    return r;
}

// glm/detail/func_matrix.inl:294-351 (cofactor expansion, same grouping)
inline mat4 inverse( const mat4& m )
{
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    vec4 Fac0( Coef00, Coef00, Coef02, Coef03 );
    vec4 Fac1( Coef04, Coef04, Coef06, Coef07 );
    vec4 Fac2( Coef08, Coef08, Coef10, Coef11 );
    vec4 Fac3( Coef12, Coef12, Coef14, Coef15 );
    vec4 Fac4( Coef16, Coef16, Coef18, Coef19 );
    vec4 Fac5( Coef20, Coef20, Coef22, Coef23 );

    vec4 Vec0( m[1][0], m[0][0], m[0][0], m[0][0] );
    vec4 Vec1( m[1][1], m[0][1], m[0][1], m[0][1] );
    vec4 Vec2( m[1][2], m[0][2], m[0][2], m[0][2] );
    vec4 Vec3( m[1][3], m[0][3], m[0][3], m[0][3] );

    vec4 Inv0 = Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2;
    vec4 Inv1 = Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4;
    vec4 Inv2 = Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5;
    vec4 Inv3 = Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5;

    vec4 SignA( +1, -1, +1, -1 );
    vec4 SignB( -1, +1, -1, +1 );
    mat4 Inverse;
    Inverse[0] = Inv0 * SignA;
    Inverse[1] = Inv1 * SignB;
    Inverse[2] = Inv2 * SignA;
    Inverse[3] = Inv3 * SignB;

    vec4 Row0( Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0] );
    vec4 Dot0 = m[0] * Row0;
    float Dot1 = ( Dot0.x + Dot0.y ) + ( Dot0.z + Dot0.w );
    float OneOverDeterminant = 1.0f / Dot1;

    mat4 r;
    for ( int i = 0; i < 4; ++i )
        r[i] = Inverse[i] * OneOverDeterminant;
    return r;
}

// Transform( position, rotation, scale ): src/transform.cpp:10-18 (T * Ry * Rx * Rz * S)
inline mat4 makeTransform( vec3 position, vec3 rotation, vec3 scl )
{
    mat4 m( 1.0f );
    m = translate( m, position );
    m = rotate( m, rotation.y, vec3( 0, 1, 0 ) );
    m = rotate( m, rotation.x, vec3( 1, 0, 0 ) );
    m = rotate( m, rotation.z, vec3( 0, 0, 1 ) );
    m = scale( m, scl );
    return m;
}

inline vec3 transformPoint( const mat4& m, vec3 p ) { return xyz( m * vec4( p, 1 ) ); }  // src/transform.cpp:30-33
inline vec3 transformVector( const mat4& m, vec3 v ) { return xyz( m * vec4( v, 0 ) ); } // src/transform.cpp:35-38

} // namespace pth
