// Wavefront path-tracing kernels for gfx950 (MI355X, wave64).
//
// One kernel per stage of the reference's Li loop (src/path_tracer.cpp:106-156), each a grid-stride pass
// over a queue of path ids:
//   raygen   primary rays with anti-aliasing jitter                    (path_tracer.cpp:183-189)
//   extend   closest-hit BVH traversal                                 (bvh.cpp:230-268)
//   shade    miss radiance, hit attributes, emission, albedo / texture, light sampling, BRDF sampling,
//            throughput update, queue compaction                       (path_tracer.cpp:115-152)
//   connect  any-hit traversal of the shadow rays + direct light sum    (bvh.cpp:270-310, path_tracer.cpp:90-104)
//   film     per-pixel accumulation in sample order                    (path_tracer.cpp:185-193)
//   tonemap  Uncharted2 / Reinhard, gamma, clamp                       (path_tracer.cpp:211-219)
//
// Arithmetic contract: binary32 in the reference's operation order, compiled with -ffp-contract=off and
// without fast-math, IEEE division and square root (sqrtf; __fsqrt_rn is the 1-ulp native instruction),
// comparisons instead of min/max in the slab test, so that hit primitives and distances equal the CPU's
// bit for bit.
//
// Traversal: the reference's node array is in depth-first pre-order with the first child at index + 1, and the
// reference always descends into the first child first.  For that fixed order the node to visit after a
// subtree is known statically (the pre-order successor of the subtree), so every device node carries it as a
// link, next to the index of its first child, and the traversal needs no stack: hit inner -> first child,
// miss or leaf done -> link.  The visit sequence is exactly the reference's stack sequence, with no per-lane
// stack in LDS or scratch, whatever order the nodes have in device memory.
// Two more layouts serve the front-to-back mode (PT_RENDER_ORDERED): records with one link per octant of ray
// directions for the closest hit (TraceClosestOrdered), and 16-byte records in the frame of a scene's one directional
// light for its shadow rays (ConnectParallelKernel).  Both keep the reference's answers, not its counters.
#include "pt_device.h"
#include "pt_api.h"
#include "pt_sincosf.h"
#include "pt_powf.h"
#include "pt_acosf.h"

#include <float.h>

namespace ptd
{

// ------------------------------------------------------------------------------------------------ vectors
struct f3
{
    float x, y, z;
};

__device__ __forceinline__ f3 F3( float x, float y, float z ) { return f3{ x, y, z }; }
__device__ __forceinline__ f3 F3( float4 v ) { return f3{ v.x, v.y, v.z }; }
__device__ __forceinline__ f3 F3( Packed3 v ) { return f3{ v.x, v.y, v.z }; }
__device__ __forceinline__ Packed3 P3( f3 v ) { return Packed3{ v.x, v.y, v.z }; }
__device__ __forceinline__ f3 F3( const float* p ) { return f3{ p[0], p[1], p[2] }; }
__device__ __forceinline__ f3 operator+( f3 a, f3 b ) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ f3 operator-( f3 a, f3 b ) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ f3 operator*( f3 a, f3 b ) { return f3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
__device__ __forceinline__ f3 operator*( f3 a, float s ) { return f3{ a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ f3 operator/( f3 a, float s ) { return f3{ a.x / s, a.y / s, a.z / s }; }
__device__ __forceinline__ f3 operator-( f3 a ) { return f3{ -a.x, -a.y, -a.z }; }
__device__ __forceinline__ bool IsZero( f3 a ) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
// glm semantics (glm/detail/func_geometric.inl): dot = (x + y) + z of the products, normalize = v * (1 / sqrt(dot))
__device__ __forceinline__ float Dot( f3 a, f3 b ) { f3 t = a * b; return t.x + t.y + t.z; }
__device__ __forceinline__ f3 Cross( f3 x, f3 y ) { return f3{ x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
__device__ __forceinline__ float Length( f3 a ) { return sqrtf( Dot( a, a ) ); }
__device__ __forceinline__ f3 Normalize( f3 a ) { return a * ( 1.0f / sqrtf( Dot( a, a ) ) ); }
__device__ __forceinline__ float AbsDot( f3 a, f3 b ) { return fabsf( Dot( a, b ) ); }
__device__ __forceinline__ float Max0( float x ) { return 0.0f < x ? x : 0.0f; } // std::max( 0.f, x )
__device__ __forceinline__ float4 F4( f3 a, float w ) { return make_float4( a.x, a.y, a.z, w ); }

constexpr float kPi      = 3.14159265358979323846f; // src/math.hpp:9-11
constexpr float kEpsilon = 0.00001f;                // src/path_tracer.cpp:16

// ------------------------------------------------------------------------------------------------ record loads
// BVH nodes and primitive records are read with buffer_load_dwordx4 through a buffer descriptor: the
// address is one 32-bit byte offset per lane (no 64-bit address arithmetic), a record is always fetched
// whole with all of its loads in flight together, and an offset past the end reads zeros instead of faulting.
typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
using BufferDesc = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ BufferDesc MakeBuffer( const void* base, uint32_t bytes )
{
    return __builtin_amdgcn_make_buffer_rsrc( const_cast< void* >( base ), 0, static_cast< int >( bytes ), 0x00020000 );
}

__device__ __forceinline__ float4 Load16( BufferDesc buffer, uint32_t byteOffset )
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128( buffer, static_cast< int >( byteOffset ), 0, 0 );
    return make_float4( __uint_as_float( v.x ), __uint_as_float( v.y ), __uint_as_float( v.z ), __uint_as_float( v.w ) );
}

// One ds_read_b128
__device__ __forceinline__ float4 LoadLds16( const float4* p )
{
    const u32x4 v = *reinterpret_cast< const u32x4* >( p );
    return make_float4( __uint_as_float( v.x ), __uint_as_float( v.y ), __uint_as_float( v.z ), __uint_as_float( v.w ) );
}

__device__ __forceinline__ uint32_t Load4( BufferDesc buffer, uint32_t byteOffset )
{
    return static_cast< uint32_t >( __builtin_amdgcn_raw_buffer_load_b32( buffer, static_cast< int >( byteOffset ), 0, 0 ) );
}

struct SceneBuffers
{
    BufferDesc nodes, prims;
};

// kOrdered: `nodes` is the front-to-back layout (64-byte records), else the reference-order layout (32-byte records)
template < bool kOrdered = false >
__device__ __forceinline__ SceneBuffers MakeSceneBuffers( const SceneView& sv )
{
    return SceneBuffers{ kOrdered ? MakeBuffer( sv.orderedNodes, sv.numOrderedNodes * 64u ) : MakeBuffer( sv.nodes, sv.numNodes * 32u ),
                         MakeBuffer( sv.prims, sv.numPrims * 64u ) };
}

// std::cos / std::sin of a float in the reference: the host C library's sincosf, reproduced exactly
__device__ __forceinline__ void SinCosF( float x, float* s, float* c ) { SinCosLibm( x, s, c ); }

// ------------------------------------------------------------------------------------------------ counters
struct Counters
{
    uint32_t rays = 0, nodes = 0, tris = 0, spheres = 0, shaded = 0;
};

__device__ __forceinline__ uint32_t WaveSum( uint32_t v )
{
    for ( int off = 32; off > 0; off >>= 1 ) v += __shfl_down( v, off, 64 );
    return v;
}

// One atomic per wave and counter.  Called by every lane of the wave at the end of a kernel.
__device__ __forceinline__ void FlushCounter( uint64_t* stats, int slot, uint32_t v )
{
    uint32_t sum = WaveSum( v );
    if ( ( threadIdx.x & 63 ) == 0 && sum ) atomicAdd( reinterpret_cast< unsigned long long* >( stats + slot ), static_cast< unsigned long long >( sum ) );
}

// ------------------------------------------------------------------------------------------------ primitive tests

// intersect::RayTriangle (src/intersection_tests.cpp:34-66) on {v0, v1 - v0, v2 - v0}
__device__ __forceinline__ bool RayTriangle( f3 o, f3 d, f3 v0, f3 v0v1, f3 v0v2, float maxT, float* t, float* u, float* v )
{
    f3 pvec   = Cross( d, v0v2 );
    float det = Dot( v0v1, pvec );
    if ( static_cast< double >( fabsf( det ) ) < 0.000000001 ) return false;
    float invDet = 1.0f / det;
    f3 tvec      = o - v0;
    *u           = Dot( tvec, pvec ) * invDet;
    if ( *u < 0 || *u > 1 ) return false;
    f3 qvec = Cross( tvec, v0v1 );
    *v      = Dot( d, qvec ) * invDet;
    if ( *v < 0 || *u + *v > 1 ) return false;
    *t = Dot( v0v2, qvec ) * invDet;
    return *t < maxT && *t > 0;
}

// intersect::RaySphere (src/intersection_tests.cpp:10-31)
__device__ __forceinline__ bool RaySphere( f3 o, f3 d, f3 centre, float radius, float maxT, float* t )
{
    f3 OC      = o - centre;
    float a    = Dot( d, d );
    float b    = 2 * Dot( OC, d );
    float c    = Dot( OC, OC ) - radius * radius;
    float disc = b * b - 4 * a * c;
    if ( disc < 0 ) return false;
    float sq = sqrtf( disc );
    *t       = ( -b - sq ) / ( 2 * a );
    if ( *t < 0 ) *t = ( -b + sq ) / ( 2 * a );
    return *t < maxT && *t > 0;
}

// glm mat4 * vec4 rows 0..2 (glm/detail/type_mat4x4.inl:561-572): (m0 * x + m1 * y) + (m2 * z + m3 * w)
__device__ __forceinline__ f3 MatMul( float4 c0, float4 c1, float4 c2, float4 c3, f3 p, float w )
{
    return f3{ ( c0.x * p.x + c1.x * p.y ) + ( c2.x * p.z + c3.x * w ),
               ( c0.y * p.x + c1.y * p.y ) + ( c2.y * p.z + c3.y * w ),
               ( c0.z * p.x + c1.z * p.y ) + ( c2.z * p.z + c3.z * w ) };
}

// intersect::RayAABBFastest (src/intersection_tests.cpp:91-110): comparisons and selects only, so that a
// NaN slab value behaves as on the CPU.
typedef float f2 __attribute__( ( ext_vector_type( 2 ) ) );

// Node record: a = {lo.x, lo.y, hi.x, hi.y}, b = {lo.z, hi.z, link, word} - the x and y planes of either side sit
// in adjacent registers, which is what the packed two-float subtract / multiply of the slab arithmetic wants.
// kInclusive: the box is entered when tMin <= maxT instead of tMin < maxT - the front-to-back closest-hit traversal
// needs it, because a primitive at exactly the distance of the closest hit so far may still win by its lower index.
// *entry: the distance at which the ray enters the box (the reference's tMin after the three slabs)
template < bool kInclusive = false >
__device__ __forceinline__ bool RayBoxEntry( f3 o, f3 inv, bool nx, bool ny, bool nz, float4 a, float4 b, float maxT, float* entry )
{
    // Both early returns of the reference are folded into flags: once one of them would have fired the later
    // values are never looked at, so evaluating everything up front gives the same answer and lets the whole
    // 32-byte node be fetched with two 16-byte loads before any arithmetic.
    // two planes per instruction: v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar forms
    const f2 tNear = ( f2{ nx ? a.z : a.x, ny ? a.w : a.y } - f2{ o.x, o.y } ) * f2{ inv.x, inv.y };
    const f2 tFar  = ( f2{ nx ? a.x : a.z, ny ? a.y : a.w } - f2{ o.x, o.y } ) * f2{ inv.x, inv.y };
    const f2 tZ    = ( f2{ nz ? b.y : b.x, nz ? b.x : b.y } - f2{ o.z, o.z } ) * f2{ inv.z, inv.z };
    float tMin = tNear.x, tMax = tFar.x;
    const float tyMin = tNear.y, tyMax = tFar.y, tzMin = tZ.x, tzMax = tZ.y;
    const bool missXY = ( tMin > tyMax ) | ( tyMin > tMax );
    if ( tyMin > tMin ) tMin = tyMin;
    if ( tyMax < tMax ) tMax = tyMax;
    const bool missZ = ( tMin > tzMax ) | ( tzMin > tMax );
    if ( tzMin > tMin ) tMin = tzMin;
    if ( tzMax < tMax ) tMax = tzMax;
    *entry = tMin;
    return !missXY & !missZ & ( kInclusive ? tMin <= maxT : tMin < maxT ) & ( tMax > 0 );
}

template < bool kInclusive = false >
__device__ __forceinline__ bool RayBox( f3 o, f3 inv, bool nx, bool ny, bool nz, float4 a, float4 b, float maxT )
{
    float entry;
    return RayBoxEntry< kInclusive >( o, inv, nx, ny, nz, a, b, maxT, &entry );
}

struct HitRecord
{
    float t, u, v;
    int prim;
};

// One primitive test of a closest-hit leaf: Triangle::Intersect / Sphere::Intersect (src/shapes.cpp:117-123,
// 52-61).  The record (the first 48 bytes of its 64-byte slot) is fetched whole before its tag is looked at (one memory round trip); the
// triangle arithmetic runs on whatever was fetched and is discarded for a sphere record.
// kOrdered (front-to-back traversal, see TraceClosestOrdered): the test accepts hits up to kBand times the closest
// distance so far, and `ambiguous` records whether a second hit lies within that band of the closest one.
constexpr float kBand = 1.0f + 1.0f / 4096.0f;

// kTrust: a hit of the front-to-back modes whose distance lies this far in front of the point where the ray enters the
// primitive's own leaf box comes from a primitive test that has lost its precision (a sphere seen from ten thousand radii
// away, a triangle met at a millionth of a degree).  The band argument rests on a hit lying inside its boxes, so such a ray
// is traced in the reference's order from scratch (bit 1 of `ambiguous`; bit 0: a second hit within the band).
constexpr float kTrust = 1.0f + 1.0f / 16384.0f;
constexpr uint32_t kAmbiguous = 1u, kUntrusted = 2u;

struct RaySide // what the reference's box test derives from a direction (bvh.cpp:235-236)
{
    f3 inv;
    bool nx, ny, nz;
};

template < bool kSpheres = true, bool kOrdered = false >
__device__ __forceinline__ void ClosestPrimTest( const SceneView& sv, const SceneBuffers& sb, uint32_t prim, f3 o, f3 d, HitRecord& h, bool* wasSphere, uint32_t* ambiguous = nullptr,
                                                 const RaySide* side = nullptr )
{
    float4 q0 = Load16( sb.prims, 64u * prim );
    const float4 q1 = Load16( sb.prims, 64u * prim + 16u ), q2 = Load16( sb.prims, 64u * prim + 32u );
    // Without the tag test nothing needs v0 before the determinant is known, and the compiler moves its load
    // behind that branch: a second memory round trip per leaf.  The empty statement pins the load up here.
    if constexpr ( !kSpheres ) asm volatile( "" : "+v"( q0.x ), "+v"( q0.y ), "+v"( q0.z ) );
    float t = 0.0f, u = 0.0f, v = 0.0f;
    const float limit = kOrdered ? h.t * kBand : h.t;
    bool hit   = RayTriangle( o, d, F3( q0 ), F3( q1 ), F3( q2 ), limit, &t, &u, &v );
    *wasSphere = kSpheres && __float_as_uint( q0.w ) != kTagTriangle; // a scene without spheres has triangle records only
    if ( *wasSphere )
    {
        // closest-hit sphere test runs in the sphere's local space against the unit sphere
        const float4* s = sv.spheres + 6 * __float_as_uint( q1.y );
        const float4 c0 = s[0], c1 = s[1], c2 = s[2], c3 = s[3];
        f3 lo = MatMul( c0, c1, c2, c3, o, 1.0f );
        f3 ld = MatMul( c0, c1, c2, c3, d, 0.0f );
        u = v = 0.0f;
        hit   = RaySphere( lo, ld, F3( 0.0f, 0.0f, 0.0f ), 1.0f, limit, &t );
    }
    if constexpr ( kOrdered )
    {
        // where the ray enters the box of the primitive's leaf (the box travels in the record)
        const float4 q3 = Load16( sb.prims, 64u * prim + 48u );
        float entry;
        RayBoxEntry< true >( o, side->inv, side->nx, side->ny, side->nz, make_float4( q1.w, q2.w, q3.y, q3.z ), make_float4( q3.x, q3.w, 0.0f, 0.0f ), limit, &entry );
        if ( hit && entry > t * kTrust ) *ambiguous |= kUntrusted;
        if ( hit && t < h.t )
        {
            // the hit it replaces was within the band of the new one (else everything so far lies beyond it)
            *ambiguous = ( *ambiguous & kUntrusted ) | ( h.t <= t * kBand ? kAmbiguous : 0u );
            h          = HitRecord{ t, u, v, static_cast< int >( prim ) };
        }
        else if ( hit )
            *ambiguous |= kAmbiguous;
    }
    else if ( hit )
        h = HitRecord{ t, u, v, static_cast< int >( prim ) };
}

// One primitive test of an any-hit leaf: TestIfHit (src/shapes.cpp:78-82, 136-141), spheres in world space.
__device__ __forceinline__ bool AnyPrimTest( const SceneBuffers& sb, uint32_t prim, f3 o, f3 d, float tMax, bool* wasSphere )
{
    const float4 q0 = Load16( sb.prims, 64u * prim ), q1 = Load16( sb.prims, 64u * prim + 16u ), q2 = Load16( sb.prims, 64u * prim + 32u );
    float t = 0.0f, u, v;
    bool hit   = RayTriangle( o, d, F3( q0 ), F3( q1 ), F3( q2 ), tMax, &t, &u, &v );
    *wasSphere = __float_as_uint( q0.w ) != kTagTriangle;
    if ( *wasSphere ) hit = RaySphere( o, d, F3( q0 ), q1.x, tMax, &t );
    return hit;
}

// BVH::Intersect (src/bvh.cpp:230-268).  Closest hit; the first primitive reaching a given t keeps it (strict <).
// maxT: FLT_MAX for Scene::Intersect (src/scene.cpp:372); the second pass over an ambiguous ray starts lower.
__device__ __forceinline__ HitRecord TraceClosest( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, Counters& cnt, float maxT = FLT_MAX )
{
    HitRecord h{ maxT, 0.0f, 0.0f, -1 };
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z );
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    const uint32_t end = sv.numNodes;
    uint32_t node = 0;
    while ( node != end )
    {
        const float4 a = Load16( sb.nodes, 32u * node ), b = Load16( sb.nodes, 32u * node + 16u );
        cnt.nodes++;
        const uint32_t link = __float_as_uint( b.z ), leaf = __float_as_uint( b.w );
        const bool boxHit = RayBox( o, inv, nx, ny, nz, a, b, h.t );
        if ( boxHit && !( link & kLeafFlag ) )
        {
            node = leaf; // first child
            continue;
        }
        node = link & ~kLeafFlag;
        if ( !boxHit ) continue;
        const uint32_t first = leaf & kLeafFirstMask, count = ( leaf >> 28 ) + 1;
        for ( uint32_t k = 0; k < count; ++k )
        {
            bool wasSphere;
            ClosestPrimTest( sv, sb, first + k, o, d, h, &wasSphere );
            cnt.tris += wasSphere ? 0 : 1;
            cnt.spheres += wasSphere ? 1 : 0;
        }
    }
    return h;
}

// BVH::Occluded (src/bvh.cpp:270-310).
__device__ __forceinline__ bool TraceAny( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, float tMax, Counters& cnt )
{
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z );
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    const uint32_t end = sv.numNodes;
    uint32_t node = 0;
    while ( node != end )
    {
        const float4 a = Load16( sb.nodes, 32u * node ), b = Load16( sb.nodes, 32u * node + 16u );
        cnt.nodes++;
        const uint32_t link = __float_as_uint( b.z ), leaf = __float_as_uint( b.w );
        const bool boxHit = RayBox( o, inv, nx, ny, nz, a, b, tMax );
        if ( boxHit && !( link & kLeafFlag ) )
        {
            node = leaf; // first child
            continue;
        }
        node = link & ~kLeafFlag;
        if ( !boxHit ) continue;
        const uint32_t first = leaf & kLeafFirstMask, count = ( leaf >> 28 ) + 1;
        for ( uint32_t k = 0; k < count; ++k )
        {
            bool wasSphere;
            const bool hit = AnyPrimTest( sb, first + k, o, d, tMax, &wasSphere );
            cnt.tris += wasSphere ? 0 : 1;
            cnt.spheres += wasSphere ? 1 : 0;
            if ( hit ) return true;
        }
    }
    return false;
}

// Second pass over a ray whose front-to-back pass found two hits within kBand of each other: the reference's order,
// but starting from a distance just above the band instead of FLT_MAX.  Everything the reference's answer can depend
// on lies below that start: a hit farther away only ever culls boxes that begin beyond it, so leaving the far part
// of the tree out changes no decision about the hits inside the band - and it is the far part that makes the
// reference's order expensive (1,067 nodes per ray on the 20 M triangle soup against 111 front to back).  Should the
// pass find nothing below its start (it cannot, the closest hit of the first pass lies there), it runs unbounded.
__device__ __forceinline__ HitRecord RetraceInReferenceOrder( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, float closest, Counters& cnt )
{
    if ( closest == FLT_MAX ) return TraceClosest( sv, sb, o, d, cnt ); // nothing to start from (an untrusted hit): the reference's pass as it is
    const float start = ( closest * kBand ) * kBand;
    HitRecord h = TraceClosest( sv, sb, o, d, cnt, start );
    if ( h.prim < 0 ) h = TraceClosest( sv, sb, o, d, cnt );
    return h;
}

// ---- front-to-back traversal ----------------------------------------------------------------------------
//
// The reference stores the split axis in every inner node (src/bvh.hpp:19) and never reads it (src/bvh.cpp:254-258).
// Here it picks the child to visit first: the one on the side the ray comes from.  For a fixed sign of the three
// direction components (an octant) that order is again a fixed pre-order of the tree, so the stackless scheme
// carries over: every node of the second device layout holds eight links, one per octant, each the node that
// follows the node's subtree in that octant's order.
//   record (64 bytes): {lo.x, lo.y, hi.x, hi.y} {lo.z, hi.z, w0, w1} {link[0..3]} {link[4..7]}
//   inner node: w0 = device index of its first child (the second child is the next record, both in one 128-byte
//               line), w1 = 2 + split axis;  leaf: w0 = leaf word, w1 = kLeafFlag
// A ray keeps linkOffset = 32 + 4 * octant: the byte offset of its link inside a record; bit (2 + axis) of that
// number is the sign of the ray along the axis, i.e. whether the second child is the near one.
// What keeps the closest hit the reference's.  The reference meets the primitives in index order and culls a box
// against the closest distance found so far, so where several hits lie within rounding of one another (coplanar
// or coincident surfaces, shared edges) its answer depends on its order: a box whose computed entry distance
// exceeds the distance of the primitive inside it by an ulp is skipped or not depending on what was found before.
// Only hits that close to the closest one can be involved.  The front-to-back pass therefore enters every box
// with tMin <= kBand * t, tests its primitives against the same limit, and notes whether a second hit lies within
// kBand of the closest: such a ray is ambiguous and is traced again in the reference's order (the rest of the
// tree cannot change the reference's answer for the others: a farther hit only ever culls boxes beyond the band).
__device__ __forceinline__ uint32_t LinkOffset( bool nx, bool ny, bool nz ) { return 32u + 4u * ( ( nx ? 1u : 0u ) | ( ny ? 2u : 0u ) | ( nz ? 4u : 0u ) ); }
__device__ __forceinline__ uint32_t NearChild( uint32_t w0, uint32_t w1, uint32_t linkOffset ) { return w0 + ( ( linkOffset >> ( w1 & 7u ) ) & 1u ); }

__device__ __forceinline__ HitRecord TraceClosestOrdered( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, Counters& cnt, uint32_t* ambiguous )
{
    HitRecord h{ FLT_MAX, 0.0f, 0.0f, -1 };
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z );
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    const uint32_t linkOffset = LinkOffset( nx, ny, nz );
    const uint32_t end = sv.numOrderedNodes;
    const RaySide side{ inv, nx, ny, nz };
    uint32_t node = 0;
    *ambiguous = 0u;
    while ( node != end )
    {
        const float4 a = Load16( sb.nodes, 64u * node ), b = Load16( sb.nodes, 64u * node + 16u );
        const uint32_t link = Load4( sb.nodes, 64u * node + linkOffset );
        cnt.nodes++;
        const uint32_t w0 = __float_as_uint( b.z ), w1 = __float_as_uint( b.w );
        // a box with a sphere inside is entered whatever has been found: the sphere test, which runs in the sphere's own
        // space, loses its precision with the distance of the origin long before the box test does, and a sphere hit that
        // the reference computes in front of the sphere's box has to be seen here to be told apart (kUntrusted)
        const bool boxHit = RayBox< true >( o, inv, nx, ny, nz, a, b, ( w1 & kSphereBelowFlag ) ? INFINITY : h.t * kBand );
        if ( boxHit && !( w1 & kLeafFlag ) )
        {
            node = NearChild( w0, w1, linkOffset );
            continue;
        }
        node = link;
        if ( !boxHit ) continue;
        const uint32_t first = w0 & kLeafFirstMask, count = ( w0 >> 28 ) + 1;
        for ( uint32_t k = 0; k < count; ++k )
        {
            bool wasSphere;
            ClosestPrimTest< true, true >( sv, sb, first + k, o, d, h, &wasSphere, ambiguous, &side );
            cnt.tris += wasSphere ? 0 : 1;
            cnt.spheres += wasSphere ? 1 : 0;
        }
    }
    return h;
}

// Any hit, near child first.  The box test keeps the reference's strict comparison against the fixed tMax, so the
// set of primitives that can be reached - and with it the answer - is the reference's whatever the order.
__device__ __forceinline__ bool TraceAnyOrdered( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, float tMax, Counters& cnt )
{
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z );
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    const uint32_t linkOffset = LinkOffset( nx, ny, nz );
    const uint32_t end = sv.numOrderedNodes;
    uint32_t node = 0;
    while ( node != end )
    {
        const float4 a = Load16( sb.nodes, 64u * node ), b = Load16( sb.nodes, 64u * node + 16u );
        const uint32_t link = Load4( sb.nodes, 64u * node + linkOffset );
        cnt.nodes++;
        const uint32_t w0 = __float_as_uint( b.z ), w1 = __float_as_uint( b.w );
        const bool boxHit = RayBox( o, inv, nx, ny, nz, a, b, tMax );
        if ( boxHit && !( w1 & kLeafFlag ) )
        {
            node = NearChild( w0, w1, linkOffset );
            continue;
        }
        node = link;
        if ( !boxHit ) continue;
        const uint32_t first = w0 & kLeafFirstMask, count = ( w0 >> 28 ) + 1;
        for ( uint32_t k = 0; k < count; ++k )
        {
            bool wasSphere;
            const bool hit = AnyPrimTest( sb, first + k, o, d, tMax, &wasSphere );
            cnt.tris += wasSphere ? 0 : 1;
            cnt.spheres += wasSphere ? 1 : 0;
            if ( hit ) return true;
        }
    }
    return false;
}

// ---- wide nodes ---------------------------------------------------------------------------------------------
//
// Fourth layout (wide_bvh.h): the reference's tree collapsed into nodes of up to eight children, the child boxes as
// bytes on a grid over the node's own box, rounded outwards.  One node visit is five 16-byte loads and eight box
// tests, where the binary layouts spend two or three loads on one; a ray makes a fifth of the visits.  The traversal
// keeps a short stack per ray: what is left of a node's hit children (the index of the node's first inner child, the
// mask of its inner slots, the mask of the hit ones).
// What keeps the answers the reference's.  The quantised boxes are outer bounds of the reference's boxes, and the
// test below is an outer bound of the reference's test: a child is skipped only if the reference's own slab test
// (src/intersection_tests.cpp:91-110) of the child's box - and with it of every box inside - would fail.  So the
// traversal reaches every primitive the front-to-back traversal of the exact boxes (TraceClosestOrdered) reaches, and
// possibly more; a primitive then counts only if the reference's test of its leaf's box, carried in its record,
// passes - exactly the condition under which the exact-box traversal would have tested it.  From there on the rules
// of the front-to-back mode apply unchanged: hits within kBand of the closest are kept alive, a ray that saw two is
// traced again in the reference's order.
//
// The box test.  Along one axis a child plane is X = p + q * cell (exact, as a real number), and the reference would
// compute t = fl( fl( L - o ) * inv ) for a plane L of the exact box, L >= X for a low plane.  Here t~ = fma( q, A, B )
// with A = cell * inv (exact: cell is a power of two) and B = fl( fl( p - o ) * inv ).  With u = 2^-24:
//   |t~ - ( X - o ) inv| <= 2.02 u |B| + 1.01 u |t~|,   |t - ( L - o ) inv| <= 2.01 u ( 1.01 |B| + 255 |A| ),
//   |t~| <= ( |B| + 255 |A| )( 1 + u ),
// so t >= t~ - E for a near plane and t <= t~ + E for a far plane with E = 2^-21 ( |B| + 255 |A| ) (6.1 u would do), per axis.
// The reference passes a box only if max( near ) <= min( far ), max( near ) <= limit and min( far ) > 0; the test
// below passes whenever max( near~ - E ) <= min( far~ + E ), max( near~ - E ) <= limit and min( far~ + E ) >= 0.
// Anything that is not a number passes (comparisons are written so that a NaN enters).
// A direction component of zero makes inv infinite, and the reference's slab test on that axis a containment test
// of the origin, whatever the magnitude of inv: the wide test uses 2^100 in its place, which keeps the arithmetic
// finite.  A finite reciprocal of 2^100 or more (a direction component below 2^-100) takes its axis out of the test
// (WideInv); a closest-hit ray with one is not traversed here at all but handed to the pass in the reference's order.
constexpr uint32_t kWideNodeBytes = 80u;
constexpr float kWideInvLimit = 1.2676506002282294e30f; // 2^100
constexpr float kWideSlack = 1.0f / 2097152.0f;          // 2^-21
constexpr uint32_t kWideStackEntries = 56;                // kWideMaxDepth of wide_bvh.h

// The reciprocal direction as the wide box test takes it: an infinite component (direction component zero) as 2^100 of its
// sign, and a finite one beyond 2^100 as a NaN - whatever is computed from it is then not a number, and the maxima and
// minima of the test pass over such operands: the axis is left out of the test, which only lets more through.
__device__ __forceinline__ float WideInvComponent( float x )
{
    const float a = fabsf( x );
    return ( a >= kWideInvLimit && a < INFINITY ) ? __builtin_nanf( "" ) : __builtin_amdgcn_fmed3f( x, -kWideInvLimit, kWideInvLimit );
}
__device__ __forceinline__ f3 WideInv( f3 inv ) { return F3( WideInvComponent( inv.x ), WideInvComponent( inv.y ), WideInvComponent( inv.z ) ); }

__device__ __forceinline__ bool WideDegenerate( f3 inv )
{
    const float ax = fabsf( inv.x ), ay = fabsf( inv.y ), az = fabsf( inv.z );
    return ( ax >= kWideInvLimit && ax < INFINITY ) || ( ay >= kWideInvLimit && ay < INFINITY ) || ( az >= kWideInvLimit && az < INFINITY );
}

// Per axis the slots a ray prefers - those on the side it comes from -, z in the low byte, then y, then x
__device__ __forceinline__ uint32_t WideOctMasks( bool nx, bool ny, bool nz )
{
    return ( nz ? 0xF0u : 0x0Fu ) | ( ny ? 0xCC00u : 0x3300u ) | ( nx ? 0xAA0000u : 0x550000u );
}

// Of the slots in `hits` (not empty) the one with the smallest slot ^ octant: the nearest, as far as slots tell
__device__ __forceinline__ uint32_t WidePick( uint32_t hits, uint32_t octMasks )
{
    uint32_t t = hits & octMasks;
    hits = t ? t : hits;
    t    = hits & ( octMasks >> 8 );
    hits = t ? t : hits;
    t    = hits & ( octMasks >> 16 );
    hits = t ? t : hits;
    return static_cast< uint32_t >( __builtin_ctz( hits ) );
}

__device__ __forceinline__ float WideByte( uint32_t w, int k ) { return static_cast< float >( ( w >> ( 8 * k ) ) & 0xFFu ); }

// The eight box tests of one wide node: n0 = words 0..3 of its record, n2..n4 = the planes.  Bit s of the result: the
// ray may enter the child in slot s (empty slots are the caller's to mask off).
// kUnlimited: the high half of the result holds the same bits for limit = infinity.
template < bool kUnlimited = false >
__device__ __forceinline__ uint32_t WideBoxes( float4 n0, float4 n2, float4 n3, float4 n4, f3 o, f3 inv, bool nx, bool ny, bool nz, float limit )
{
    const uint32_t w3 = __float_as_uint( n0.w );
    const float ix = inv.x, iy = inv.y, iz = inv.z; // WideInv of the ray's reciprocal direction
    const float ax = __uint_as_float( ( w3 & 0xFFu ) << 23 ) * ix, ay = __uint_as_float( ( w3 & 0xFF00u ) << 15 ) * iy, az = __uint_as_float( ( w3 & 0xFF0000u ) << 7 ) * iz;
    const float bx = ( n0.x - o.x ) * ix, by = ( n0.y - o.y ) * iy, bz = ( n0.z - o.z ) * iz;
    // every axis with its own E: a direction along an axis (the shadow rays of a light straight above) has bounds of the
    // order 2^100 on the other two, which must not loosen the one that counts
    const float ex = __builtin_fmaf( fabsf( ax ), 255.0f * kWideSlack, fabsf( bx ) * kWideSlack ), ey = __builtin_fmaf( fabsf( ay ), 255.0f * kWideSlack, fabsf( by ) * kWideSlack ),
                ez = __builtin_fmaf( fabsf( az ), 255.0f * kWideSlack, fabsf( bz ) * kWideSlack );
    // near planes carry -E, far planes +E: max( near ) <= min( far ) below is then the test with E on either side
    const f2 ax2 = f2{ ax, ax }, ay2 = f2{ ay, ay }, az2 = f2{ az, az };
    const f2 bx2 = f2{ bx - ex, bx + ex }, by2 = f2{ by - ey, by + ey }, bz2 = f2{ bz - ez, bz + ez };
    const float lower = 0.0f, upper = limit;
    const uint32_t lox[2] = { __float_as_uint( n2.x ), __float_as_uint( n2.y ) }, loy[2] = { __float_as_uint( n2.z ), __float_as_uint( n2.w ) },
                   loz[2] = { __float_as_uint( n3.x ), __float_as_uint( n3.y ) }, hix[2] = { __float_as_uint( n3.z ), __float_as_uint( n3.w ) },
                   hiy[2] = { __float_as_uint( n4.x ), __float_as_uint( n4.y ) }, hiz[2] = { __float_as_uint( n4.z ), __float_as_uint( n4.w ) };
    uint32_t nearX[2], farX[2], nearY[2], farY[2], nearZ[2], farZ[2];
#pragma unroll
    for ( int k = 0; k < 2; ++k )
    {
        nearX[k] = nx ? hix[k] : lox[k]; farX[k] = nx ? lox[k] : hix[k];
        nearY[k] = ny ? hiy[k] : loy[k]; farY[k] = ny ? loy[k] : hiy[k];
        nearZ[k] = nz ? hiz[k] : loz[k]; farZ[k] = nz ? loz[k] : hiz[k];
    }
    uint32_t hits = 0;
#pragma unroll
    for ( int s = 0; s < 8; ++s )
    {
        const int w = s >> 2, k = s & 3;
        const f2 tx = __builtin_elementwise_fma( f2{ WideByte( nearX[w], k ), WideByte( farX[w], k ) }, ax2, bx2 );
        const f2 ty = __builtin_elementwise_fma( f2{ WideByte( nearY[w], k ), WideByte( farY[w], k ) }, ay2, by2 );
        const f2 tz = __builtin_elementwise_fma( f2{ WideByte( nearZ[w], k ), WideByte( farZ[w], k ) }, az2, bz2 );
        const float tMin = __builtin_fmaxf( __builtin_fmaxf( __builtin_fmaxf( tx.x, ty.x ), tz.x ), lower );
        const float tFar = __builtin_fminf( __builtin_fminf( tx.y, ty.y ), tz.y );
        const float tMax = __builtin_fminf( tFar, upper );
        hits |= !( tMin > tMax ) ? 1u << s : 0u;
        if constexpr ( kUnlimited ) hits |= !( tMin > tFar ) ? 0x10000u << s : 0u;
    }
    return hits;
}

// The state a traversal keeps of the node it tested last: what is left to visit of its children
struct WideGroup
{
    uint32_t childBase, primBase;
    uint32_t bits; // hit inner slots | inner slots << 8 | hit leaf slots << 16 | leaf slots << 24
};

// kSpheres (closest-hit traversals of a scene with spheres): a child with a sphere below it is entered whatever has been
// found - see TraceClosestOrdered
template < bool kSpheres = false >
__device__ __forceinline__ WideGroup WideVisit( BufferDesc nodes, uint32_t node, f3 o, f3 inv, bool nx, bool ny, bool nz, float limit )
{
    const uint32_t at = kWideNodeBytes * node;
    const float4 n0 = Load16( nodes, at ), n1 = Load16( nodes, at + 16u ), n2 = Load16( nodes, at + 32u ), n3 = Load16( nodes, at + 48u ), n4 = Load16( nodes, at + 64u );
    uint32_t hits = WideBoxes< kSpheres >( n0, n2, n3, n4, o, inv, nx, ny, nz, limit );
    if constexpr ( kSpheres ) hits = ( hits | ( ( hits >> 16 ) & __float_as_uint( n1.w ) ) ) & 0xFFu;
    const uint32_t imask = __float_as_uint( n0.w ) >> 24, lmask = __float_as_uint( n1.z ) & 0xFFu;
    return WideGroup{ __float_as_uint( n1.x ), __float_as_uint( n1.y ), ( hits & imask ) | imask << 8 | ( hits & lmask ) << 16 | lmask << 24 };
}

// Closest-hit test of the primitive in position `slot` of the wide primitive array (ClosestPrimTest< ., true > plus the
// reference's test of the leaf box, see above)
template < bool kSpheres >
__device__ __forceinline__ void ClosestPrimTestWide( const SceneView& sv, BufferDesc prims, uint32_t slot, f3 o, f3 d, f3 inv, bool nx, bool ny, bool nz,
                                                     HitRecord& h, bool* wasSphere, uint32_t* ambiguous )
{
    const float4 q0 = Load16( prims, 64u * slot ), q1 = Load16( prims, 64u * slot + 16u ), q2 = Load16( prims, 64u * slot + 32u ), q3 = Load16( prims, 64u * slot + 48u );
    const uint32_t tag = __float_as_uint( q0.w ); // index in the reference's order, top bit: sphere
    float t = 0.0f, u = 0.0f, v = 0.0f;
    const float limit = h.t * kBand;
    bool hit   = RayTriangle( o, d, F3( q0 ), F3( q1 ), F3( q2 ), limit, &t, &u, &v );
    *wasSphere = kSpheres && ( tag & 0x80000000u ) != 0;
    if ( *wasSphere )
    {
        const float4* s = sv.spheres + 6 * __float_as_uint( q1.y );
        const float4 c0 = s[0], c1 = s[1], c2 = s[2], c3 = s[3];
        f3 lo = MatMul( c0, c1, c2, c3, o, 1.0f );
        f3 ld = MatMul( c0, c1, c2, c3, d, 0.0f );
        u = v = 0.0f;
        hit   = RaySphere( lo, ld, F3( 0.0f, 0.0f, 0.0f ), 1.0f, limit, &t );
    }
    // the reference reaches the primitive only through its leaf's box (bvh.cpp:242)
    float entry;
    hit = hit & RayBoxEntry< true >( o, inv, nx, ny, nz, make_float4( q1.w, q2.w, q3.y, q3.z ), make_float4( q3.x, q3.w, 0.0f, 0.0f ), *wasSphere ? INFINITY : limit, &entry );
    if ( hit && entry > t * kTrust ) *ambiguous |= kUntrusted;
    if ( hit && t < h.t )
    {
        *ambiguous = ( *ambiguous & kUntrusted ) | ( h.t <= t * kBand ? kAmbiguous : 0u );
        h          = HitRecord{ t, u, v, static_cast< int >( tag & 0x7FFFFFFFu ) };
    }
    else if ( hit )
        *ambiguous |= kAmbiguous;
}

__device__ __forceinline__ bool AnyPrimTestWide( BufferDesc prims, uint32_t slot, f3 o, f3 d, f3 inv, bool nx, bool ny, bool nz, float tMax, bool* wasSphere )
{
    const float4 q0 = Load16( prims, 64u * slot ), q1 = Load16( prims, 64u * slot + 16u ), q2 = Load16( prims, 64u * slot + 32u ), q3 = Load16( prims, 64u * slot + 48u );
    float t = 0.0f, u, v;
    bool hit   = RayTriangle( o, d, F3( q0 ), F3( q1 ), F3( q2 ), tMax, &t, &u, &v );
    *wasSphere = ( __float_as_uint( q0.w ) & 0x80000000u ) != 0;
    if ( *wasSphere ) hit = RaySphere( o, d, F3( q0 ), q1.x, tMax, &t );
    return hit & RayBox( o, inv, nx, ny, nz, make_float4( q1.w, q2.w, q3.y, q3.z ), make_float4( q3.x, q3.w, 0.0f, 0.0f ), tMax );
}

__device__ __forceinline__ SceneBuffers MakeWideBuffers( const SceneView& sv )
{
    return SceneBuffers{ MakeBuffer( sv.wideNodes, sv.numWideNodes * kWideNodeBytes ), MakeBuffer( sv.widePrims, sv.numPrims * 64u ) };
}

// Closest hit over the wide layout, one ray per thread (the batch queries; the render path is ExtendWideKernel).
// *ambiguous: the ray has to be traced again in the reference's order.
__device__ __forceinline__ HitRecord TraceClosestWide( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, Counters& cnt, uint32_t* ambiguous )
{
    HitRecord h{ FLT_MAX, 0.0f, 0.0f, -1 };
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z );
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    *ambiguous    = WideDegenerate( inv ) ? kUntrusted : 0u;
    if ( *ambiguous ) return h;
    const uint32_t octMasks = WideOctMasks( nx, ny, nz );
    const f3 invW = WideInv( inv );
    uint2 stack[kWideStackEntries];
    uint32_t sp = 0;
    WideGroup g{ 0u, 0u, 0x0101u }; // the root as the one child of a node above it
    for ( ;; )
    {
        uint32_t ih = g.bits & 0xFFu, im = ( g.bits >> 8 ) & 0xFFu, base = g.childBase;
        if ( ih == 0 )
        {
            if ( sp == 0 ) break;
            --sp;
            base = stack[sp].x;
            ih   = stack[sp].y & 0xFFu;
            im   = stack[sp].y >> 8;
        }
        const uint32_t s = WidePick( ih, octMasks );
        ih &= ~( 1u << s );
        if ( ih && sp < kWideStackEntries ) stack[sp++] = make_uint2( base, im << 8 | ih );
        g = WideVisit< true >( sb.nodes, base + __popc( im & ( ( 1u << s ) - 1u ) ), o, invW, nx, ny, nz, h.t * kBand );
        cnt.nodes++;
        const uint32_t lmask = g.bits >> 24;
        for ( uint32_t l = ( g.bits >> 16 ) & 0xFFu; l; l &= l - 1u )
        {
            const uint32_t k = static_cast< uint32_t >( __builtin_ctz( l ) );
            bool wasSphere;
            ClosestPrimTestWide< true >( sv, sb.prims, g.primBase + __popc( lmask & ( ( 1u << k ) - 1u ) ), o, d, inv, nx, ny, nz, h, &wasSphere, ambiguous );
            cnt.tris += wasSphere ? 0 : 1;
            cnt.spheres += wasSphere ? 1 : 0;
        }
    }
    return h;
}

// Any hit over the wide layout: the reference's answer in any order, since a primitive counts only if the reference's
// test of its leaf box against the fixed tMax passes
__device__ __forceinline__ bool TraceAnyWide( const SceneView& sv, const SceneBuffers& sb, f3 o, f3 d, float tMax, Counters& cnt )
{
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z );
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    const uint32_t octMasks = WideOctMasks( nx, ny, nz );
    const f3 invW = WideInv( inv );
    uint2 stack[kWideStackEntries];
    uint32_t sp = 0;
    WideGroup g{ 0u, 0u, 0x0101u };
    for ( ;; )
    {
        uint32_t ih = g.bits & 0xFFu, im = ( g.bits >> 8 ) & 0xFFu, base = g.childBase;
        if ( ih == 0 )
        {
            if ( sp == 0 ) break;
            --sp;
            base = stack[sp].x;
            ih   = stack[sp].y & 0xFFu;
            im   = stack[sp].y >> 8;
        }
        const uint32_t s = WidePick( ih, octMasks );
        ih &= ~( 1u << s );
        if ( ih && sp < kWideStackEntries ) stack[sp++] = make_uint2( base, im << 8 | ih );
        g = WideVisit( sb.nodes, base + __popc( im & ( ( 1u << s ) - 1u ) ), o, invW, nx, ny, nz, tMax );
        cnt.nodes++;
        const uint32_t lmask = g.bits >> 24;
        for ( uint32_t l = ( g.bits >> 16 ) & 0xFFu; l; l &= l - 1u )
        {
            const uint32_t k = static_cast< uint32_t >( __builtin_ctz( l ) );
            bool wasSphere;
            const bool hit = AnyPrimTestWide( sb.prims, g.primBase + __popc( lmask & ( ( 1u << k ) - 1u ) ), o, d, inv, nx, ny, nz, tMax, &wasSphere );
            cnt.tris += wasSphere ? 0 : 1;
            cnt.spheres += wasSphere ? 1 : 0;
            if ( hit ) return true;
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ random numbers

// Philox4x32-10 (Salmon et al., SC'11).
__device__ __forceinline__ void Philox( uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4] )
{
#pragma unroll
    for ( int round = 0; round < 10; ++round )
    {
        uint32_t hi0 = __umulhi( 0xD2511F53u, c0 ), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi( 0xCD9E8D57u, c2 ), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Draws d and d + 1 of (pixel, sample), d even: the stand-in for two consecutive Random::Rand() calls
// (src/utils/random.cpp:9-16).  Both lie in one Philox block because 4 divides the block size.
__device__ __forceinline__ void DrawPair( const RngParams& rng, uint32_t pixel, uint32_t sample, uint32_t d, float* first, float* second )
{
    if ( rng.mode == PT_RNG_TABLE )
    {
        uint64_t base = ( static_cast< uint64_t >( pixel ) * static_cast< uint64_t >( rng.sppTotal ) + sample ) * static_cast< uint64_t >( rng.tableDraws );
        *first  = rng.table[( base + d ) % rng.tableLen];
        *second = rng.table[( base + d + 1 ) % rng.tableLen];
        return;
    }
    uint32_t w[4];
    Philox( pixel, sample, d >> 2, 0u, rng.seedLo, rng.seedHi, w );
    const uint32_t a = ( d & 2u ) ? w[2] : w[0], b = ( d & 2u ) ? w[3] : w[1];
    *first  = static_cast< float >( a >> 8 ) * ( 1.0f / 16777216.0f );
    *second = static_cast< float >( b >> 8 ) * ( 1.0f / 16777216.0f );
}

// ------------------------------------------------------------------------------------------------ sampling

// ConcentricSampleDisk / CosineSampleHemisphere (src/sampling.cpp:15-38, 48-53)
__device__ __forceinline__ f3 CosineSampleHemisphere( float u1, float u2 )
{
    u1 = 2 * u1 - 1;
    u2 = 2 * u2 - 1;
    float dx = 0.0f, dy = 0.0f;
    if ( !( u1 == 0 && u2 == 0 ) )
    {
        float theta, r = 1.0f;
        if ( u1 * u1 > u2 * u2 )
        {
            r *= u1;
            theta = ( kPi / 4 ) * ( u2 / u1 );
        }
        else
        {
            r *= u2;
            theta = ( kPi / 2 ) - ( kPi / 4 ) * ( u1 / u2 );
        }
        float s, c;
        SinCosF( theta, &s, &c );
        dx = r * c;
        dy = r * s;
    }
    float z = sqrtf( Max0( 1 - dx * dx - dy * dy ) );
    return F3( dx, dy, z );
}

// UniformSampleSphere (src/sampling.cpp:55-61)
__device__ __forceinline__ f3 UniformSampleSphere( float u1, float u2 )
{
    float z     = 1 - 2 * u1;
    float r     = sqrtf( Max0( 1 - z * z ) );
    float theta = 2 * kPi * u2;
    float s, c;
    SinCosF( theta, &s, &c );
    return F3( r * c, r * s, z );
}

// ------------------------------------------------------------------------------------------------ textures / environment

// Texture::GetPixel (src/resource/texture.cpp:49-66), rgb only
__device__ __forceinline__ f3 TextureFetch( const SceneView& sv, int tex, float u, float v )
{
    const DevTexture t = sv.textures[tex];
    // fmodf( x, 1 ) is x - trunc( x ), exactly: both have the same sign and exponent range, so the
    // subtraction does not round; for |x| >= 2^23 both give a zero, and a zero of either sign maps to texel 0
    u = u - truncf( u );
    u = u < 0 ? u + 1 : u;
    v = v - truncf( v );
    v = v < 0 ? v + 1 : v;
    int w = static_cast< int >( u * t.width );
    int h = static_cast< int >( v * t.height );
    if ( t.width - 1 < w ) w = t.width - 1;
    if ( t.height - 1 < h ) h = t.height - 1;
    const uint64_t offset = ( static_cast< uint64_t >( t.offsetHi ) << 32 | t.offsetLo ) + static_cast< uint64_t >( h * t.width + w );
    const uint32_t px = sv.texels[offset];
    const float k = 1.0f / 255.0f;
    return F3( k * static_cast< float >( px & 0xffu ), k * static_cast< float >( ( px >> 8 ) & 0xffu ), k * static_cast< float >( ( px >> 16 ) & 0xffu ) );
}

// Scene::LEnvironment (src/scene.cpp:386-393) with Skybox::GetPixel( ray ) (src/resource/skybox.cpp:84-118)
__device__ __forceinline__ f3 Environment( const SceneView& sv, f3 v )
{
    if ( !sv.skybox ) return F3( sv.background );
    f3 a = F3( fabsf( v.x ), fabsf( v.y ), fabsf( v.z ) );
    int face;
    float ma, ux, uy;
    if ( a.z >= a.x && a.z >= a.y )
    {
        face = v.z < 0.0f ? 5 : 4;
        ma   = 0.5f / a.z;
        ux   = v.z < 0.0f ? -v.x : v.x;
        uy   = v.y;
    }
    else if ( a.y >= a.x )
    {
        face = v.y < 0.0f ? 3 : 2;
        ma   = 0.5f / a.y;
        ux   = v.x;
        uy   = -v.z;
    }
    else
    {
        face = v.x < 0.0f ? 1 : 0;
        ma   = 0.5f / a.x;
        ux   = v.x < 0.0f ? v.z : -v.z;
        uy   = v.y;
    }
    ux = ux * ma + 0.5f;
    uy = uy * ma + 0.5f;
    int w = static_cast< int >( ux * sv.skyboxWidth );
    int h = static_cast< int >( uy * sv.skyboxHeight );
    if ( sv.skyboxWidth - 1 < w ) w = sv.skyboxWidth - 1;
    if ( sv.skyboxHeight - 1 < h ) h = sv.skyboxHeight - 1;
    const size_t idx = ( static_cast< size_t >( face ) * sv.skyboxHeight + h ) * sv.skyboxWidth + w;
    return F3( sv.skybox[idx] );
}

// ------------------------------------------------------------------------------------------------ kernels

constexpr int kBlock = 256;

// Path id -> (sample, pixel).  A slot is a pixel in tile order: the part of the image that 16 x 16 tiles cover,
// tile by tile and row-major inside a tile, then the columns right of it, then the rows below.  Path ids run tile
// by tile, and for a tile sample by sample: 256 consecutive ids - one unit of work, four wave loads - are one
// sample of one tile, and the units the GPU works on at any moment are all the samples of a few tiles, so the rays
// traced together start from neighbouring surface points.  The pixels outside the tiled area follow, sample-major.
struct PixelOrder
{
    uint32_t width, height;
};

struct PathOrder
{
    uint32_t width, height, samples; // samples of every pixel in this pass
};

__device__ __forceinline__ void PathToSampleSlot( PathOrder po, uint32_t path, uint32_t* sample, uint32_t* slot )
{
    const uint32_t tiled = ( po.width & ~15u ) * ( po.height & ~15u ), tiledPaths = tiled * po.samples;
    if ( path < tiledPaths )
    {
        const uint32_t unit = path >> 8;
        *sample = unit % po.samples;
        *slot   = ( unit / po.samples ) * 256u + ( path & 255u );
        return;
    }
    const uint32_t rest = po.width * po.height - tiled;
    path -= tiledPaths;
    *sample = path / rest;
    *slot   = tiled + path % rest;
}

__device__ __forceinline__ uint32_t SampleSlotToPath( PathOrder po, uint32_t sample, uint32_t slot )
{
    const uint32_t tiled = ( po.width & ~15u ) * ( po.height & ~15u );
    if ( slot < tiled ) return ( ( slot >> 8 ) * po.samples + sample ) * 256u + ( slot & 255u );
    return tiled * po.samples + sample * ( po.width * po.height - tiled ) + ( slot - tiled );
}

__device__ __forceinline__ void SlotToPixel( PixelOrder po, uint32_t slot, uint32_t* row, uint32_t* col )
{
    const uint32_t tiledW = po.width & ~15u, tiledH = po.height & ~15u, tiled = tiledW * tiledH;
    if ( slot < tiled )
    {
        const uint32_t tile = slot >> 8, k = slot & 255u, tilesX = tiledW >> 4;
        *row = ( tile / tilesX ) * 16u + ( k >> 4 );
        *col = ( tile % tilesX ) * 16u + ( k & 15u );
        return;
    }
    slot -= tiled;
    const uint32_t rightW = po.width - tiledW;
    if ( slot < rightW * po.height )
    {
        *row = slot / rightW;
        *col = tiledW + slot % rightW;
        return;
    }
    slot -= rightW * po.height;
    *row = tiledH + slot / tiledW;
    *col = slot % tiledW;
}

// Primary rays: pixel position, jitter, normalised direction (src/path_tracer.cpp:183-189,
// src/anti_aliasing.cpp:89-92).
__global__ __launch_bounds__( kBlock ) void RaygenKernel( CameraFrame cam, RngParams rng, PathBuffers pb, uint32_t width,
                                                         uint32_t height, uint32_t sampleBegin, uint32_t numPaths )
{
    const uint32_t numPixels = width * height;
    for ( uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < numPaths; p += gridDim.x * kBlock )
    {
        uint32_t local, slot, row, col;
        PathToSampleSlot( PathOrder{ width, height, numPaths / numPixels }, p, &local, &slot );
        SlotToPixel( PixelOrder{ width, height }, slot, &row, &col );
        const uint32_t pixel = row * width + col, sample = sampleBegin + local;
        const f3 position = F3( cam.position ), UL = F3( cam.UL ), dU = F3( cam.dU ), dV = F3( cam.dV );
        f3 imagePlanePos = UL + dV * static_cast< float >( row ) + dU * static_cast< float >( col );
        // RandFloat( -0.5f, 0.5f ) = Rand() * ( h - l ) + l (src/utils/random.cpp:18-21).  The reference as
        // built by g++ evaluates the dV factor of `c + R() * dU + R() * dV` first: draw 0 scales dV.
        float r0, r1;
        DrawPair( rng, pixel, sample, 0u, &r0, &r1 );
        const float jv = r0 * ( 0.5f - -0.5f ) + -0.5f;
        const float ju = r1 * ( 0.5f - -0.5f ) + -0.5f;
        f3 aa  = imagePlanePos + dU * ju + dV * jv;
        f3 dir = Normalize( aa - position );
        // at bounce 0 the queue is the identity (entry = path id); the origin of every path is the camera and the throughput it
        // starts with 1 (path_tracer.cpp:111, 189): the kernels that read them at bounce 0 know (PathBuffers::origin0)
        pb.rayD[0][p] = P3( dir );
    }
}

// ------------------------------------------------------------------------------------------------ work division
//
// The paths of a pass are cut into units of kUnitPaths consecutive path ids, and every queue into one segment
// per unit.  Whatever a path appends to a queue goes to the segment of the unit the path started in, and a unit
// is always processed by ONE wave at a time, so a segment's length is a wave-uniform register while the wave
// works on it and is written once when the unit is done - no atomics on the queues, no ordering between waves.
// The shading kernels take one unit per wave (the hardware's block scheduler balances them); the traversal
// kernels run a fixed number of resident waves, each pulling unit after unit from a shared cursor and feeding
// its lanes from that stream.
constexpr uint32_t kNoRay = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t WaveIndex() { return blockIdx.x * ( kBlock / 64 ) + ( threadIdx.x >> 6 ); }
__device__ __forceinline__ uint32_t LaneIndex() { return threadIdx.x & 63u; }
// Lane mask of a predicate, straight from the compare (the library's __ballot takes an int and compares it again)
__device__ __forceinline__ unsigned long long Ballot( bool flag ) { return __builtin_amdgcn_ballot_w64( flag ); }
__device__ __forceinline__ uint32_t LanesSet( bool flag ) { return static_cast< uint32_t >( __popcll( Ballot( flag ) ) ); }
__device__ __forceinline__ uint32_t LanesBelow( unsigned long long mask ) { return static_cast< uint32_t >( __popcll( mask & ( ( 1ull << LaneIndex() ) - 1ull ) ) ); }

__device__ __forceinline__ uint32_t* UnitCount( const PathBuffers& pb, int bounce, int which, uint32_t unit )
{
    return pb.counts + ( static_cast< size_t >( bounce ) * kCountsPerBounce + which ) * pb.units + unit;
}

// Appends `value` to a segment for every lane with `flag` set.  Must be reached by the whole wave (count is
// wave-uniform).
__device__ __forceinline__ void SegmentAppend( bool flag, uint32_t value, uint32_t* segment, uint32_t& count )
{
    const unsigned long long mask = Ballot( flag );
    if ( flag ) segment[count + LanesBelow( mask )] = value;
    count += static_cast< uint32_t >( __popcll( mask ) );
}

// The stream of queue entries a traversal wave feeds its lanes from: unit after unit, taken from a cursor that
// all waves of the launch share.  Everything in here is wave-uniform.
struct UnitStream
{
    uint32_t* cursor;       // next unit to hand out
    const uint32_t* counts; // entries per unit, or null: all paths of the pass (bounce 0)
    uint32_t numUnits, entriesPerUnit, numPaths;
    uint32_t base = 0, count = 0, taken = 0; // current unit: first entry, entries, entries handed out
    uint32_t nextUnit = 0, unitsHeld = 0;    // units taken from the cursor and not started yet
    bool ended = false;
};

// Units a wave takes from the cursor at a time: an atomic with a result is a round trip to the L2 during which the
// whole wave stands still, and all waves of a launch use the one address
#ifndef PT_UNIT_BATCH
#define PT_UNIT_BATCH 2
#endif

// Moves on to the next unit; false once the cursor has passed the last one.  Whole wave.
__device__ __forceinline__ bool NextUnit( UnitStream& s )
{
    if ( s.unitsHeld == 0 )
    {
        uint32_t first = 0;
        if ( LaneIndex() == 0 ) first = atomicAdd( s.cursor, static_cast< uint32_t >( PT_UNIT_BATCH ) );
        s.nextUnit  = __builtin_amdgcn_readfirstlane( first );
        s.unitsHeld = PT_UNIT_BATCH;
    }
    const uint32_t unit = s.nextUnit++;
    s.unitsHeld--;
    if ( unit >= s.numUnits )
    {
        s.ended = true;
        return false;
    }
    s.base  = unit * s.entriesPerUnit;
    s.taken = 0;
    if ( s.counts )
        s.count = s.counts[unit];
    else
    {
        const uint32_t left = s.numPaths - unit * kUnitPaths;
        s.count = left < kUnitPaths ? left : kUnitPaths;
    }
    return true;
}

// Adds a wave-uniform count to a statistics slot: one atomic per wave, issued by its first lane.
__device__ __forceinline__ void FlushWaveCounter( uint64_t* stats, int slot, uint32_t waveTotal )
{
    if ( LaneIndex() == 0 && waveTotal ) atomicAdd( reinterpret_cast< unsigned long long* >( stats + slot ), static_cast< unsigned long long >( waveTotal ) );
}

// ------------------------------------------------------------------------------------------------ traversal kernels
//
// Wave-level scheduling.  Rays differ widely in the number of nodes they visit, and a leaf turns up in some lane
// at almost every step.  A lane-per-ray loop therefore runs with most of the 64 lanes masked off: the wave waits
// for its longest ray, and every step pays for the primitive test of the few lanes that sit at a leaf.  Here
// every lane is a small state machine - traversing, waiting at a leaf, or waiting for a ray - and the WAVE
// decides what to do next:
//   refill    lanes without a ray store their result and take the next entries of the wave's stream, once
//             enough lanes are waiting (kExtendRefillLanes / kConnectRefillLanes)
//   leaf      lanes waiting at a leaf run the primitive tests together, once kLeafLanes lanes are waiting
//   traverse  otherwise: up to kTraverseBurst node visits for the traversing lanes
// A phase also runs as soon as nothing else can make progress, so every lane terminates.  Per ray the sequence
// of node visits and primitive tests is unchanged: a waiting lane does nothing, it never runs ahead.
#ifndef PT_EXTEND_REFILL_LANES
#define PT_EXTEND_REFILL_LANES 24
#endif
#ifndef PT_CONNECT_REFILL_LANES
#define PT_CONNECT_REFILL_LANES 32
#endif
#ifndef PT_LEAF_LANES
#define PT_LEAF_LANES 12
#endif
#ifndef PT_TRAVERSE_BURST
#define PT_TRAVERSE_BURST 3
#endif
// The traversal kernels keep the first kLdsTopNodes nodes of the node array (the top of the tree, most visited boxes first) in LDS, one copy
// per workgroup, and read them with ds_read instead of going through the texture addresser and the L1 - the
// busiest unit of these kernels.  Large workgroups share one copy between more waves: two workgroups per CU,
// 2,400 nodes = 75 KB each; 768 threads (6 waves per SIMD) at 80 VGPRs, 1,024 threads (8 waves per SIMD) at 64.  Measured, extend per launch of
// 66.4 M paths: no LDS copy, 256 threads x 7 per CU, 17.6 ms; 384 / 704 nodes in that shape 15.9 / 14.9 ms;
// 512 threads x 3 with 1,600 nodes 14.1 ms; 768 x 2 with 2,400 nodes 14.0 ms.  (Without the copy 7 waves per SIMD
// beat 6 by 7 % and 8, which spills in the node loop, lost 6 %.)
#ifndef PT_LDS_TOP_NODES
#define PT_LDS_TOP_NODES 2400
#endif
#ifndef PT_TRAVERSE_BLOCK
#define PT_TRAVERSE_BLOCK 768
#endif
#ifndef PT_TRAVERSE_WAVES
#define PT_TRAVERSE_WAVES 6
#endif
#define PT_TRAVERSE_OCCUPANCY __attribute__( ( amdgpu_waves_per_eu( PT_TRAVERSE_WAVES, PT_TRAVERSE_WAVES ) ) )
// The any-hit kernel keeps less per lane (no u, v, primitive; 63 VGPRs) and runs 8 waves per SIMD: 1,024 threads,
// two workgroups per CU.  Measured against 768 x 2 (6 waves): -4.7 % per launch; 896 x 2 (7 waves, 72 VGPRs): +32 %.
// ExtendKernel< false > (no sphere path, 61 VGPRs) runs 8 waves per SIMD: -2.5 % per launch against 6
#ifndef PT_TRIANGLE_BLOCK
#define PT_TRIANGLE_BLOCK 1024
#endif
#ifndef PT_TRIANGLE_WAVES
#define PT_TRIANGLE_WAVES 8
#endif
#ifndef PT_CONNECT_BLOCK
#define PT_CONNECT_BLOCK 1024
#endif
#ifndef PT_CONNECT_WAVES
#define PT_CONNECT_WAVES 8
#endif
#define PT_CONNECT_OCCUPANCY __attribute__( ( amdgpu_waves_per_eu( PT_CONNECT_WAVES, PT_CONNECT_WAVES ) ) )
constexpr uint32_t kLdsTopNodes = PT_LDS_TOP_NODES;
constexpr int kTraverseBlock    = PT_TRAVERSE_BLOCK;
constexpr int kConnectBlock     = PT_CONNECT_BLOCK;
constexpr int kTriangleBlock    = PT_TRIANGLE_BLOCK;
constexpr int kExtendRefillLanes  = PT_EXTEND_REFILL_LANES;
constexpr int kConnectRefillLanes = PT_CONNECT_REFILL_LANES;
constexpr int kLeafLanes     = PT_LEAF_LANES;
constexpr int kTraverseBurst = PT_TRAVERSE_BURST;

// Closest hits of the rays entering `bounce`: BVH::Intersect (src/bvh.cpp:230-268) under Scene::Intersect
// (src/scene.cpp:370-373).
// kOrdered: front-to-back traversal over the second node layout (see TraceClosestOrdered); the LDS copy then holds
// half as many records, each twice the size.
template < bool kSpheres, bool kOrdered >
__global__ __launch_bounds__( kSpheres ? kTraverseBlock : kTriangleBlock ) __attribute__( ( amdgpu_waves_per_eu( kSpheres ? PT_TRAVERSE_WAVES : PT_TRIANGLE_WAVES, kSpheres ? PT_TRAVERSE_WAVES : PT_TRIANGLE_WAVES ) ) ) void ExtendKernel( SceneView sv, PathBuffers pb, int bounce, uint32_t numPaths, uint64_t* stats )
{
    UnitStream stream{ pb.cursors + 2 * bounce, bounce == 0 ? nullptr : UnitCount( pb, bounce - 1, kCountNextPaths, 0 ), pb.units, kUnitPaths, numPaths };
    const Packed3* rayO = pb.rayO[bounce & 1];
    const Packed3* rayD = pb.rayD[bounce & 1];
    const uint32_t end    = kOrdered ? sv.numOrderedNodes : sv.numNodes;
    const SceneBuffers sb = MakeSceneBuffers< kOrdered >( sv );
    __shared__ float4 topLds[kLdsTopNodes > 0 ? 2 * kLdsTopNodes : 1];
    const uint32_t topCount = kOrdered ? ( kLdsTopNodes / 2 < sv.numTopOrdered ? kLdsTopNodes / 2 : sv.numTopOrdered )
                                       : ( kLdsTopNodes < sv.numTopNodes ? kLdsTopNodes : sv.numTopNodes );
    if ( kLdsTopNodes > 0 )
    {
        for ( uint32_t i = threadIdx.x; i < ( kOrdered ? 4 : 2 ) * topCount; i += ( kSpheres ? kTraverseBlock : kTriangleBlock ) ) topLds[i] = Load16( sb.nodes, 16u * i );
        __syncthreads();
    }
    // node: the node to visit next (or, while a leaf is pending, the node to go to after it); leaf: pending leaf word
    uint32_t entry = kNoRay, node = end, leaf = kInnerNode; // entry: the lane's ray, as its position in the bounce's queue
    f3 o = F3( 0.0f, 0.0f, 0.0f ), d = o, inv = o;
    bool nx = false, ny = false, nz = false;
    uint32_t linkOffset = 32u; // kOrdered: byte offset of the ray's link in a node record
    uint32_t ambiguous = 0;    // kOrdered: kAmbiguous (a second hit within kBand of the closest one, see TraceClosestOrdered) | kUntrusted
    HitRecord h{ FLT_MAX, 0.0f, 0.0f, -1 };
    uint32_t waveRays = 0, waveNodes = 0, waveTris = 0, waveSpheres = 0, waveSteps = 0; // whole-wave totals, wave-uniform

    for ( ;; )
    {
        const bool atLeaf     = leaf != kInnerNode;
        const bool idle       = node == end && !atLeaf; // a lane wants a ray if it is idle and has a result to store or the stream goes on
        // masks built from single compares (a ballot of a compound predicate costs two more vector instructions)
        const unsigned long long mLeaf = Ballot( atLeaf ), mEnd = Ballot( node == end );
        const unsigned long long mTrav = ~mEnd & ~mLeaf;
        const unsigned long long mRay  = mEnd & ~mLeaf & ( stream.ended ? Ballot( entry != kNoRay ) : ~0ull );
        if ( ( mLeaf | mTrav | mRay ) == 0 ) break;

        if ( mRay && ( __popcll( mRay ) >= kExtendRefillLanes || ( mTrav | mLeaf ) == 0 ) )
        {
            // (an untrusted hit leaves no distance for the second pass to start from)
            if ( idle && entry != kNoRay ) pb.hit[entry] = make_float4( kOrdered && ( ambiguous & kUntrusted ) ? FLT_MAX : h.t, h.u, h.v, __int_as_float( h.prim ) );
            if constexpr ( kOrdered )
            {
                // ambiguous rays are traced again in the reference's order by RedoKernel (rare: one atomic each)
                if ( idle && entry != kNoRay && ambiguous ) pb.redo[atomicAdd( pb.redoCount + bounce, 1u )] = entry;
            }
            if ( idle ) entry = kNoRay;
            while ( !stream.ended )
            {
                const unsigned long long mIdle = Ballot( idle && entry == kNoRay );
                if ( mIdle == 0 ) break;
                if ( stream.taken == stream.count )
                {
                    NextUnit( stream );
                    continue;
                }
                const uint32_t left = stream.count - stream.taken, asking = static_cast< uint32_t >( __popcll( mIdle ) );
                const uint32_t rank = LanesBelow( mIdle );
                if ( idle && entry == kNoRay && rank < left )
                {
                    entry = stream.base + stream.taken + rank;
                    o   = bounce == 0 ? F3( pb.origin0[0], pb.origin0[1], pb.origin0[2] ) : F3( rayO[entry] );
                    d   = F3( rayD[entry] );
                    inv = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:235-236
                    nx = inv.x < 0; ny = inv.y < 0; nz = inv.z < 0;
                    if constexpr ( kOrdered ) linkOffset = LinkOffset( nx, ny, nz );
                    h    = HitRecord{ FLT_MAX, 0.0f, 0.0f, -1 };     // scene.cpp:372
                    ambiguous = 0u;
                    node = 0;
                }
                const uint32_t given = asking < left ? asking : left;
                stream.taken += given;
                waveRays += given;
            }
            continue;
        }

        if ( mLeaf && ( __popcll( mLeaf ) >= kLeafLanes || mTrav == 0 ) )
        {
            const uint32_t first = leaf & kLeafFirstMask, shapes = atLeaf ? ( leaf >> 28 ) + 1 : 0;
            for ( uint32_t k = 0; Ballot( k < shapes ) != 0; ++k ) // bvh.cpp:247-250: every shape of the leaf, t shrinking
            {
                bool isTri = false, isSphere = false;
                if ( k < shapes )
                {
                    const RaySide side{ inv, nx, ny, nz };
                    ClosestPrimTest< kSpheres, kOrdered >( sv, sb, first + k, o, d, h, &isSphere, &ambiguous, &side );
                    isTri = !isSphere;
                }
                waveTris += LanesSet( isTri );
                waveSpheres += LanesSet( isSphere );
            }
            leaf = kInnerNode;
            continue;
        }

        // traverse: a few node visits in a row; stop early once half of the lanes have dropped out
        const uint32_t started = static_cast< uint32_t >( __popcll( mTrav ) );
        for ( int burst = 0; burst < kTraverseBurst; ++burst )
        {
            const bool go = node != end && leaf == kInnerNode;
            const uint32_t active = static_cast< uint32_t >( __popcll( Ballot( node != end ) & Ballot( leaf == kInnerNode ) ) );
            if ( burst && 2 * active < started ) break;
            waveNodes += active;
            waveSteps++;
            if ( go )
            {
                if constexpr ( kOrdered )
                {
                    float4 a, b;
                    uint32_t link;
                    // (reading only the box half of the top records from LDS, twice as many of them, and every link from
                    // memory: closest-hit launch 54.1 instead of 39.7 ms - the texture addresser is the busy unit)
                    if ( kLdsTopNodes > 0 && node < topCount )
                    {
                        a    = LoadLds16( topLds + 4 * node );
                        b    = LoadLds16( topLds + 4 * node + 1 );
                        link = *reinterpret_cast< const uint32_t* >( reinterpret_cast< const char* >( topLds ) + 64u * node + linkOffset );
                    }
                    else
                    {
                        a    = Load16( sb.nodes, 64u * node );
                        b    = Load16( sb.nodes, 64u * node + 16u );
                        link = Load4( sb.nodes, 64u * node + linkOffset );
                    }
                    const uint32_t w0 = __float_as_uint( b.z ), w1 = __float_as_uint( b.w );
                    const bool boxHit = RayBox< true >( o, inv, nx, ny, nz, a, b, ( kSpheres && ( w1 & kSphereBelowFlag ) ) ? INFINITY : h.t * kBand );
                    if ( boxHit && !( w1 & kLeafFlag ) )
                        node = NearChild( w0, w1, linkOffset );
                    else
                    {
                        node = link;
                        if ( boxHit ) leaf = w0;
                    }
                }
                else
                {
                    float4 a, b;
                    if ( kLdsTopNodes > 0 && node < topCount )
                    {
                        a = LoadLds16( topLds + 2 * node );
                        b = LoadLds16( topLds + 2 * node + 1 );
                    }
                    else
                    {
                        a = Load16( sb.nodes, 32u * node );
                        b = Load16( sb.nodes, 32u * node + 16u );
                    }
                    const uint32_t link = __float_as_uint( b.z ), word = __float_as_uint( b.w );
                    const bool boxHit = RayBox( o, inv, nx, ny, nz, a, b, h.t );
                    if ( boxHit && !( link & kLeafFlag ) )
                        node = word; // first child
                    else
                    {
                        node = link & ~kLeafFlag;
                        if ( boxHit ) leaf = word;
                    }
                }
            }
        }
    }
    FlushWaveCounter( stats, ST_RAYS_CLOSEST, waveRays );
    FlushWaveCounter( stats, ST_NODES_CLOSEST, waveNodes );
    FlushWaveCounter( stats, ST_TRI_CLOSEST, waveTris );
    FlushWaveCounter( stats, ST_SPH_CLOSEST, waveSpheres );
    FlushWaveCounter( stats, ST_STEPS_CLOSEST, waveSteps );
}

// Blocks of RedoKernel's grid.  How many rays there are is known on the device only; a thread takes every gridth of them, one
// after the other, each a chain of dependent fetches - so the grid is as large as the longest queue seen (a million
// threads; the ones without a ray leave at once)
#ifndef PT_REDO_BLOCKS
#define PT_REDO_BLOCKS 4096
#endif
// The rays ExtendKernel< ., true > found ambiguous, traced again in the reference's order (BVH::Intersect,
// src/bvh.cpp:230-268) over the reference-order layout: one thread per ray, the result replaces the first one.
__global__ __launch_bounds__( kBlock ) void RedoKernel( SceneView sv, PathBuffers pb, int bounce, uint64_t* stats )
{
    const uint32_t count = pb.redoCount[bounce];
    const SceneBuffers sb = MakeSceneBuffers< false >( sv );
    Counters cnt;
    uint32_t rays = 0;
    for ( uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock )
    {
        const uint32_t entry = pb.redo[i];
        const f3 o = bounce == 0 ? F3( pb.origin0[0], pb.origin0[1], pb.origin0[2] ) : F3( pb.rayO[bounce & 1][entry] );
        const HitRecord h = RetraceInReferenceOrder( sv, sb, o, F3( pb.rayD[bounce & 1][entry] ), pb.hit[entry].x, cnt );
        pb.hit[entry] = make_float4( h.t, h.u, h.v, __int_as_float( h.prim ) );
        rays++;
    }
    FlushCounter( stats, ST_RAYS_REDONE, rays );
    FlushCounter( stats, ST_NODES_CLOSEST, cnt.nodes );
    FlushCounter( stats, ST_TRI_CLOSEST, cnt.tris );
    FlushCounter( stats, ST_SPH_CLOSEST, cnt.spheres );
}

// ExtendKernel over the wide layout (see "wide nodes" above): the same wave-level scheduling - refill, primitive tests,
// bursts of node visits - with a short stack per lane, and with the primitive tests put off: a lane that has found leaf
// children keeps them in a short list and goes on visiting nodes (see leafLds).  The first kWideLdsStack entries of a lane's stack live in LDS
// (entry-major, so that the lanes of a wave touch consecutive words), deeper ones in `spill`, one slot per thread of
// the grid and level; the tree's depth bounds the stack, because a level pushes at most one entry.  Results and the
// queue of rays to trace again are ExtendKernel< ., true >'s.
#ifndef PT_WIDE_BLOCK
#define PT_WIDE_BLOCK 768
#endif
#ifndef PT_WIDE_WAVES
#define PT_WIDE_WAVES 6
#endif
#ifndef PT_WIDE_SPHERE_BLOCK
#define PT_WIDE_SPHERE_BLOCK 768
#endif
#ifndef PT_WIDE_SPHERE_WAVES
#define PT_WIDE_SPHERE_WAVES 6
#endif
#ifndef PT_WIDE_LDS_STACK
#define PT_WIDE_LDS_STACK 8
#endif
#ifndef PT_WIDE_REFILL_LANES
#define PT_WIDE_REFILL_LANES 24
#endif
#ifndef PT_WIDE_BURST
#define PT_WIDE_BURST 2
#endif
// refill, start of the primitive tests (lanes with untested leaf children) and burst length of the any-hit kernel over the
// wide layout; PT_WIDE_SHADOW_DEFERRED / PT_WIDE_SHADOW_KEEP_LANES below are its list length and stop threshold (list of
// 1 / 2 / 3: 13.10 / 13.06 / 13.46 ms per launch; start at 16 / 20 / 28 / 36: 13.22 / 13.06 / 13.22 / 13.60; testing at once: 13.7)
#ifndef PT_WIDE_SHADOW_REFILL_LANES
#define PT_WIDE_SHADOW_REFILL_LANES 24
#endif
#ifndef PT_WIDE_SHADOW_LEAF_LANES
#define PT_WIDE_SHADOW_LEAF_LANES 20
#endif
#ifndef PT_WIDE_SHADOW_BURST
#define PT_WIDE_SHADOW_BURST 2
#endif
constexpr int kWideBlock = PT_WIDE_BLOCK, kWideSphereBlock = PT_WIDE_SPHERE_BLOCK;
constexpr uint32_t kWideLdsStack = PT_WIDE_LDS_STACK;

// The list of leaf children a lane has found and not tested yet (see leafLds in ExtendWideKernel): its length, the lanes
// with a non-empty list at which the wave starts testing primitives, and at which it stops again.  Measured on a
// Sponza-class scene, closest-hit launch: list of 1 / 2 / 3 / 4: 30.3 / 28.9 / 29.1 / 29.6 ms (testing at once, as the kernel
// did before: 31.7); start at 12 / 16 / 20 / 24 / 32 lanes: 29.6 / 29.0 / 28.9 / 29.7 / 31.0; stop at 8 / 12 / 16: 29.2 / 28.9 / 29.0.
#ifndef PT_WIDE_DEFERRED
#define PT_WIDE_DEFERRED 2
#endif
#ifndef PT_WIDE_DEFER_LEAF_LANES
#define PT_WIDE_DEFER_LEAF_LANES 20
#endif
#ifndef PT_WIDE_DEFER_KEEP_LANES
#define PT_WIDE_DEFER_KEEP_LANES 12
#endif
constexpr uint32_t kWideDeferred = PT_WIDE_DEFERRED;

template < bool kSpheres >
__global__ __launch_bounds__( kSpheres ? kWideSphereBlock : kWideBlock ) __attribute__( ( amdgpu_waves_per_eu( kSpheres ? PT_WIDE_SPHERE_WAVES : PT_WIDE_WAVES, kSpheres ? PT_WIDE_SPHERE_WAVES : PT_WIDE_WAVES ) ) )
void ExtendWideKernel( SceneView sv, PathBuffers pb, int bounce, uint32_t numPaths, uint2* spill, uint64_t* stats )
{
    constexpr uint32_t kThreads = kSpheres ? kWideSphereBlock : kWideBlock;
    UnitStream stream{ pb.cursors + 2 * bounce, bounce == 0 ? nullptr : UnitCount( pb, bounce - 1, kCountNextPaths, 0 ), pb.units, kUnitPaths, numPaths };
    const Packed3* rayO = pb.rayO[bounce & 1];
    const Packed3* rayD = pb.rayD[bounce & 1];
    const SceneBuffers sb = MakeWideBuffers( sv );
    // A lane's stack: entries { index of a node's first inner child, its inner slots << 8 | those of them still to visit };
    // the first kWideLdsStack of them in LDS, entry-major
    __shared__ uint2 stackLds[kWideLdsStack * kThreads];
    uint2* const myLds = stackLds + threadIdx.x;
    // Primitive tests can wait: they only ever shrink the distance that culls.  A lane keeps the leaf children it has found
    // and not tested yet - per node { first leaf primitive, leaf slots << 8 | hit slots left } - in a short list and goes on
    // visiting nodes; the wave tests primitives when enough lanes have some, for as long as enough lanes have some.
    __shared__ uint2 leafLds[kWideDeferred * kThreads];
    uint2* const myLeaves = leafLds + threadIdx.x;
    uint32_t numLeaves = 0; // groups in the list
    // WidePick as a table: [octant << 8 | hit slots] -> the slot to visit next.  One LDS byte instead of fifteen vector
    // instructions per visit, in a kernel whose time goes by its vector instructions.
    __shared__ uint8_t pickLds[8 * 256];
    for ( uint32_t i = threadIdx.x; i < 8u * 256u; i += kThreads )
        pickLds[i] = ( i & 0xFFu ) ? static_cast< uint8_t >( WidePick( i & 0xFFu, WideOctMasks( ( i & 0x100u ) != 0, ( i & 0x200u ) != 0, ( i & 0x400u ) != 0 ) ) ) : uint8_t( 0 );
    __syncthreads();
    const size_t spillStride = static_cast< size_t >( gridDim.x ) * kThreads, spillLane = static_cast< size_t >( blockIdx.x ) * kThreads + threadIdx.x;
    uint32_t entry = kNoRay;
    f3 o = F3( 0.0f, 0.0f, 0.0f ), d = o, invW = o; // invW: WideInv of the reciprocal direction
    bool nx = false, ny = false, nz = false;
    uint32_t pickRow = 0; // the ray's row of pickLds: its octant << 8
    uint32_t sp = 0;
    uint32_t ambiguous = 0;
    HitRecord h{ FLT_MAX, 0.0f, 0.0f, -1 };
    uint32_t waveRays = 0, waveNodes = 0, waveTris = 0, waveSpheres = 0, waveSteps = 0; // whole-wave totals, wave-uniform

    for ( ;; )
    {
        const bool idle = ( numLeaves | sp ) == 0;
        const unsigned long long mLeaf = Ballot( numLeaves != 0 ), mIdle = Ballot( idle );
        const unsigned long long mTrav = Ballot( sp != 0 ) & Ballot( numLeaves < kWideDeferred );
        const unsigned long long mRay  = mIdle & ( stream.ended ? Ballot( entry != kNoRay ) : ~0ull );
        if ( ( mLeaf | mTrav | mRay ) == 0 ) break;

        if ( mRay && ( __popcll( mRay ) >= PT_WIDE_REFILL_LANES || ( mTrav | mLeaf ) == 0 ) )
        {
            if ( idle && entry != kNoRay )
            {
                pb.hit[entry] = make_float4( ( ambiguous & kUntrusted ) ? FLT_MAX : h.t, h.u, h.v, __int_as_float( h.prim ) );
                if ( ambiguous ) pb.redo[atomicAdd( pb.redoCount + bounce, 1u )] = entry;
                entry = kNoRay;
            }
            while ( !stream.ended )
            {
                const unsigned long long mAsk = Ballot( idle && entry == kNoRay );
                if ( mAsk == 0 ) break;
                if ( stream.taken == stream.count )
                {
                    NextUnit( stream );
                    continue;
                }
                const uint32_t left = stream.count - stream.taken, asking = static_cast< uint32_t >( __popcll( mAsk ) );
                const uint32_t rank = LanesBelow( mAsk );
                if ( idle && entry == kNoRay && rank < left )
                {
                    entry = stream.base + stream.taken + rank;
                    o   = bounce == 0 ? F3( pb.origin0[0], pb.origin0[1], pb.origin0[2] ) : F3( rayO[entry] );
                    d   = F3( rayD[entry] );
                    const f3 inv = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:235-236
                    nx = inv.x < 0; ny = inv.y < 0; nz = inv.z < 0;
                    pickRow   = ( nx ? 0x100u : 0u ) | ( ny ? 0x200u : 0u ) | ( nz ? 0x400u : 0u );
                    invW      = WideInv( inv );
                    h         = HitRecord{ FLT_MAX, 0.0f, 0.0f, -1 };     // scene.cpp:372
                    ambiguous = WideDegenerate( inv ) ? kUntrusted : 0u;  // such a ray goes straight to the second pass
                    sp        = ambiguous ? 0u : 1u;
                    myLds[0]  = make_uint2( 0u, 0x0101u );                // the root as the one child of a node above it
                }
                const uint32_t given = asking < left ? asking : left;
                stream.taken += given;
                waveRays += given;
            }
            continue;
        }

        if ( mLeaf && ( __popcll( mLeaf ) >= PT_WIDE_DEFER_LEAF_LANES || mTrav == 0 ) )
        {
            do
            {
                bool isTri = false, isSphere = false;
                if ( numLeaves != 0 )
                {
                    uint2 g = myLeaves[( numLeaves - 1u ) * kThreads];
                    const uint32_t k = static_cast< uint32_t >( __builtin_ctz( g.y & 0xFFu ) );
                    g.y &= ~( 1u << k );
                    if ( ( g.y & 0xFFu ) == 0 )
                        numLeaves--;
                    else
                        myLeaves[( numLeaves - 1u ) * kThreads].y = g.y;
                    // the reciprocal direction itself, for the reference's box test: invW but for the infinities it replaced
                    // (a ray with a component WideInv takes out never gets here)
                    const f3 inv = F3( fabsf( invW.x ) == kWideInvLimit ? __builtin_copysignf( INFINITY, invW.x ) : invW.x,
                                       fabsf( invW.y ) == kWideInvLimit ? __builtin_copysignf( INFINITY, invW.y ) : invW.y,
                                       fabsf( invW.z ) == kWideInvLimit ? __builtin_copysignf( INFINITY, invW.z ) : invW.z );
                    ClosestPrimTestWide< kSpheres >( sv, sb.prims, g.x + __popc( ( g.y >> 8 ) & ( ( 1u << k ) - 1u ) ), o, d, inv, nx, ny, nz, h, &isSphere, &ambiguous );
                    isTri = !isSphere;
                }
                waveTris += LanesSet( isTri );
                waveSpheres += LanesSet( isSphere );
            } while ( LanesSet( numLeaves != 0 ) >= PT_WIDE_DEFER_KEEP_LANES );
            continue;
        }

        const uint32_t started = static_cast< uint32_t >( __popcll( mTrav ) );
        for ( int burst = 0; burst < PT_WIDE_BURST; ++burst )
        {
            const bool go = sp != 0 && numLeaves < kWideDeferred;
            const uint32_t active = LanesSet( go );
            if ( burst && 2 * active < started ) break;
            waveNodes += active;
            waveSteps++;
            if ( go )
            {
                // the youngest entry: visit the nearest of its children, drop the entry if that was the last.  LDS is read
                // without asking where the entry lives and the rare deeper entry fetched on top: a pointer that may point
                // to either home makes every access a flat one.
                const uint32_t top = sp - 1u;
                uint2 e = myLds[( top < kWideLdsStack ? top : kWideLdsStack - 1u ) * kThreads];
                asm volatile( "" : "+v"( e.x ), "+v"( e.y ) ); // the LDS read stays an LDS read
                if ( top >= kWideLdsStack ) e = spill[( top - kWideLdsStack ) * spillStride + spillLane];
                const uint32_t s = pickLds[( e.y & 0xFFu ) | pickRow];
                const uint32_t left = e.y & ~( 1u << s );
                if ( ( left & 0xFFu ) == 0 )
                    sp = top;
                else if ( top < kWideLdsStack )
                    myLds[top * kThreads].y = left;
                else
                    spill[( top - kWideLdsStack ) * spillStride + spillLane].y = left;
                const WideGroup g = WideVisit< kSpheres >( sb.nodes, e.x + __popc( ( e.y >> 8 ) & ( ( 1u << s ) - 1u ) ), o, invW, nx, ny, nz, h.t * kBand );
                if ( g.bits & 0xFF0000u ) myLeaves[numLeaves++ * kThreads] = make_uint2( g.primBase, g.bits >> 16 );
                if ( g.bits & 0xFFu )
                {
                    if ( sp < kWideLdsStack )
                        myLds[sp * kThreads] = make_uint2( g.childBase, g.bits & 0xFFFFu );
                    else
                        spill[( sp - kWideLdsStack ) * spillStride + spillLane] = make_uint2( g.childBase, g.bits & 0xFFFFu );
                    sp++;
                }
            }
        }
    }
    FlushWaveCounter( stats, ST_RAYS_CLOSEST, waveRays );
    FlushWaveCounter( stats, ST_NODES_CLOSEST, waveNodes );
    FlushWaveCounter( stats, ST_TRI_CLOSEST, waveTris );
    FlushWaveCounter( stats, ST_SPH_CLOSEST, waveSpheres );
    FlushWaveCounter( stats, ST_STEPS_CLOSEST, waveSteps );
}

// Visibility of the queued shadow rays: BVH::Occluded (src/bvh.cpp:270-310) under Scene::Occluded
// (src/scene.cpp:381-384), same scheduling as ExtendKernel.  One work item per shadow record; an occluded
// record gets the distance -2 written back, so that ResolveKernel leaves it out of the sum.  kDirectSum: the
// scene has one light with one sample, so a path has at most one record per bounce and its record holds
// throughput * LDirect and the path id (ShadeKernel): an unoccluded record is added to the path's radiance
// here (path_tracer.cpp:133-134) and ResolveKernel does not run.
template < bool kDirectSum >
__global__ __launch_bounds__( kConnectBlock ) PT_CONNECT_OCCUPANCY void ConnectKernel( SceneView sv, PathBuffers pb, int bounce, uint64_t* stats )
{
    UnitStream stream{ pb.cursors + 2 * bounce + 1, UnitCount( pb, bounce, kCountShadowRays, 0 ), pb.units, kUnitPaths * pb.shadowSlots, 0u };
    const uint32_t end      = sv.numNodes;
    const uint32_t entries  = pb.units * kUnitPaths;
    const SceneBuffers sb   = MakeSceneBuffers< false >( sv );
    __shared__ float4 topLds[kLdsTopNodes > 0 ? 2 * kLdsTopNodes : 1];
    const uint32_t topCount = kLdsTopNodes < sv.numTopNodes ? kLdsTopNodes : sv.numTopNodes;
    if ( kLdsTopNodes > 0 )
    {
        for ( uint32_t i = threadIdx.x; i < 2 * topCount; i += kConnectBlock ) topLds[i] = Load16( sb.nodes, 16u * i );
        __syncthreads();
    }
    uint32_t rec = kNoRay, node = end, leaf = kInnerNode;
    f3 o = F3( 0.0f, 0.0f, 0.0f ), d = o, inv = o;
    float tMax = 0.0f;
    bool nx = false, ny = false, nz = false, occluded = false;
    uint32_t waveRays = 0, waveNodes = 0, waveTris = 0, waveSpheres = 0, waveSteps = 0; // whole-wave totals, wave-uniform

    for ( ;; )
    {
        const bool atLeaf     = leaf != kInnerNode;
        const bool idle       = node == end && !atLeaf; // a lane wants a ray if it is idle and has a result to store or the stream goes on
        // masks built from single compares (a ballot of a compound predicate costs two more vector instructions)
        const unsigned long long mLeaf = Ballot( atLeaf ), mEnd = Ballot( node == end );
        const unsigned long long mTrav = ~mEnd & ~mLeaf;
        const unsigned long long mRay  = mEnd & ~mLeaf & ( stream.ended ? Ballot( rec != kNoRay ) : ~0ull );
        if ( ( mLeaf | mTrav | mRay ) == 0 ) break;

        if ( mRay && ( __popcll( mRay ) >= kConnectRefillLanes || ( mTrav | mLeaf ) == 0 ) )
        {
            if constexpr ( kDirectSum )
            {
                if ( idle && rec != kNoRay && !occluded )
                {
                    const float4 e = pb.shEst[rec];
                    const uint32_t p = __float_as_uint( e.w );
                    pb.rad[p] = F4( F3( pb.rad[p] ) + F3( e ), 0.0f );
                }
            }
            else if ( idle && rec != kNoRay && occluded )
                reinterpret_cast< float* >( pb.shDir )[4ull * rec + 3] = -2.0f;
            if ( idle ) rec = kNoRay;
            while ( !stream.ended )
            {
                const unsigned long long mIdle = Ballot( idle && rec == kNoRay );
                if ( mIdle == 0 ) break;
                if ( stream.taken == stream.count )
                {
                    NextUnit( stream );
                    continue;
                }
                const uint32_t left = stream.count - stream.taken, asking = static_cast< uint32_t >( __popcll( mIdle ) );
                const uint32_t rank = LanesBelow( mIdle );
                if ( idle && rec == kNoRay && rank < left )
                {
                    rec = pb.shadowRays[stream.base + stream.taken + rank];
                    uint32_t entry = pb.shadowSlots == 1 ? rec : rec % entries; // record index = slot * entries + entry
                    asm volatile( "" : "+v"( entry ) );                              // the index first, then both loads in one go
                    float4 dir = pb.shDir[rec];
                    Packed3 org = pb.shadowO[entry];
                    asm volatile( "" : "+v"( dir.x ), "+v"( org.x ) ); // both loads in flight before either is waited for
                    o    = F3( org );
                    d    = F3( dir );
                    tMax = dir.w;
                    inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:275-276
                    nx = inv.x < 0; ny = inv.y < 0; nz = inv.z < 0;
                    occluded = false;
                    node     = 0;
                }
                const uint32_t given = asking < left ? asking : left;
                stream.taken += given;
                waveRays += given;
            }
            continue;
        }

        if ( mLeaf && ( __popcll( mLeaf ) >= kLeafLanes || mTrav == 0 ) )
        {
            const uint32_t first = leaf & kLeafFirstMask;
            uint32_t shapes = atLeaf ? ( leaf >> 28 ) + 1 : 0;
            for ( uint32_t k = 0; Ballot( k < shapes ) != 0; ++k ) // bvh.cpp:287-293: stop at the first hit
            {
                bool isTri = false, isSphere = false;
                if ( k < shapes )
                {
                    if ( AnyPrimTest( sb, first + k, o, d, tMax, &isSphere ) )
                    {
                        occluded = true;
                        shapes   = 0;
                        node     = end;
                    }
                    isTri = !isSphere;
                }
                waveTris += LanesSet( isTri );
                waveSpheres += LanesSet( isSphere );
            }
            leaf = kInnerNode;
            continue;
        }

        const uint32_t started = static_cast< uint32_t >( __popcll( mTrav ) );
        for ( int burst = 0; burst < kTraverseBurst; ++burst )
        {
            const bool go = node != end && leaf == kInnerNode;
            const uint32_t active = static_cast< uint32_t >( __popcll( Ballot( node != end ) & Ballot( leaf == kInnerNode ) ) );
            if ( burst && 2 * active < started ) break;
            waveNodes += active;
            waveSteps++;
            if ( go )
            {
                float4 a, b;
                if ( kLdsTopNodes > 0 && node < topCount )
                {
                    a = LoadLds16( topLds + 2 * node );
                    b = LoadLds16( topLds + 2 * node + 1 );
                }
                else
                {
                    a = Load16( sb.nodes, 32u * node );
                    b = Load16( sb.nodes, 32u * node + 16u );
                }
                const uint32_t link = __float_as_uint( b.z ), word = __float_as_uint( b.w );
                const bool boxHit = RayBox( o, inv, nx, ny, nz, a, b, tMax );
                if ( boxHit && !( link & kLeafFlag ) )
                    node = word; // first child
                else
                {
                    node = link & ~kLeafFlag;
                    if ( boxHit ) leaf = word;
                }
            }
        }
    }
    FlushWaveCounter( stats, ST_RAYS_SHADOW, waveRays );
    FlushWaveCounter( stats, ST_NODES_SHADOW, waveNodes );
    FlushWaveCounter( stats, ST_TRI_SHADOW, waveTris );
    FlushWaveCounter( stats, ST_SPH_SHADOW, waveSpheres );
    FlushWaveCounter( stats, ST_STEPS_SHADOW, waveSteps );
}

// ConnectKernel over the wide layout: any hit, with ExtendWideKernel's stack and scheduling.  The order of the search
// does not matter to the answer: a primitive counts only if the reference's own test of its leaf box against the ray's
// fixed tMax passes (AnyPrimTestWide), which is the condition under which BVH::Occluded reaches it.
#ifndef PT_WIDE_SHADOW_DEFERRED
#define PT_WIDE_SHADOW_DEFERRED 2
#endif
#ifndef PT_WIDE_SHADOW_KEEP_LANES
#define PT_WIDE_SHADOW_KEEP_LANES 12
#endif
constexpr uint32_t kWideShadowDeferred = PT_WIDE_SHADOW_DEFERRED;

// kOneDirection (pb.constantShadowDir): every ray has the direction pb.shadowDir and no end - the direction, its
// reciprocals and its signs are the same in every lane, and a refill reads the ray's origin only.
template < bool kDirectSum, bool kOneDirection = false >
__global__ __launch_bounds__( kWideBlock ) __attribute__( ( amdgpu_waves_per_eu( PT_WIDE_WAVES, PT_WIDE_WAVES ) ) )
void ConnectWideKernel( SceneView sv, PathBuffers pb, int bounce, uint2* spill, uint64_t* stats )
{
    constexpr uint32_t kThreads = kWideBlock;
    UnitStream stream{ pb.cursors + 2 * bounce + 1, UnitCount( pb, bounce, kCountShadowRays, 0 ), pb.units, kUnitPaths * pb.shadowSlots, 0u };
    const uint32_t entries = pb.units * kUnitPaths;
    const SceneBuffers sb  = MakeWideBuffers( sv );
    __shared__ uint2 stackLds[kWideLdsStack * kThreads];
    uint2* const myLds = stackLds + threadIdx.x;
    __shared__ uint2 leafLds[kWideShadowDeferred * kThreads]; // leaf children found and not tested yet, as in ExtendWideKernel
    uint2* const myLeaves = leafLds + threadIdx.x;
    uint32_t numLeaves = 0;
    const size_t spillStride = static_cast< size_t >( gridDim.x ) * kThreads, spillLane = static_cast< size_t >( blockIdx.x ) * kThreads + threadIdx.x;
    uint32_t rec = kNoRay;
    f3 o = F3( 0.0f, 0.0f, 0.0f ), d = o, inv = o, invW = o;
    float tMax = 0.0f;
    bool nx = false, ny = false, nz = false, occluded = false;
    if constexpr ( kOneDirection )
    {
        d    = F3( pb.shadowDir[0], pb.shadowDir[1], pb.shadowDir[2] );
        tMax = FLT_MAX;                                       // src/lights.cpp:31
        inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:275-276
        nx = inv.x < 0; ny = inv.y < 0; nz = inv.z < 0;
        invW = WideInv( inv );
    }
    uint32_t sp = 0;
    uint32_t waveRays = 0, waveNodes = 0, waveTris = 0, waveSpheres = 0, waveSteps = 0; // whole-wave totals, wave-uniform

    for ( ;; )
    {
        const bool idle = ( numLeaves | sp ) == 0;
        const unsigned long long mLeaf = Ballot( numLeaves != 0 ), mIdle = Ballot( idle );
        const unsigned long long mTrav = Ballot( sp != 0 ) & Ballot( numLeaves < kWideShadowDeferred );
        const unsigned long long mRay  = mIdle & ( stream.ended ? Ballot( rec != kNoRay ) : ~0ull );
        if ( ( mLeaf | mTrav | mRay ) == 0 ) break;

        if ( mRay && ( __popcll( mRay ) >= PT_WIDE_SHADOW_REFILL_LANES || ( mTrav | mLeaf ) == 0 ) )
        {
            if constexpr ( kDirectSum )
            {
                if ( idle && rec != kNoRay && !occluded )
                {
                    const float4 e = pb.shEst[rec];
                    const uint32_t p = __float_as_uint( e.w );
                    pb.rad[p] = F4( F3( pb.rad[p] ) + F3( e ), 0.0f );
                }
            }
            else if ( idle && rec != kNoRay && occluded )
                reinterpret_cast< float* >( pb.shDir )[4ull * rec + 3] = -2.0f;
            if ( idle ) rec = kNoRay;
            while ( !stream.ended )
            {
                const unsigned long long mAsk = Ballot( idle && rec == kNoRay );
                if ( mAsk == 0 ) break;
                if ( stream.taken == stream.count )
                {
                    NextUnit( stream );
                    continue;
                }
                const uint32_t left = stream.count - stream.taken, asking = static_cast< uint32_t >( __popcll( mAsk ) );
                const uint32_t rank = LanesBelow( mAsk );
                if ( idle && rec == kNoRay && rank < left )
                {
                    rec = pb.shadowRays[stream.base + stream.taken + rank];
                    uint32_t entry = pb.shadowSlots == 1 ? rec : rec % entries; // record index = slot * entries + entry
                    if constexpr ( kOneDirection )
                        o = F3( pb.shadowO[entry] );
                    else
                    {
                        asm volatile( "" : "+v"( entry ) );                              // the index first, then both loads in one go
                        float4 dir = pb.shDir[rec];
                        Packed3 org = pb.shadowO[entry];
                        asm volatile( "" : "+v"( dir.x ), "+v"( org.x ) ); // both loads in flight before either is waited for
                        o    = F3( org );
                        d    = F3( dir );
                        tMax = dir.w;
                        inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:275-276
                        nx = inv.x < 0; ny = inv.y < 0; nz = inv.z < 0;
                        invW = WideInv( inv );
                    }
                    occluded = false;
                    sp       = 1u;
                    myLds[0] = make_uint2( 0u, 0x0101u ); // the root as the one child of a node above it
                }
                const uint32_t given = asking < left ? asking : left;
                stream.taken += given;
                waveRays += given;
            }
            continue;
        }

        if ( mLeaf && ( __popcll( mLeaf ) >= PT_WIDE_SHADOW_LEAF_LANES || mTrav == 0 ) )
        {
            do
            {
                bool isTri = false, isSphere = false;
                if ( numLeaves != 0 )
                {
                    uint2 g = myLeaves[( numLeaves - 1u ) * kThreads];
                    const uint32_t k = static_cast< uint32_t >( __builtin_ctz( g.y & 0xFFu ) );
                    g.y &= ~( 1u << k );
                    if ( ( g.y & 0xFFu ) == 0 )
                        numLeaves--;
                    else
                        myLeaves[( numLeaves - 1u ) * kThreads].y = g.y;
                    if ( AnyPrimTestWide( sb.prims, g.x + __popc( ( g.y >> 8 ) & ( ( 1u << k ) - 1u ) ), o, d, inv, nx, ny, nz, tMax, &isSphere ) )
                    {
                        occluded  = true; // bvh.cpp:287-293: stop at the first hit
                        numLeaves = 0;
                        sp        = 0;
                    }
                    isTri = !isSphere;
                }
                waveTris += LanesSet( isTri );
                waveSpheres += LanesSet( isSphere );
            } while ( LanesSet( numLeaves != 0 ) >= PT_WIDE_SHADOW_KEEP_LANES );
            continue;
        }

        const uint32_t started = static_cast< uint32_t >( __popcll( mTrav ) );
        for ( int burst = 0; burst < PT_WIDE_SHADOW_BURST; ++burst )
        {
            const bool go = sp != 0 && numLeaves < kWideShadowDeferred;
            const uint32_t active = LanesSet( go );
            if ( burst && 2 * active < started ) break;
            waveNodes += active;
            waveSteps++;
            if ( go )
            {
                const uint32_t top = sp - 1u;
                uint2 e = myLds[( top < kWideLdsStack ? top : kWideLdsStack - 1u ) * kThreads];
                asm volatile( "" : "+v"( e.x ), "+v"( e.y ) ); // the LDS read stays an LDS read
                if ( top >= kWideLdsStack ) e = spill[( top - kWideLdsStack ) * spillStride + spillLane];
                // any hit: the order is free, and the lowest slot costs a tenth of what the nearest does (per launch on a
                // Sponza-class scene 13.75 instead of 14.12 ms, Cornell box 73.1 instead of 75.1, knot 5.8 instead of 6.15)
                const uint32_t s = static_cast< uint32_t >( __builtin_ctz( e.y & 0xFFu ) );
                const uint32_t left = e.y & ~( 1u << s );
                if ( ( left & 0xFFu ) == 0 )
                    sp = top;
                else if ( top < kWideLdsStack )
                    myLds[top * kThreads].y = left;
                else
                    spill[( top - kWideLdsStack ) * spillStride + spillLane].y = left;
                const WideGroup g = WideVisit< false >( sb.nodes, e.x + __popc( ( e.y >> 8 ) & ( ( 1u << s ) - 1u ) ), o, invW, nx, ny, nz, tMax );
                if ( g.bits & 0xFF0000u ) myLeaves[numLeaves++ * kThreads] = make_uint2( g.primBase, g.bits >> 16 );
                if ( g.bits & 0xFFu )
                {
                    if ( sp < kWideLdsStack )
                        myLds[sp * kThreads] = make_uint2( g.childBase, g.bits & 0xFFFFu );
                    else
                        spill[( sp - kWideLdsStack ) * spillStride + spillLane] = make_uint2( g.childBase, g.bits & 0xFFFFu );
                    sp++;
                }
            }
        }
    }
    FlushWaveCounter( stats, ST_RAYS_SHADOW, waveRays );
    FlushWaveCounter( stats, ST_NODES_SHADOW, waveNodes );
    FlushWaveCounter( stats, ST_TRI_SHADOW, waveTris );
    FlushWaveCounter( stats, ST_SPH_SHADOW, waveSpheres );
    FlushWaveCounter( stats, ST_STEPS_SHADOW, waveSteps );
}

// ---- the wide tree in the light's frame ------------------------------------------------------------------------
//
// Shadow rays of one directional light over the nodes of the wide layout (same children, same indices), each child kept
// as ConnectParallelKernel keeps a node of the binary tree: the rectangle around its geometry on the light's grid, and
// how far it reaches along the rays' main axis - here in bytes, relative to the node (layout_build.hip,
// LightWideNodesKernel).  A node visit is four 16-byte loads and, per child, four integer compares and one more for the
// reach, where the box test of ConnectWideKernel converts six bytes and computes three packed FMAs, a max3 and a min3.
// Outer bounds again: what is found counts once the reference's own test of the primitive's leaf box has passed.
// The rectangle: a ray's cell c lies in [first, last] (cells) if and only if ( c - origin ) >> shift lies in
// [( first - origin ) >> shift, ( last - origin ) >> shift] or between values that map to the same bytes - the shift is
// monotone, so nothing inside is lost; a cell below the origin wraps to a large number and fails the upper compare.
// The reach: a child whose far plane lies at `far` can only hold what the ray meets if far > pw.  The record has
// byte = ceil( ( far - d0 ) / step ), the ray computes t = fl( pw - d0 ) / step (step is a power of two); for |t| < 2^23
// the rounding of the difference moves t by less than 1 / 2, so far > pw implies byte >= floor( t ); beyond 2^23 the
// sign of t decides, rightly.  The child passes if byte >= clamp( floor( t ), 0, 256 ).
__device__ __forceinline__ uint32_t ByteOf( uint32_t w, int k ) { return ( w >> ( 8 * k ) ) & 0xFFu; }

// scheduling of ConnectLightWideKernel, as PT_WIDE_SHADOW_* for ConnectWideKernel (per launch on a Sponza-class scene:
// refill at 24 / 32 / 40 idle lanes with bursts of 3: 11.25 / 11.14 / 11.10 ms, bursts of 2 / 3 / 4: 11.32 / 11.14 / 11.29)
#ifndef PT_LIGHT_WIDE_REFILL_LANES
#define PT_LIGHT_WIDE_REFILL_LANES 32
#endif
#ifndef PT_LIGHT_WIDE_LEAF_LANES
#define PT_LIGHT_WIDE_LEAF_LANES 20
#endif
#ifndef PT_LIGHT_WIDE_KEEP_LANES
#define PT_LIGHT_WIDE_KEEP_LANES 12
#endif
#ifndef PT_LIGHT_WIDE_BURST
#define PT_LIGHT_WIDE_BURST 3
#endif
#ifndef PT_LIGHT_WIDE_DEFERRED
#define PT_LIGHT_WIDE_DEFERRED 2
#endif
constexpr uint32_t kLightWideDeferred = PT_LIGHT_WIDE_DEFERRED;

__device__ __forceinline__ WideGroup LightWideVisit( BufferDesc nodes, uint32_t node, uint32_t cellU, uint32_t cellV, float pw )
{
    const uint32_t at = 64u * node;
    const float4 n0 = Load16( nodes, at ), n1 = Load16( nodes, at + 16u ), n2 = Load16( nodes, at + 32u ), n3 = Load16( nodes, at + 48u );
    const uint32_t w0 = __float_as_uint( n0.x ), w1 = __float_as_uint( n0.y );
    const uint32_t lu = ( cellU - ( w0 & 0xFFFFu ) ) >> ( w1 & 0xFFu ), lv = ( cellV - ( w0 >> 16 ) ) >> ( ( w1 >> 8 ) & 0xFFu );
    const float t = floorf( ( pw - n1.x ) * n1.y );
    const uint32_t reach = static_cast< uint32_t >( __builtin_amdgcn_fmed3f( t, 0.0f, 256.0f ) ); // a NaN (no finite origin) passes as 0
    const uint32_t far[2] = { __float_as_uint( n1.z ), __float_as_uint( n1.w ) }, u0[2] = { __float_as_uint( n2.x ), __float_as_uint( n2.y ) },
                   u1[2] = { __float_as_uint( n2.z ), __float_as_uint( n2.w ) }, v0[2] = { __float_as_uint( n3.x ), __float_as_uint( n3.y ) },
                   v1[2] = { __float_as_uint( n3.z ), __float_as_uint( n3.w ) };
    uint32_t hits = 0;
#pragma unroll
    for ( int s = 0; s < 8; ++s )
    {
        const int w = s >> 2, k = s & 3;
        const bool in = ( ByteOf( u0[w], k ) <= lu ) & ( lu <= ByteOf( u1[w], k ) ) & ( ByteOf( v0[w], k ) <= lv ) & ( lv <= ByteOf( v1[w], k ) ) & ( ByteOf( far[w], k ) >= reach );
        hits |= in ? 1u << s : 0u;
    }
    const uint32_t imask = ( w1 >> 16 ) & 0xFFu, lmask = w1 >> 24;
    return WideGroup{ __float_as_uint( n0.z ), __float_as_uint( n0.w ), ( hits & imask ) | imask << 8 | ( hits & lmask ) << 16 | lmask << 24 };
}

// ConnectWideKernel< true, true > over those records: the same stack, lists and scheduling; a lane keeps its ray as two
// cells, the start along the main axis and the origin (for the primitive tests, which run in world space as ever).
__global__ __launch_bounds__( kWideBlock ) __attribute__( ( amdgpu_waves_per_eu( PT_WIDE_WAVES, PT_WIDE_WAVES ) ) )
void ConnectLightWideKernel( SceneView sv, PathBuffers pb, int bounce, uint2* spill, uint64_t* stats )
{
    constexpr uint32_t kThreads = kWideBlock;
    UnitStream stream{ pb.cursors + 2 * bounce + 1, UnitCount( pb, bounce, kCountShadowRays, 0 ), pb.units, kUnitPaths * pb.shadowSlots, 0u };
    const uint32_t entries = pb.units * kUnitPaths;
    const BufferDesc nodes = MakeBuffer( sv.lightWideNodes, sv.numWideNodes * 64u ), prims = MakeBuffer( sv.widePrims, sv.numPrims * 64u );
    __shared__ uint2 stackLds[kWideLdsStack * kThreads];
    uint2* const myLds = stackLds + threadIdx.x;
    __shared__ uint2 leafLds[kLightWideDeferred * kThreads];
    uint2* const myLeaves = leafLds + threadIdx.x;
    uint32_t numLeaves = 0;
    const size_t spillStride = static_cast< size_t >( gridDim.x ) * kThreads, spillLane = static_cast< size_t >( blockIdx.x ) * kThreads + threadIdx.x;
    uint32_t rec = kNoRay;
    const f3 d    = F3( pb.shadowDir[0], pb.shadowDir[1], pb.shadowDir[2] );
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:275-276
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    constexpr float tMax = FLT_MAX;                            // src/lights.cpp:35
    const f3 axisU = F3( sv.lightU ), axisV = F3( sv.lightV );
    f3 o = F3( 0.0f, 0.0f, 0.0f );
    uint32_t cellU = 0, cellV = 0;
    float pw = 0.0f;
    bool occluded = false;
    uint32_t sp = 0;
    uint32_t waveRays = 0, waveNodes = 0, waveTris = 0, waveSpheres = 0, waveSteps = 0; // whole-wave totals, wave-uniform

    for ( ;; )
    {
        const bool idle = ( numLeaves | sp ) == 0;
        const unsigned long long mLeaf = Ballot( numLeaves != 0 ), mIdle = Ballot( idle );
        const unsigned long long mTrav = Ballot( sp != 0 ) & Ballot( numLeaves < kLightWideDeferred );
        const unsigned long long mRay  = mIdle & ( stream.ended ? Ballot( rec != kNoRay ) : ~0ull );
        if ( ( mLeaf | mTrav | mRay ) == 0 ) break;

        if ( mRay && ( __popcll( mRay ) >= PT_LIGHT_WIDE_REFILL_LANES || ( mTrav | mLeaf ) == 0 ) )
        {
            if ( idle && rec != kNoRay && !occluded )
            {
                const float4 e = pb.shEst[rec];
                const uint32_t p = __float_as_uint( e.w );
                pb.rad[p] = F4( F3( pb.rad[p] ) + F3( e ), 0.0f );
            }
            if ( idle ) rec = kNoRay;
            while ( !stream.ended )
            {
                const unsigned long long mAsk = Ballot( idle && rec == kNoRay );
                if ( mAsk == 0 ) break;
                if ( stream.taken == stream.count )
                {
                    NextUnit( stream );
                    continue;
                }
                const uint32_t left = stream.count - stream.taken, asking = static_cast< uint32_t >( __popcll( mAsk ) );
                const uint32_t rank = LanesBelow( mAsk );
                if ( idle && rec == kNoRay && rank < left )
                {
                    rec = pb.shadowRays[stream.base + stream.taken + rank];
                    const uint32_t entry = pb.shadowSlots == 1 ? rec : rec % entries; // record index = slot * entries + entry
                    o = F3( pb.shadowO[entry] );
                    // the cell the ray's point falls into, as in ConnectParallelKernel (cells 0 and 65535 lie outside every rectangle)
                    const float gu = ( Dot( o, axisU ) - sv.shadowOrigin[0] ) * sv.shadowInvCell[0];
                    const float gv = ( Dot( o, axisV ) - sv.shadowOrigin[1] ) * sv.shadowInvCell[1];
                    cellU = static_cast< uint32_t >( __builtin_fminf( __builtin_fmaxf( gu, 0.0f ), 65535.0f ) );
                    cellV = static_cast< uint32_t >( __builtin_fminf( __builtin_fmaxf( gv, 0.0f ), 65535.0f ) );
                    pw = ( sv.shadowDepthAxis == 0 ? o.x : sv.shadowDepthAxis == 1 ? o.y : o.z ) * sv.shadowDepthSign;
                    occluded = false;
                    sp       = 1u;
                    myLds[0] = make_uint2( 0u, 0x0101u ); // the root as the one child of a node above it
                }
                const uint32_t given = asking < left ? asking : left;
                stream.taken += given;
                waveRays += given;
            }
            continue;
        }

        if ( mLeaf && ( __popcll( mLeaf ) >= PT_LIGHT_WIDE_LEAF_LANES || mTrav == 0 ) )
        {
            do
            {
                bool isTri = false, isSphere = false;
                if ( numLeaves != 0 )
                {
                    uint2 g = myLeaves[( numLeaves - 1u ) * kThreads];
                    const uint32_t k = static_cast< uint32_t >( __builtin_ctz( g.y & 0xFFu ) );
                    g.y &= ~( 1u << k );
                    if ( ( g.y & 0xFFu ) == 0 )
                        numLeaves--;
                    else
                        myLeaves[( numLeaves - 1u ) * kThreads].y = g.y;
                    if ( AnyPrimTestWide( prims, g.x + __popc( ( g.y >> 8 ) & ( ( 1u << k ) - 1u ) ), o, d, inv, nx, ny, nz, tMax, &isSphere ) )
                    {
                        occluded  = true; // bvh.cpp:287-293: stop at the first hit
                        numLeaves = 0;
                        sp        = 0;
                    }
                    isTri = !isSphere;
                }
                waveTris += LanesSet( isTri );
                waveSpheres += LanesSet( isSphere );
            } while ( LanesSet( numLeaves != 0 ) >= PT_LIGHT_WIDE_KEEP_LANES );
            continue;
        }

        const uint32_t started = static_cast< uint32_t >( __popcll( mTrav ) );
        for ( int burst = 0; burst < PT_LIGHT_WIDE_BURST; ++burst )
        {
            const bool go = sp != 0 && numLeaves < kLightWideDeferred;
            const uint32_t active = LanesSet( go );
            if ( burst && 2 * active < started ) break;
            waveNodes += active;
            waveSteps++;
            if ( go )
            {
                const uint32_t top = sp - 1u;
                uint2 e = myLds[( top < kWideLdsStack ? top : kWideLdsStack - 1u ) * kThreads];
                asm volatile( "" : "+v"( e.x ), "+v"( e.y ) ); // the LDS read stays an LDS read
                if ( top >= kWideLdsStack ) e = spill[( top - kWideLdsStack ) * spillStride + spillLane];
                const uint32_t s = static_cast< uint32_t >( __builtin_ctz( e.y & 0xFFu ) ); // any hit: the order is free
                const uint32_t left = e.y & ~( 1u << s );
                if ( ( left & 0xFFu ) == 0 )
                    sp = top;
                else if ( top < kWideLdsStack )
                    myLds[top * kThreads].y = left;
                else
                    spill[( top - kWideLdsStack ) * spillStride + spillLane].y = left;
                const WideGroup g = LightWideVisit( nodes, e.x + __popc( ( e.y >> 8 ) & ( ( 1u << s ) - 1u ) ), cellU, cellV, pw );
                if ( g.bits & 0xFF0000u ) myLeaves[numLeaves++ * kThreads] = make_uint2( g.primBase, g.bits >> 16 );
                if ( g.bits & 0xFFu )
                {
                    if ( sp < kWideLdsStack )
                        myLds[sp * kThreads] = make_uint2( g.childBase, g.bits & 0xFFFFu );
                    else
                        spill[( sp - kWideLdsStack ) * spillStride + spillLane] = make_uint2( g.childBase, g.bits & 0xFFFFu );
                    sp++;
                }
            }
        }
    }
    FlushWaveCounter( stats, ST_RAYS_SHADOW, waveRays );
    FlushWaveCounter( stats, ST_NODES_SHADOW, waveNodes );
    FlushWaveCounter( stats, ST_TRI_SHADOW, waveTris );
    FlushWaveCounter( stats, ST_SPH_SHADOW, waveSpheres );
    FlushWaveCounter( stats, ST_STEPS_SHADOW, waveSteps );
}

// Shadow rays of a scene lit by one directional light (src/lights.cpp:25-38): they all have the light's direction,
// so BVH::Occluded (src/bvh.cpp:270-310) becomes a search in two dimensions.  Seen along the light's direction - lightU
// and lightV are two axes across it - a ray is a point, a node the rectangle around its geometry, and the node can
// only be entered if the point lies in the rectangle and the node's box reaches beyond the ray's start: its far plane
// along the world axis the rays mostly travel on lies ahead of the origin (the reference's own test on that axis).  The
// records of the third node layout (api.hip) hold exactly that, as an outer
// bound: 16 bytes per node visit instead of 32, four integer compares and a float compare instead of the slab
// arithmetic, and no per-ray reciprocal.  Being outer bounds, they may lead to a primitive that the reference's
// descent never reaches; the answer stays the reference's because a hit only counts if the reference's test of the
// primitive's leaf box passes (with it, the boxes being nested, those of all its ancestors): any-hit asks whether SOME
// primitive that the reference can reach is hit, and the order of the search does not matter.
// State per lane: the point as two grid cells, the start along that axis, and the origin for the primitive tests.
// (Two rays per lane, to have twice the loads in flight: 31.2 ms against 24.7 per 32 samples per pixel of the
// benchmark's frame - the kernel does not wait for single loads but for the memory pipeline as a whole.)
template < bool kDirectSum >
__global__ __launch_bounds__( kConnectBlock ) PT_CONNECT_OCCUPANCY void ConnectParallelKernel( SceneView sv, PathBuffers pb, int bounce, uint64_t* stats )
{
    UnitStream stream{ pb.cursors + 2 * bounce + 1, UnitCount( pb, bounce, kCountShadowRays, 0 ), pb.units, kUnitPaths * pb.shadowSlots, 0u };
    const uint32_t end      = sv.numShadowNodes;
    const uint32_t entries  = pb.units * kUnitPaths;
    const uint32_t topCount = sv.numShadowTop; // at most kShadowTopNodes
    const BufferDesc nodes  = MakeBuffer( sv.shadowNodes, sv.numShadowNodes * 16u );
    const SceneBuffers sb{ nodes, MakeBuffer( sv.prims, sv.numPrims * 64u ) };
    __shared__ float4 topLds[kShadowTopNodes];
    __shared__ uint32_t topWords[kShadowTopNodes];
    for ( uint32_t i = threadIdx.x; i < topCount; i += kConnectBlock )
    {
        topLds[i]   = Load16( nodes, 16u * i );
        topWords[i] = sv.shadowTopWords[i];
    }
    __syncthreads();
    // the one direction, as ShadeKernel wrote it into every record (wi = -direction), and what BVH::Occluded derives from it
    const f3 d    = -F3( sv.lights[0].a );
    const f3 inv  = F3( 1.0f / d.x, 1.0f / d.y, 1.0f / d.z ); // bvh.cpp:275-276
    const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
    constexpr float tMax = FLT_MAX;                            // src/lights.cpp:35
    const f3 axisU = F3( sv.lightU ), axisV = F3( sv.lightV );
    uint32_t rec = kNoRay, node = end, leaf = kInnerNode;
    f3 o = F3( 0.0f, 0.0f, 0.0f );
    uint32_t cell = 0; // the ray's point: its grid cell along u in the low half, along v in the high half
    float pw = 0.0f;   // where the ray starts on the world axis the rays mostly travel along, in their sense
    bool occluded = false;
    uint32_t waveRays = 0, waveNodes = 0, waveTris = 0, waveSpheres = 0, waveSteps = 0; // whole-wave totals, wave-uniform

    for ( ;; )
    {
        const bool atLeaf = leaf != kInnerNode;
        const bool idle   = node == end && !atLeaf;
        const unsigned long long mLeaf = Ballot( atLeaf ), mEnd = Ballot( node == end );
        const unsigned long long mTrav = ~mEnd & ~mLeaf;
        const unsigned long long mRay  = mEnd & ~mLeaf & ( stream.ended ? Ballot( rec != kNoRay ) : ~0ull );
        if ( ( mLeaf | mTrav | mRay ) == 0 ) break;

        if ( mRay && ( __popcll( mRay ) >= kConnectRefillLanes || ( mTrav | mLeaf ) == 0 ) )
        {
            if constexpr ( kDirectSum )
            {
                if ( idle && rec != kNoRay && !occluded )
                {
                    const float4 e = pb.shEst[rec];
                    const uint32_t p = __float_as_uint( e.w );
                    pb.rad[p] = F4( F3( pb.rad[p] ) + F3( e ), 0.0f );
                }
            }
            else if ( idle && rec != kNoRay && occluded )
                reinterpret_cast< float* >( pb.shDir )[4ull * rec + 3] = -2.0f;
            if ( idle ) rec = kNoRay;
            while ( !stream.ended )
            {
                const unsigned long long mIdle = Ballot( idle && rec == kNoRay );
                if ( mIdle == 0 ) break;
                if ( stream.taken == stream.count )
                {
                    NextUnit( stream );
                    continue;
                }
                const uint32_t left = stream.count - stream.taken, asking = static_cast< uint32_t >( __popcll( mIdle ) );
                const uint32_t rank = LanesBelow( mIdle );
                if ( idle && rec == kNoRay && rank < left )
                {
                    rec = pb.shadowRays[stream.base + stream.taken + rank];
                    const uint32_t entry = pb.shadowSlots == 1 ? rec : rec % entries; // record index = slot * entries + entry
                    o = F3( pb.shadowO[entry] );
                    // the cell the ray's point falls into (the conversion saturates; cells 0 and 65535 lie outside every rectangle)
                    const float gu = ( Dot( o, axisU ) - sv.shadowOrigin[0] ) * sv.shadowInvCell[0];
                    const float gv = ( Dot( o, axisV ) - sv.shadowOrigin[1] ) * sv.shadowInvCell[1];
                    cell = static_cast< uint32_t >( __builtin_fminf( __builtin_fmaxf( gu, 0.0f ), 65535.0f ) ) |
                           static_cast< uint32_t >( __builtin_fminf( __builtin_fmaxf( gv, 0.0f ), 65535.0f ) ) << 16;
                    pw = ( sv.shadowDepthAxis == 0 ? o.x : sv.shadowDepthAxis == 1 ? o.y : o.z ) * sv.shadowDepthSign;
                    occluded = false;
                    node     = 0;
                }
                const uint32_t given = asking < left ? asking : left;
                stream.taken += given;
                waveRays += given;
            }
            continue;
        }

        if ( mLeaf && ( __popcll( mLeaf ) >= kLeafLanes || mTrav == 0 ) )
        {
            const uint32_t first = leaf & kLeafFirstMask;
            uint32_t shapes = atLeaf ? ( leaf >> 28 ) + 1 : 0;
            for ( uint32_t k = 0; Ballot( k < shapes ) != 0; ++k ) // bvh.cpp:287-293: stop at the first hit
            {
                bool isTri = false, isSphere = false;
                if ( k < shapes )
                {
                    const uint32_t prim = first + k;
                    if ( AnyPrimTest( sb, prim, o, d, tMax, &isSphere ) )
                    {
                        // the reference reaches this primitive only through its leaf's box (bvh.cpp:282)
                        const float4 q1 = Load16( sb.prims, 64u * prim + 16u ), q2 = Load16( sb.prims, 64u * prim + 32u ), q3 = Load16( sb.prims, 64u * prim + 48u );
                        if ( RayBox( o, inv, nx, ny, nz, make_float4( q1.w, q2.w, q3.y, q3.z ), make_float4( q3.x, q3.w, 0.0f, 0.0f ), tMax ) )
                        {
                            occluded = true;
                            shapes   = 0;
                            node     = end;
                        }
                    }
                    isTri = !isSphere;
                }
                waveTris += LanesSet( isTri );
                waveSpheres += LanesSet( isSphere );
            }
            leaf = kInnerNode;
            continue;
        }

        const uint32_t started = static_cast< uint32_t >( __popcll( mTrav ) );
        for ( int burst = 0; burst < kTraverseBurst; ++burst )
        {
            const bool go = node != end && leaf == kInnerNode;
            const uint32_t active = static_cast< uint32_t >( __popcll( Ballot( node != end ) & Ballot( leaf == kInnerNode ) ) );
            if ( burst && 2 * active < started ) break;
            waveNodes += active;
            waveSteps++;
            if ( go )
            {
                float4 r;
                uint32_t second = node + 1u; // below the top: the first child of an inner node, the successor of a leaf
                if ( node < topCount )
                {
                    r      = LoadLds16( topLds + node );
                    second = topWords[node];
                }
                else
                    r = Load16( nodes, 16u * node );
                const uint32_t x = __float_as_uint( r.x ), y = __float_as_uint( r.y ), word = __float_as_uint( r.w );
                const bool inside = ( cell & 0xFFFFu ) - ( x & 0xFFFFu ) <= ( x >> 16 ) && ( cell >> 16 ) - ( y & 0xFFFFu ) <= ( y >> 16 ) && r.z > pw;
                const bool isLeaf = static_cast< int32_t >( word ) < 0; // kLeafFlag
                node = ( isLeaf || inside ) ? second : word;
                if ( isLeaf && inside ) leaf = word & ~kLeafFlag;
            }
        }
    }
    FlushWaveCounter( stats, ST_RAYS_SHADOW, waveRays );
    FlushWaveCounter( stats, ST_NODES_SHADOW, waveNodes );
    FlushWaveCounter( stats, ST_TRI_SHADOW, waveTris );
    FlushWaveCounter( stats, ST_SPH_SHADOW, waveSpheres );
    FlushWaveCounter( stats, ST_STEPS_SHADOW, waveSteps );
}

// ------------------------------------------------------------------------------------------------ shading

// What the light loop and the BRDF sampling need from a shaded hit
struct Surface
{
    f3 position, normal, tangent, bitangent;
    f3 F;  // BRDF::F = Kd / pi
    f3 wo;
};

// Shading of the closest hits (path_tracer.cpp:115-152).  Two instances: kSphereHits = false handles misses
// and triangle hits and defers the paths that hit a sphere to pb.sphereQueue; kSphereHits = true runs over
// that queue.  The sphere attributes need double precision atan2 / acos / sincos, whose registers would
// otherwise cap the occupancy of the triangle path that nearly every hit takes.
template < bool kSphereHits >
__global__ __launch_bounds__( kBlock ) void ShadeKernel( SceneView sv, RngParams rng, PathBuffers pb, int bounce, int maxDepth,
                                                        uint32_t width, uint32_t height, uint32_t sampleBegin, uint32_t numPaths, uint64_t* stats )
{
    const uint32_t numPixels = width * height;
    const int cur = bounce & 1, nxt = cur ^ 1;
    const uint32_t unit = WaveIndex(); // one unit per wave
    if ( unit >= pb.units ) return;
    const uint32_t segment = unit * kUnitPaths;
    // entries of this unit: its sphere hits, all of its paths (bounce 0), or its paths still alive
    const uint32_t* items = pb.sphereQueue + segment; // the sphere instance goes by this list of entries
    uint32_t limit;
    if ( kSphereHits )
        limit = *UnitCount( pb, bounce, kCountSphereHits, unit );
    else if ( bounce == 0 )
        limit = numPaths - segment < kUnitPaths ? numPaths - segment : kUnitPaths;
    else
        limit = *UnitCount( pb, bounce - 1, kCountNextPaths, unit );
    uint32_t* outShadowPaths = pb.shadowPaths + segment;
    uint32_t* outShadowRays  = pb.shadowRays + static_cast< size_t >( segment ) * pb.shadowSlots;
    // the sphere instance continues the segments the triangle instance has started
    uint32_t nNext        = kSphereHits ? *UnitCount( pb, bounce, kCountNextPaths, unit ) : 0u;
    uint32_t nShadowPaths = kSphereHits ? *UnitCount( pb, bounce, kCountShadowPaths, unit ) : 0u;
    uint32_t nShadowRays  = kSphereHits ? *UnitCount( pb, bounce, kCountShadowRays, unit ) : 0u;
    uint32_t nSpheres = 0, waveShaded = 0;
    const uint32_t drawBase = 2u + static_cast< uint32_t >( bounce ) * static_cast< uint32_t >( sv.drawsPerBounce );

    for ( uint32_t j = LaneIndex(); Ballot( j < limit ) != 0; j += 64u )
    {
        bool lit = false, deferred = false, alive = false, anyShadow = false;
        uint32_t p = 0, q = 0, pixel = 0, sample = 0; // path id, entry in this bounce's queue
        Surface sf{};
        f3 T = F3( 0.0f, 0.0f, 0.0f ), nextD = T, nextT = T;

        // hit attributes and material
        if ( j < limit )
        {
            q = kSphereHits ? items[j] : segment + j;
            p = bounce == 0 ? q : pb.queue[cur][q];
            const float4 hr = pb.hit[q];
            const int prim  = __float_as_int( hr.w );
            const f3 d = F3( pb.rayD[cur][q] );
            T    = bounce == 0 ? F3( 1.0f, 1.0f, 1.0f ) : F3( pb.thr[cur][q] ); // see RaygenKernel
            f3 L = F3( 0.0f, 0.0f, 0.0f ); // radiance so far: nothing at bounce 0; a hit at a later bounce does not touch it here

            if ( prim < 0 )
            {
                if ( bounce != 0 ) L = F3( pb.rad[p] );
                L = L + T * Environment( sv, d ); // path_tracer.cpp:116-120
                pb.rad[p] = F4( L, 0.0f );
            }
            else
            {
                const f3 o = bounce == 0 ? F3( pb.origin0[0], pb.origin0[1], pb.origin0[2] ) : F3( pb.rayO[cur][q] );
                const float t = hr.x, u = hr.y, v = hr.z;
                sf.position = o + d * t; // Ray::Evaluate
                float tu, tv;
                int material;
                if constexpr ( !kSphereHits )
                {
                    // Triangle::Intersect attribute interpolation (src/shapes.cpp:123-130).  The attribute record
                    // is fetched together with the tag; for a sphere primitive it is all zeros and unused.
                    const float4* a = sv.triShade + 8 * static_cast< size_t >( prim );
                    const float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6];
                    deferred = sv.numSpheres != 0 && __float_as_uint( sv.prims[4 * prim].w ) != kTagTriangle;
                    const f3 n0 = F3( a0.x, a0.y, a0.z ), n1 = F3( a0.w, a1.x, a1.y ), n2 = F3( a1.z, a1.w, a2.x );
                    const f3 t0 = F3( a2.y, a2.z, a2.w ), t1 = F3( a3.x, a3.y, a3.z ), t2 = F3( a3.w, a4.x, a4.y );
                    const float w = 1 - u - v;
                    sf.normal    = Normalize( n0 * w + n1 * u + n2 * v );
                    sf.tangent   = Normalize( t0 * w + t1 * u + t2 * v );
                    sf.bitangent = Cross( sf.normal, sf.tangent );
                    tu        = ( w * a4.z + u * a5.x ) + v * a5.z;
                    tv        = ( w * a4.w + u * a5.y ) + v * a5.w;
                    material  = __float_as_int( a6.x );
                }
                else
                {
                    // Sphere::Intersect (src/shapes.cpp:61-73).  atan2 / sin / cos are the double functions in
                    // the reference's build (accurate to a fraction of a double ulp on either side, so the rounded
                    // floats agree but for about one call in a billion), acosf the float one.
                    const float4 q0 = sv.prims[4 * prim], q1 = sv.prims[4 * prim + 1];
                    const float4* s = sv.spheres + 6 * __float_as_uint( q1.y );
                    const float4 c0 = s[0], c1 = s[1], c2 = s[2], c3 = s[3];
                    sf.normal = Normalize( sf.position - F3( q0 ) );
                    f3 lo     = MatMul( c0, c1, c2, c3, o, 1.0f );
                    f3 ld     = MatMul( c0, c1, c2, c3, d, 0.0f );
                    f3 lp     = lo + ld * t;
                    float theta = static_cast< float >( atan2( static_cast< double >( lp.z ), static_cast< double >( lp.x ) ) );
                    float phi   = AcosLibm( -lp.y ); // the C library's acosf, which is not correctly rounded
                    tu        = -0.5f * ( theta / kPi + 1 );
                    tv        = phi / kPi;
                    double ds, dc;
                    sincos( static_cast< double >( theta ), &ds, &dc );
                    sf.tangent   = F3( static_cast< float >( -ds ), 0.0f, static_cast< float >( dc ) );
                    sf.bitangent = Cross( sf.normal, sf.tangent );
                    material  = __float_as_int( s[5].x );
                }
                if ( !deferred )
                {
                    lit         = true;
                    sf.wo       = -d;
                    sf.position = sf.position + sf.normal * kEpsilon; // path_tracer.cpp:122

                    const float4 m0 = sv.materials[2 * material], m1 = sv.materials[2 * material + 1];
                    if ( bounce == 0 && Dot( sf.wo, sf.normal ) > 0 ) L = L + T * F3( m1 ); // path_tracer.cpp:125-128
                    if ( bounce == 0 ) pb.rad[p] = F4( L, 0.0f );

                    f3 Kd = F3( m0 ); // Material::GetAlbedo (src/resource/material.cpp:34-43)
                    const int tex = __float_as_int( m0.w );
                    if ( tex >= 0 ) Kd = Kd * TextureFetch( sv, tex, tu, tv );
                    sf.F   = Kd / kPi; // BRDF::F (src/resource/material.cpp:15-18)
                    uint32_t local, slot, row, col;
                    PathToSampleSlot( PathOrder{ width, height, numPaths / numPixels }, p, &local, &slot );
                    SlotToPixel( PixelOrder{ width, height }, slot, &row, &col );
                    pixel  = row * width + col;
                    sample = sampleBegin + local;
                }
            }
        }
        waveShaded += LanesSet( lit );

        // LDirect / EstimateSingleDirect (path_tracer.cpp:74-104): one shadow record per light sample, queued for
        // ConnectKernel (visibility) and summed by ResolveKernel.  The loop bounds are the same for every lane.
        uint32_t slot = 0;
        for ( uint32_t li = 0; li < sv.numLights; ++li )
        {
            const DevLight& l = sv.lights[li];
            for ( int smp = 0; smp < l.nSamples; ++smp, ++slot )
            {
                bool skip = true;
                const uint32_t rec = slot * ( pb.units * kUnitPaths ) + q;
                if ( lit )
                {
                    skip = false;
                    f3 wi, Li = F3( l.Lemit );
                    float pdf = 1.0f, tMax;
                    if ( l.type == PT_LIGHT_POINT ) // src/lights.cpp:8-23
                    {
                        f3 toLight = F3( l.a ) - sf.position;
                        wi   = Normalize( toLight );
                        tMax = Length( toLight );
                        Li   = Li / ( tMax * tMax );
                    }
                    else if ( l.type == PT_LIGHT_DIRECTIONAL ) // src/lights.cpp:25-38
                    {
                        wi   = -F3( l.a );
                        tMax = FLT_MAX;
                    }
                    else // src/lights.cpp:40-55 over src/shapes.cpp:11-30
                    {
                        // f( Rand(), Rand() ): g++ evaluates the arguments right to left, so the first
                        // draw is u2 and the second u1
                        float u2, u1;
                        DrawPair( rng, pixel, sample, drawBase + 2u * ( l.drawOffset + smp ), &u2, &u1 );
                        f3 sp, sn;
                        if ( l.shapeType == PT_PRIM_TRIANGLE ) // src/shapes.cpp:104-115, src/sampling.cpp:63-67
                        {
                            float su0 = sqrtf( u1 );
                            float bu = 1 - su0, bv = u2 * su0, bw = 1 - bu - bv;
                            sp = F3( l.v0 ) * bu + F3( l.v1 ) * bv + F3( l.v2 ) * bw;
                            sn = Normalize( F3( l.n0 ) * bu + F3( l.n1 ) * bv + F3( l.n2 ) * bw );
                        }
                        else // src/shapes.cpp:42-50
                        {
                            sn = UniformSampleSphere( u1, u2 );
                            sp = F3( l.a ) + sn * l.radius;
                        }
                        pdf = l.invArea;
                        f3 toLight = sp - sf.position;
                        if ( Length( toLight ) == 0 )
                            pdf = 0;
                        else
                        {
                            float r2 = Dot( toLight, toLight );
                            f3 w     = Normalize( toLight );
                            pdf *= r2 / AbsDot( sn, -w );
                            if ( isinf( pdf ) ) pdf = 0.0f;
                        }
                        wi   = Normalize( sp - sf.position );
                        tMax = Length( sp - sf.position );
                        if ( static_cast< double >( tMax ) < 0.002 ) skip = true;
                        if ( !( Dot( -wi, sn ) > 0 ) ) skip = true; // one-sided emitter
                    }
                    f3 est = F3( 0.0f, 0.0f, 0.0f );
                    if ( pdf == 0 )
                        skip = true;
                    else
                        est = ( ( sf.F * Li ) * AbsDot( sf.normal, wi ) ) / pdf; // path_tracer.cpp:87
                    if ( IsZero( est ) ) skip = true;
                    if ( !pb.constantShadowDir ) pb.shDir[rec] = F4( wi, skip ? -1.0f : tMax ); // else: nobody reads it
                    if ( !skip )
                    {
                        if ( pb.directSum )
                        {
                            // what ResolveKernel computes for one light with one sample, in its order
                            const f3 zero = F3( 0.0f, 0.0f, 0.0f ), Ld = zero + est;
                            const f3 Ldirect = zero + Ld / 1.0f;
                            pb.shEst[rec] = F4( T * Ldirect, __uint_as_float( p ) );
                        }
                        else
                            pb.shEst[rec] = F4( est, 0.0f );
                        anyShadow = true;
                    }
                }
                SegmentAppend( !skip, rec, outShadowRays, nShadowRays );
            }
        }

        // BRDF::Sample_F / Pdf (src/resource/material.cpp:20-32) and the throughput update (path_tracer.cpp:139-152)
        if ( lit )
        {
            float u2, u1;
            DrawPair( rng, pixel, sample, drawBase + static_cast< uint32_t >( sv.drawsPerBounce ) - 2u, &u2, &u1 );
            f3 local = CosineSampleHemisphere( u1, u2 );
            f3 wi    = sf.tangent * local.x + sf.bitangent * local.y + sf.normal * local.z;
            bool sameHemisphere = Dot( sf.wo, sf.normal ) * Dot( wi, sf.normal ) > 0;
            float pdf = sameHemisphere ? AbsDot( wi, sf.normal ) / kPi : 0.0f;
            if ( !( pdf == 0.0f || IsZero( sf.F ) ) )
            {
                f3 Tn = T * ( ( sf.F * AbsDot( wi, sf.normal ) ) / pdf );
                if ( !IsZero( Tn ) && bounce + 1 < maxDepth )
                {
                    alive = true;
                    nextD = wi;
                    nextT = Tn;
                }
            }
            if ( anyShadow ) pb.shadowO[q] = P3( sf.position );
        }
        {
            // the path's state for the next bounce goes where its queue entry goes
            const unsigned long long mAlive = Ballot( alive );
            if ( alive )
            {
                const uint32_t next = segment + nNext + LanesBelow( mAlive );
                pb.queue[nxt][next] = p;
                pb.rayO[nxt][next] = P3( sf.position );
                pb.rayD[nxt][next] = P3( nextD );
                pb.thr[nxt][next] = P3( nextT );
            }
            nNext += static_cast< uint32_t >( __popcll( mAlive ) );
        }
        if ( !pb.directSum ) SegmentAppend( anyShadow, q, outShadowPaths, nShadowPaths ); // ResolveKernel's work list
        if constexpr ( !kSphereHits ) SegmentAppend( deferred, q, pb.sphereQueue + segment, nSpheres );
    }
    if ( LaneIndex() == 0 )
    {
        *UnitCount( pb, bounce, kCountNextPaths, unit )   = nNext;
        *UnitCount( pb, bounce, kCountShadowPaths, unit ) = nShadowPaths;
        *UnitCount( pb, bounce, kCountShadowRays, unit )  = nShadowRays;
        if constexpr ( !kSphereHits ) *UnitCount( pb, bounce, kCountSphereHits, unit ) = nSpheres;
    }
    FlushWaveCounter( stats, ST_COUNT + static_cast< int >( ( unit % kShadedSlots ) * kShadedStride ), waveShaded ); // see kShadedSlots
}

// L += throughput * sum over lights of ( sum over the unoccluded samples ) / nSamples (path_tracer.cpp:90-104,
// 133-134), in light and sample order, for every path that queued a shadow ray.
__global__ __launch_bounds__( kBlock ) void ResolveKernel( SceneView sv, PathBuffers pb, int bounce )
{
    const uint32_t unit = WaveIndex(); // one unit per wave
    if ( unit >= pb.units ) return;
    const uint32_t* items = pb.shadowPaths + unit * kUnitPaths;
    const uint32_t limit  = *UnitCount( pb, bounce, kCountShadowPaths, unit );
    const int cur = bounce & 1;
    const uint32_t entries = pb.units * kUnitPaths;
    for ( uint32_t j = LaneIndex(); j < limit; j += 64u )
    {
        const uint32_t q = items[j], p = bounce == 0 ? q : pb.queue[cur][q];
        f3 Ldirect = F3( 0.0f, 0.0f, 0.0f );
        uint32_t slot = 0;
        for ( uint32_t li = 0; li < sv.numLights; ++li )
        {
            const int nSamples = sv.lights[li].nSamples;
            f3 Ld = F3( 0.0f, 0.0f, 0.0f );
            for ( int smp = 0; smp < nSamples; ++smp, ++slot )
            {
                const uint32_t rec = slot * entries + q;
                const float tMax   = reinterpret_cast< const float* >( pb.shDir )[4ull * rec + 3];
                if ( tMax < 0.0f ) continue; // -1: no ray was cast, -2: occluded
                Ld = Ld + F3( pb.shEst[rec] );
            }
            Ldirect = Ldirect + Ld / static_cast< float >( nSamples );
        }
        const f3 T = bounce == 0 ? F3( 1.0f, 1.0f, 1.0f ) : F3( pb.thr[cur][q] ); // see RaygenKernel
        f3 L       = F3( pb.rad[p] );
        pb.rad[p]  = F4( L + T * Ldirect, 0.0f );
    }
}

// totalColor += Li( sample ) in sample order (path_tracer.cpp:185-191); film.w counts the samples
__global__ __launch_bounds__( kBlock ) void FilmKernel( PathBuffers pb, float4* film, uint32_t width, uint32_t height, uint32_t samplesInPass )
{
    const uint32_t numPixels = width * height;
    for ( uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < numPixels; slot += gridDim.x * kBlock )
    {
        uint32_t row, col;
        SlotToPixel( PixelOrder{ width, height }, slot, &row, &col );
        const uint32_t pixel = row * width + col;
        float4 f = film[pixel];
        for ( uint32_t s = 0; s < samplesInPass; ++s )
        {
            const float4 L = pb.rad[SampleSlotToPath( PathOrder{ width, height, samplesInPass }, s, slot )];
            f.x += L.x;
            f.y += L.y;
            f.z += L.z;
        }
        f.w += static_cast< float >( samplesInPass );
        film[pixel] = f;
    }
}

// film += other, element by element: the films of several GPUs (or of several scenes on one GPU) summed onto one.
// `other` may be memory of another GPU mapped for peer access, read over xGMI.
__global__ __launch_bounds__( kBlock ) void AddFilmKernel( float4* film, const float4* other, uint32_t numPixels )
{
    for ( uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < numPixels; i += gridDim.x * kBlock )
    {
        const float4 a = film[i], b = other[i];
        film[i] = make_float4( a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w );
    }
}

// Uncharted2TonemapHelper (src/tonemap.cpp:36-45)
__device__ __forceinline__ float U2Helper( float x )
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ( ( x * ( A * x + C * B ) + D * E ) / ( x * ( A * x + B ) + D * F ) ) - E / F;
}

// mean = total / spp (path_tracer.cpp:193), then tonemap, gamma, clamp (path_tracer.cpp:211-219,
// src/tonemap.cpp:6-9, 29-33, 47-53).  The power is the host C library's powf restated (pt_powf.h), so the
// picture equals the reference's pass bit for bit.
__global__ __launch_bounds__( kBlock ) void TonemapKernel( const float4* film, float* linearOut, float* tonemappedOut, uint32_t numPixels,
                                                          float exposure, float gamma, uint32_t op )
{
    const float whiteScale = 1.0f / U2Helper( 11.2f );
    const float invGamma   = 1.0f / gamma;
    for ( uint32_t pixel = blockIdx.x * kBlock + threadIdx.x; pixel < numPixels; pixel += gridDim.x * kBlock )
    {
        const float4 f = film[pixel];
        float c[3] = { f.x / f.w, f.y / f.w, f.z / f.w };
        if ( linearOut )
        {
            linearOut[3 * pixel + 0] = c[0];
            linearOut[3 * pixel + 1] = c[1];
            linearOut[3 * pixel + 2] = c[2];
        }
        if ( tonemappedOut )
        {
            for ( int k = 0; k < 3; ++k )
            {
                float x = c[k];
                if ( op == PT_TONEMAP_UNCHARTED2 )
                    x = U2Helper( exposure * x ) * whiteScale;
                else if ( op == PT_TONEMAP_REINHARD )
                {
                    float a = exposure * x;
                    x       = a / ( 1.0f + a );
                }
                if ( op != PT_TONEMAP_NONE )
                {
                    x = PowLibm( x, invGamma );
                    x = x < 0.0f ? 0.0f : x;
                    x = 1.0f < x ? 1.0f : x;
                }
                tonemappedOut[3 * pixel + k] = x;
            }
        }
    }
}

// Batch queries behind pt_trace_closest / pt_trace_any: device arrays of n rays, one thread per ray.
// kMode: PT_TRAVERSAL_*
template < int kMode >
__global__ __launch_bounds__( kBlock ) void TraceClosestKernel( SceneView sv, uint64_t n, const float* origins, const float* dirs, int32_t* prim,
                                                               float* t, float* u, float* v, uint32_t* nodes, uint32_t* tests )
{
    for ( uint64_t i = static_cast< uint64_t >( blockIdx.x ) * kBlock + threadIdx.x; i < n; i += static_cast< uint64_t >( gridDim.x ) * kBlock )
    {
        Counters cnt;
        HitRecord h;
        if constexpr ( kMode != PT_TRAVERSAL_REFERENCE )
        {
            uint32_t ambiguous;
            if constexpr ( kMode == PT_TRAVERSAL_WIDE )
                h = TraceClosestWide( sv, MakeWideBuffers( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), cnt, &ambiguous );
            else
                h = TraceClosestOrdered( sv, MakeSceneBuffers< true >( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), cnt, &ambiguous );
            if ( ambiguous ) h = RetraceInReferenceOrder( sv, MakeSceneBuffers< false >( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), ( ambiguous & kUntrusted ) ? FLT_MAX : h.t, cnt );
        }
        else
            h = TraceClosest( sv, MakeSceneBuffers< false >( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), cnt );
        if ( prim ) prim[i] = h.prim;
        if ( t ) t[i] = h.t;
        if ( u ) u[i] = h.u;
        if ( v ) v[i] = h.v;
        if ( nodes ) nodes[i] = cnt.nodes;
        if ( tests ) tests[i] = cnt.tris + cnt.spheres;
    }
}

template < int kMode >
__global__ __launch_bounds__( kBlock ) void TraceAnyKernel( SceneView sv, uint64_t n, const float* origins, const float* dirs, const float* tMax,
                                                           uint8_t* occluded, uint32_t* nodes, uint32_t* tests )
{
    for ( uint64_t i = static_cast< uint64_t >( blockIdx.x ) * kBlock + threadIdx.x; i < n; i += static_cast< uint64_t >( gridDim.x ) * kBlock )
    {
        Counters cnt;
        bool occ;
        if constexpr ( kMode == PT_TRAVERSAL_WIDE )
            occ = TraceAnyWide( sv, MakeWideBuffers( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), tMax[i], cnt );
        else if constexpr ( kMode == PT_TRAVERSAL_ORDERED )
            occ = TraceAnyOrdered( sv, MakeSceneBuffers< true >( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), tMax[i], cnt );
        else
            occ = TraceAny( sv, MakeSceneBuffers< false >( sv ), F3( origins + 3 * i ), F3( dirs + 3 * i ), tMax[i], cnt );
        occluded[i] = occ ? 1 : 0;
        if ( nodes ) nodes[i] = cnt.nodes;
        if ( tests ) tests[i] = cnt.tris + cnt.spheres;
    }
}

// ------------------------------------------------------------------------------------------------ launch wrappers

static int GridFor( const LaunchCfg& cfg, uint64_t work )
{
    uint64_t blocks = ( work + kBlock - 1 ) / kBlock;
    if ( blocks < 1 ) blocks = 1;
    if ( blocks > static_cast< uint64_t >( cfg.blocks ) ) blocks = cfg.blocks;
    return static_cast< int >( blocks );
}

// One unit per wave, four waves per block
static dim3 UnitGrid( const PathBuffers& pb ) { return dim3( ( pb.units + kBlock / 64 - 1 ) / ( kBlock / 64 ) ); }

void TraversalOccupancy( bool spheres, int* extendBlocksPerCu, int* connectBlocksPerCu )
{
    int e = 0, c = 0;
    const hipError_t rc = spheres ? hipOccupancyMaxActiveBlocksPerMultiprocessor( &e, ExtendKernel< true, true >, kTraverseBlock, 0 )
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor( &e, ExtendKernel< false, true >, kTriangleBlock, 0 );
    if ( rc != hipSuccess || e < 1 ) e = 2;
    if ( hipOccupancyMaxActiveBlocksPerMultiprocessor( &c, ConnectKernel< false >, kConnectBlock, 0 ) != hipSuccess || c < 1 ) c = 4;
    *extendBlocksPerCu  = e;
    *connectBlocksPerCu = c;
}

void LaunchRaygen( const LaunchCfg& cfg, const CameraFrame& cam, const RngParams& rng, const PathBuffers& pb, uint32_t width,
                   uint32_t height, uint32_t sampleBegin, uint32_t numPaths )
{
    hipLaunchKernelGGL( RaygenKernel, dim3( GridFor( cfg, numPaths ) ), dim3( kBlock ), 0, cfg.stream, cam, rng, pb, width, height, sampleBegin, numPaths );
}

// The any-hit kernel over the wide layout: resident blocks per compute unit (its blocks have kWideBlock threads)
void WideShadowOccupancy( int* blocksPerCu, int* threadsPerBlock )
{
    int b = 0;
    if ( hipOccupancyMaxActiveBlocksPerMultiprocessor( &b, ConnectWideKernel< true >, kWideBlock, 0 ) != hipSuccess || b < 1 ) b = 1;
    *blocksPerCu     = b;
    *threadsPerBlock = kWideBlock;
}

void WideOccupancy( bool spheres, int* blocksPerCu, int* threadsPerBlock, int* ldsStackEntries )
{
    int b = 0;
    const hipError_t rc = spheres ? hipOccupancyMaxActiveBlocksPerMultiprocessor( &b, ExtendWideKernel< true >, kWideSphereBlock, 0 )
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor( &b, ExtendWideKernel< false >, kWideBlock, 0 );
    if ( rc != hipSuccess || b < 1 ) b = 1;
    *blocksPerCu     = b;
    *threadsPerBlock = spheres ? kWideSphereBlock : kWideBlock;
    *ldsStackEntries = static_cast< int >( kWideLdsStack );
}

void LaunchExtend( const LaunchCfg& cfg, const SceneView& sv, const PathBuffers& pb, int bounce, uint32_t numPaths, uint64_t* stats )
{
    if ( cfg.wide )
    {
        if ( sv.numSpheres )
            hipLaunchKernelGGL( ( ExtendWideKernel< true > ), dim3( cfg.wideBlocks ), dim3( kWideSphereBlock ), 0, cfg.stream, sv, pb, bounce, numPaths, cfg.wideSpill, stats );
        else
            hipLaunchKernelGGL( ( ExtendWideKernel< false > ), dim3( cfg.wideBlocks ), dim3( kWideBlock ), 0, cfg.stream, sv, pb, bounce, numPaths, cfg.wideSpill, stats );
        hipLaunchKernelGGL( RedoKernel, dim3( PT_REDO_BLOCKS ), dim3( kBlock ), 0, cfg.stream, sv, pb, bounce, stats );
        return;
    }
    const dim3 grid( cfg.extendBlocks ), block( sv.numSpheres ? kTraverseBlock : kTriangleBlock );
    if ( sv.numSpheres && cfg.ordered )
        hipLaunchKernelGGL( ( ExtendKernel< true, true > ), grid, block, 0, cfg.stream, sv, pb, bounce, numPaths, stats );
    else if ( sv.numSpheres )
        hipLaunchKernelGGL( ( ExtendKernel< true, false > ), grid, block, 0, cfg.stream, sv, pb, bounce, numPaths, stats );
    else if ( cfg.ordered )
        hipLaunchKernelGGL( ( ExtendKernel< false, true > ), grid, block, 0, cfg.stream, sv, pb, bounce, numPaths, stats );
    else
        hipLaunchKernelGGL( ( ExtendKernel< false, false > ), grid, block, 0, cfg.stream, sv, pb, bounce, numPaths, stats );
    // one thread per ambiguous ray where the grid reaches (its length is known on the device only); a thread's rays run one after the other
    if ( cfg.ordered ) hipLaunchKernelGGL( RedoKernel, dim3( PT_REDO_BLOCKS ), dim3( kBlock ), 0, cfg.stream, sv, pb, bounce, stats );
}

void LaunchShade( const LaunchCfg& cfg, const SceneView& sv, const RngParams& rng, const PathBuffers& pb, int bounce, int maxDepth,
                  uint32_t width, uint32_t height, uint32_t sampleBegin, uint32_t numPaths, uint64_t* stats )
{
    hipLaunchKernelGGL( ShadeKernel< false >, UnitGrid( pb ), dim3( kBlock ), 0, cfg.stream, sv, rng, pb, bounce, maxDepth, width, height,
                        sampleBegin, numPaths, stats );
    // paths whose closest hit is a sphere were deferred to their own queue
    if ( sv.numSpheres != 0 )
        hipLaunchKernelGGL( ShadeKernel< true >, UnitGrid( pb ), dim3( kBlock ), 0, cfg.stream, sv, rng, pb, bounce, maxDepth, width, height,
                            sampleBegin, numPaths, stats );
}

// Over the binary tree shadow rays always run in the reference's order: an any-hit query gives the same answer in either
// order, and the front-to-back order was measured slower for it (Sponza-class scene: 41.6 instead of 43.1 nodes per shadow
// ray, but every step reads a third word and the LDS copy holds half as many records: +11 % per launch), so ConnectKernel
// has no front-to-back instance.
void LaunchConnect( const LaunchCfg& cfg, const SceneView& sv, const PathBuffers& pb, int bounce, uint64_t* stats )
{
    if ( cfg.wideShadows )
    {
        if ( pb.directSum && pb.constantShadowDir && sv.lightWideNodes )
            hipLaunchKernelGGL( ConnectLightWideKernel, dim3( cfg.wideShadowBlocks ), dim3( kWideBlock ), 0, cfg.stream, sv, pb, bounce, cfg.wideSpill, stats );
        else if ( pb.directSum && pb.constantShadowDir )
            hipLaunchKernelGGL( ( ConnectWideKernel< true, true > ), dim3( cfg.wideShadowBlocks ), dim3( kWideBlock ), 0, cfg.stream, sv, pb, bounce, cfg.wideSpill, stats );
        else if ( pb.directSum )
            hipLaunchKernelGGL( ( ConnectWideKernel< true > ), dim3( cfg.wideShadowBlocks ), dim3( kWideBlock ), 0, cfg.stream, sv, pb, bounce, cfg.wideSpill, stats );
        else
            hipLaunchKernelGGL( ( ConnectWideKernel< false > ), dim3( cfg.wideShadowBlocks ), dim3( kWideBlock ), 0, cfg.stream, sv, pb, bounce, cfg.wideSpill, stats );
        return;
    }
    const dim3 grid( cfg.connectBlocks ), block( kConnectBlock );
    if ( cfg.ordered && sv.shadowNodes && pb.directSum )
        hipLaunchKernelGGL( ( ConnectParallelKernel< true > ), grid, block, 0, cfg.stream, sv, pb, bounce, stats );
    else if ( cfg.ordered && sv.shadowNodes )
        hipLaunchKernelGGL( ( ConnectParallelKernel< false > ), grid, block, 0, cfg.stream, sv, pb, bounce, stats );
    else if ( pb.directSum )
        hipLaunchKernelGGL( ( ConnectKernel< true > ), grid, block, 0, cfg.stream, sv, pb, bounce, stats );
    else
        hipLaunchKernelGGL( ( ConnectKernel< false > ), grid, block, 0, cfg.stream, sv, pb, bounce, stats );
}

void LaunchResolve( const LaunchCfg& cfg, const SceneView& sv, const PathBuffers& pb, int bounce )
{
    hipLaunchKernelGGL( ResolveKernel, UnitGrid( pb ), dim3( kBlock ), 0, cfg.stream, sv, pb, bounce );
}

void LaunchFilm( const LaunchCfg& cfg, const PathBuffers& pb, float4* film, uint32_t width, uint32_t height, uint32_t samplesInPass )
{
    hipLaunchKernelGGL( FilmKernel, dim3( GridFor( cfg, static_cast< uint64_t >( width ) * height ) ), dim3( kBlock ), 0, cfg.stream, pb, film, width,
                        height, samplesInPass );
}

void LaunchAddFilm( const LaunchCfg& cfg, float4* film, const float4* other, uint32_t numPixels )
{
    hipLaunchKernelGGL( AddFilmKernel, dim3( GridFor( cfg, numPixels ) ), dim3( kBlock ), 0, cfg.stream, film, other, numPixels );
}

void LaunchTonemap( const LaunchCfg& cfg, const float4* film, float* linearOut, float* tonemappedOut, uint32_t numPixels, float exposure,
                    float gamma, uint32_t op )
{
    hipLaunchKernelGGL( TonemapKernel, dim3( GridFor( cfg, numPixels ) ), dim3( kBlock ), 0, cfg.stream, film, linearOut, tonemappedOut, numPixels,
                        exposure, gamma, op );
}

void LaunchTraceClosest( const LaunchCfg& cfg, const SceneView& sv, uint64_t n, const float* origins, const float* dirs, int32_t* prim,
                         float* t, float* u, float* v, uint32_t* nodes, uint32_t* tests )
{
    if ( cfg.wide )
        hipLaunchKernelGGL( TraceClosestKernel< PT_TRAVERSAL_WIDE >, dim3( GridFor( cfg, n ) ), dim3( kBlock ), 0, cfg.stream, sv, n, origins, dirs, prim, t, u, v, nodes, tests );
    else if ( cfg.ordered )
        hipLaunchKernelGGL( TraceClosestKernel< PT_TRAVERSAL_ORDERED >, dim3( GridFor( cfg, n ) ), dim3( kBlock ), 0, cfg.stream, sv, n, origins, dirs, prim, t, u, v, nodes, tests );
    else
        hipLaunchKernelGGL( TraceClosestKernel< PT_TRAVERSAL_REFERENCE >, dim3( GridFor( cfg, n ) ), dim3( kBlock ), 0, cfg.stream, sv, n, origins, dirs, prim, t, u, v, nodes, tests );
}

void LaunchTraceAny( const LaunchCfg& cfg, const SceneView& sv, uint64_t n, const float* origins, const float* dirs, const float* tMax,
                     uint8_t* occluded, uint32_t* nodes, uint32_t* tests )
{
    if ( cfg.wide )
        hipLaunchKernelGGL( TraceAnyKernel< PT_TRAVERSAL_WIDE >, dim3( GridFor( cfg, n ) ), dim3( kBlock ), 0, cfg.stream, sv, n, origins, dirs, tMax, occluded, nodes, tests );
    else if ( cfg.ordered )
        hipLaunchKernelGGL( TraceAnyKernel< PT_TRAVERSAL_ORDERED >, dim3( GridFor( cfg, n ) ), dim3( kBlock ), 0, cfg.stream, sv, n, origins, dirs, tMax, occluded, nodes, tests );
    else
        hipLaunchKernelGGL( TraceAnyKernel< PT_TRAVERSAL_REFERENCE >, dim3( GridFor( cfg, n ) ), dim3( kBlock ), 0, cfg.stream, sv, n, origins, dirs, tMax, occluded, nodes, tests );
}

} // namespace ptd
