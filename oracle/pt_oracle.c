/*
 * pt_oracle.c - CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 * Written from the reference's algorithm, not from its code layout: one data-oriented pass over the flat
 * arrays of the C-ABI, no virtual shapes, no pointers between objects.  Every routine cites the reference
 * lines it restates; the floating point operation order inside each routine is the reference's, because
 * the tests demand bit-identical results.
 */
#include "pt_oracle.h"
#include "pt_rng.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PTO_PI 3.14159265358979323846f /* src/math.hpp:9-11: M_PI as a float */
#define PTO_EPSILON 0.00001f           /* src/path_tracer.cpp:16 */

/* ------------------------------------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

static inline v3 V( float x, float y, float z ) { v3 r = { x, y, z }; return r; }
static inline v3 ld3( const float* p ) { return V( p[0], p[1], p[2] ); }
static inline void st3( float* p, v3 a ) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add( v3 a, v3 b ) { return V( a.x + b.x, a.y + b.y, a.z + b.z ); }
static inline v3 sub( v3 a, v3 b ) { return V( a.x - b.x, a.y - b.y, a.z - b.z ); }
static inline v3 mul( v3 a, v3 b ) { return V( a.x * b.x, a.y * b.y, a.z * b.z ); }
static inline v3 scale( v3 a, float s ) { return V( a.x * s, a.y * s, a.z * s ); }
static inline v3 divs( v3 a, float s ) { return V( a.x / s, a.y / s, a.z / s ); }
static inline v3 neg( v3 a ) { return V( -a.x, -a.y, -a.z ); }
static inline int is_zero( v3 a ) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
/* glm/detail/func_geometric.inl:48-55, 68-79, 8-14, 82-90 */
static inline float dot( v3 a, v3 b ) { v3 t = mul( a, b ); return t.x + t.y + t.z; }
static inline v3 cross( v3 x, v3 y ) { return V( x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y ); }
static inline float length( v3 a ) { return sqrtf( dot( a, a ) ); }
static inline v3 normalize( v3 a ) { return scale( a, 1.0f / sqrtf( dot( a, a ) ) ); }
static inline float absdot( v3 a, v3 b ) { return fabsf( dot( a, b ) ); } /* src/math.hpp:48-51 */
static inline float max0( float x ) { return 0.0f < x ? x : 0.0f; }      /* std::max( 0.f, x ) */

/* ------------------------------------------------------------------------------------------------ scene */
struct PtoScene
{
    PtSceneDesc d; /* pointers below are owned copies */
    void* blocks[16];
    int num_blocks;
    int draws_per_bounce;
    uint32_t traversal; /* PT_TRAVERSAL_* */
    uint8_t* sphere_below; /* per node: its subtree holds a sphere (front-to-back order only), or NULL: no spheres */
};

static _Thread_local char g_error[256];
const char* pto_last_error( void ) { return g_error; }
static int fail( const char* msg )
{
    snprintf( g_error, sizeof( g_error ), "%s", msg );
    return 1;
}

static const void* keep( PtoScene* s, const void* src, size_t bytes )
{
    if ( !src || bytes == 0 ) return NULL;
    void* p = malloc( bytes );
    memcpy( p, src, bytes );
    s->blocks[s->num_blocks++] = p;
    return p;
}

int pto_scene_create( const PtSceneDesc* desc, PtoScene** out )
{
    if ( !desc || !out ) return fail( "pto_scene_create: null argument" );
    if ( desc->struct_size != sizeof( PtSceneDesc ) || desc->version != PT_API_VERSION ) return fail( "pto_scene_create: descriptor size / version mismatch" );
    PtoScene* s = (PtoScene*)calloc( 1, sizeof( PtoScene ) );
    s->d           = *desc;
    s->d.nodes     = (const PtBvhNode*)keep( s, desc->nodes, desc->num_nodes * sizeof( PtBvhNode ) );
    s->d.prims     = (const PtPrim*)keep( s, desc->prims, desc->num_prims * sizeof( PtPrim ) );
    s->d.triangles = (const PtTriangle*)keep( s, desc->triangles, desc->num_triangles * sizeof( PtTriangle ) );
    s->d.spheres   = (const PtSphere*)keep( s, desc->spheres, desc->num_spheres * sizeof( PtSphere ) );
    s->d.meshes    = (const PtMesh*)keep( s, desc->meshes, desc->num_meshes * sizeof( PtMesh ) );
    s->d.vertices  = (const float*)keep( s, desc->vertices, desc->num_vertices * 12 );
    s->d.normals   = (const float*)keep( s, desc->normals, desc->num_vertices * 12 );
    s->d.tangents  = (const float*)keep( s, desc->tangents, desc->num_vertices * 12 );
    s->d.uvs       = (const float*)keep( s, desc->uvs, desc->num_vertices * 8 );
    s->d.materials = (const PtMaterial*)keep( s, desc->materials, desc->num_materials * sizeof( PtMaterial ) );
    s->d.textures  = (const PtTexture*)keep( s, desc->textures, desc->num_textures * sizeof( PtTexture ) );
    s->d.texels    = (const uint8_t*)keep( s, desc->texels, desc->num_texels * 4 );
    s->d.lights    = (const PtLight*)keep( s, desc->lights, desc->num_lights * sizeof( PtLight ) );
    s->d.skybox    = (const float*)keep( s, desc->skybox, desc->skybox ? (size_t)6 * desc->skybox_width * desc->skybox_height * 16 : 0 );
    int area = 0;
    for ( uint64_t i = 0; i < desc->num_lights; ++i )
        if ( desc->lights[i].type == PT_LIGHT_AREA ) area += desc->lights[i].n_samples;
    s->draws_per_bounce = 2 * area + 2;
    s->sphere_below = NULL;
    if ( desc->num_spheres > 0 )
    {
        /* children lie behind their parent in the pre-order array: one backward pass */
        uint8_t* below = (uint8_t*)calloc( desc->num_nodes, 1 );
        if ( !below ) { pto_scene_destroy( s ); return fail( "pto_scene_create: out of memory" ); }
        s->blocks[s->num_blocks++] = below;
        for ( uint64_t k = desc->num_nodes; k-- > 0; )
        {
            const PtBvhNode* n = &s->d.nodes[k];
            if ( n->num_shapes > 0 )
                for ( int p = n->offset; p < n->offset + n->num_shapes; ++p ) below[k] |= s->d.prims[p].type == PT_PRIM_SPHERE;
            else
                below[k] = below[k + 1] | below[n->offset];
        }
        s->sphere_below = below;
    }
    *out = s;
    return 0;
}

int pto_scene_set_traversal( PtoScene* s, uint32_t traversal )
{
    if ( !s || traversal > PT_TRAVERSAL_ORDERED ) return fail( "pto_scene_set_traversal: bad argument" );
    s->traversal = traversal;
    return 0;
}

void pto_scene_destroy( PtoScene* s )
{
    if ( !s ) return;
    for ( int i = 0; i < s->num_blocks; ++i ) free( s->blocks[i] );
    free( s );
}

/* ------------------------------------------------------------------------------------------------ primitive tests */

/* intersect::RayAABBFastest, src/intersection_tests.cpp:91-110.  Comparisons, not min/max: a NaN slab value
 * (0 * inf) makes every comparison false, so the test passes. */
static int ray_aabb_entry( v3 o, v3 inv, const int neg[3], const float* bmin, const float* bmax, float maxT, int inclusive, float* entry )
{
    float tMin  = ( ( neg[0] ? bmax : bmin )[0] - o.x ) * inv.x;
    float tMax  = ( ( 1 - neg[0] ? bmax : bmin )[0] - o.x ) * inv.x;
    float tyMin = ( ( neg[1] ? bmax : bmin )[1] - o.y ) * inv.y;
    float tyMax = ( ( 1 - neg[1] ? bmax : bmin )[1] - o.y ) * inv.y;
    if ( tMin > tyMax || tyMin > tMax ) return 0;
    if ( tyMin > tMin ) tMin = tyMin;
    if ( tyMax < tMax ) tMax = tyMax;
    float tzMin = ( ( neg[2] ? bmax : bmin )[2] - o.z ) * inv.z;
    float tzMax = ( ( 1 - neg[2] ? bmax : bmin )[2] - o.z ) * inv.z;
    if ( tMin > tzMax || tzMin > tMax ) return 0;
    if ( tzMin > tMin ) tMin = tzMin;
    if ( tzMax < tMax ) tMax = tzMax;
    /* inclusive: the front-to-back order's test (tMin <= maxT), not the reference's */
    if ( entry ) *entry = tMin;
    return ( inclusive ? tMin <= maxT : tMin < maxT ) && ( tMax > 0 );
}

static int ray_aabb_cmp( v3 o, v3 inv, const int neg[3], const float* bmin, const float* bmax, float maxT, int inclusive )
{
    return ray_aabb_entry( o, inv, neg, bmin, bmax, maxT, inclusive, NULL );
}

static int ray_aabb( v3 o, v3 inv, const int neg[3], const float* bmin, const float* bmax, float maxT )
{
    return ray_aabb_cmp( o, inv, neg, bmin, bmax, maxT, 0 );
}

/* intersect::RayTriangle (Moller-Trumbore), src/intersection_tests.cpp:34-66 */
static int ray_triangle( v3 o, v3 d, v3 v0, v3 v1, v3 v2, float* t, float* u, float* v, float maxT )
{
    v3 v0v1   = sub( v1, v0 );
    v3 v0v2   = sub( v2, v0 );
    v3 pvec   = cross( d, v0v2 );
    float det = dot( v0v1, pvec );
    if ( fabs( (double)det ) < 0.000000001 ) return 0; /* double literal, double comparison */
    float invDet = 1 / det;
    v3 tvec      = sub( o, v0 );
    *u           = dot( tvec, pvec ) * invDet;
    if ( *u < 0 || *u > 1 ) return 0;
    v3 qvec = cross( tvec, v0v1 );
    *v      = dot( d, qvec ) * invDet;
    if ( *v < 0 || *u + *v > 1 ) return 0;
    *t = dot( v0v2, qvec ) * invDet;
    return *t < maxT && *t > 0;
}

/* intersect::RaySphere, src/intersection_tests.cpp:10-31 */
static int ray_sphere( v3 o, v3 d, v3 centre, float radius, float* t, float maxT )
{
    v3 OC      = sub( o, centre );
    float a    = dot( d, d );
    float b    = 2 * dot( OC, d );
    float c    = dot( OC, OC ) - radius * radius;
    float disc = b * b - 4 * a * c;
    if ( disc < 0 ) return 0;
    float sq = sqrtf( disc );
    *t       = ( -b - sq ) / ( 2 * a );
    if ( *t < 0 ) *t = ( -b + sq ) / ( 2 * a );
    return *t < maxT && *t > 0;
}

/* glm mat4 * vec4, glm/detail/type_mat4x4.inl:561-572: (m0*v0 + m1*v1) + (m2*v2 + m3*v3); first three rows */
static v3 mat_mul( const float* m, v3 p, float w )
{
    v3 r;
    float* out = &r.x;
    for ( int k = 0; k < 3; ++k ) out[k] = ( m[k] * p.x + m[4 + k] * p.y ) + ( m[8 + k] * p.z + m[12 + k] * w );
    return r;
}

typedef struct
{
    v3 position, normal, tangent, bitangent, wo;
    v2 uv;
    int material;
    float t, bu, bv; /* bu, bv: triangle barycentrics of the accepted hit */
    int prim;
} Hit;

typedef struct
{
    uint64_t nodes, tris, spheres, redone;
} TravCount;

static void triangle_vertices( const PtSceneDesc* d, const PtTriangle* tri, size_t idx[3] )
{
    size_t base = d->meshes[tri->mesh].vertex_offset;
    idx[0] = base + tri->i0;
    idx[1] = base + tri->i1;
    idx[2] = base + tri->i2;
}

/* Triangle::Intersect (src/shapes.cpp:117-134) / Sphere::Intersect (src/shapes.cpp:52-76): test, and on an
 * accepted hit replace the hit record. */
#define PTO_BAND ( 1.0f + 1.0f / 4096.0f ) /* front-to-back order: hits this close to the closest one are "ambiguous" */
#define PTO_TRUST ( 1.0f + 1.0f / 16384.0f ) /* a hit this far in front of its own leaf box is not to be reasoned with */

/* ambiguous == NULL: the reference's test.  Otherwise (front-to-back order): hits up to PTO_BAND times the closest
 * distance so far are looked at, and *ambiguous tells whether a second hit lies within that band of the closest
 * (bit 0).  Bit 1: a hit whose distance lies in front of the point where the ray enters the primitive's own leaf box
 * (`entry`) by more than rounding of the box test explains - a primitive test that has lost its precision (a sphere
 * seen from ten thousand radii away, a triangle met at a millionth of a degree).  The band argument rests on a hit
 * lying inside its boxes, so such a ray is traced in the reference's order from scratch. */
static int shape_intersect_band( const PtSceneDesc* d, uint32_t prim, v3 o, v3 dir, Hit* hit, TravCount* cnt, int* ambiguous, float entry )
{
    const PtPrim* p = &d->prims[prim];
    const float limit = ambiguous ? hit->t * PTO_BAND : hit->t;
    if ( p->type == PT_PRIM_TRIANGLE )
    {
        const PtTriangle* tri = &d->triangles[p->index];
        size_t i[3];
        triangle_vertices( d, tri, i );
        float t, u, v;
        cnt->tris++;
        if ( !ray_triangle( o, dir, ld3( d->vertices + 3 * i[0] ), ld3( d->vertices + 3 * i[1] ), ld3( d->vertices + 3 * i[2] ), &t, &u, &v, limit ) ) return 0;
        if ( ambiguous )
        {
            if ( entry > t * PTO_TRUST ) *ambiguous |= 2;
            if ( !( t < hit->t ) ) { *ambiguous |= 1; return 0; }
            *ambiguous = ( *ambiguous & 2 ) | ( hit->t <= t * PTO_BAND );
        }
        float w        = 1 - u - v;
        hit->t         = t;
        hit->material  = d->meshes[tri->mesh].material;
        hit->position  = add( o, scale( dir, t ) );
        hit->normal    = normalize( add( add( scale( ld3( d->normals + 3 * i[0] ), w ), scale( ld3( d->normals + 3 * i[1] ), u ) ), scale( ld3( d->normals + 3 * i[2] ), v ) ) );
        hit->tangent   = normalize( add( add( scale( ld3( d->tangents + 3 * i[0] ), w ), scale( ld3( d->tangents + 3 * i[1] ), u ) ), scale( ld3( d->tangents + 3 * i[2] ), v ) ) );
        hit->bitangent = cross( hit->normal, hit->tangent );
        hit->uv.x      = ( w * d->uvs[2 * i[0]] + u * d->uvs[2 * i[1]] ) + v * d->uvs[2 * i[2]];
        hit->uv.y      = ( w * d->uvs[2 * i[0] + 1] + u * d->uvs[2 * i[1] + 1] ) + v * d->uvs[2 * i[2] + 1];
        hit->bu        = u;
        hit->bv        = v;
        hit->prim      = (int)prim;
        return 1;
    }
    const PtSphere* s = &d->spheres[p->index];
    cnt->spheres++;
    v3 lo = mat_mul( s->world_to_local, o, 1.0f );  /* Transform::operator*( Ray ), src/transform.cpp:40-47 */
    v3 ld = mat_mul( s->world_to_local, dir, 0.0f );
    float t;
    if ( !ray_sphere( lo, ld, V( 0, 0, 0 ), 1.0f, &t, limit ) ) return 0;
    if ( ambiguous )
    {
        if ( entry > t * PTO_TRUST ) *ambiguous |= 2;
        if ( !( t < hit->t ) ) { *ambiguous |= 1; return 0; }
        *ambiguous = ( *ambiguous & 2 ) | ( hit->t <= t * PTO_BAND );
    }
    hit->t        = t;
    hit->material = s->material;
    hit->position = add( o, scale( dir, t ) );
    hit->normal   = normalize( sub( hit->position, ld3( s->position ) ) );
    v3 lp         = add( lo, scale( ld, t ) );
    /* atan2 / sin / cos resolve to the double functions in the reference's build, acosf is float */
    float theta   = (float)atan2( (double)lp.z, (double)lp.x );
    float phi     = acosf( -lp.y );
    hit->uv.x     = -0.5f * ( theta / PTO_PI + 1 );
    hit->uv.y     = phi / PTO_PI;
    hit->tangent  = V( (float)( -sin( (double)theta ) ), 0.0f, (float)cos( (double)theta ) );
    hit->bitangent = cross( hit->normal, hit->tangent );
    hit->bu = hit->bv = 0.0f;
    hit->prim = (int)prim;
    return 1;
}

static int shape_intersect( const PtSceneDesc* d, uint32_t prim, v3 o, v3 dir, Hit* hit, TravCount* cnt )
{
    return shape_intersect_band( d, prim, o, dir, hit, cnt, NULL, 0.0f );
}

/* Triangle::TestIfHit / Sphere::TestIfHit (src/shapes.cpp:136-141, 78-82): the sphere is tested in world space */
static int shape_test( const PtSceneDesc* d, uint32_t prim, v3 o, v3 dir, float maxT, TravCount* cnt )
{
    const PtPrim* p = &d->prims[prim];
    float t, u, v;
    if ( p->type == PT_PRIM_TRIANGLE )
    {
        size_t i[3];
        triangle_vertices( d, &d->triangles[p->index], i );
        cnt->tris++;
        return ray_triangle( o, dir, ld3( d->vertices + 3 * i[0] ), ld3( d->vertices + 3 * i[1] ), ld3( d->vertices + 3 * i[2] ), &t, &u, &v, maxT );
    }
    const PtSphere* s = &d->spheres[p->index];
    cnt->spheres++;
    return ray_sphere( o, dir, ld3( s->position ), s->radius, &t, maxT );
}

/* BVH::Intersect, src/bvh.cpp:230-268: first child first, 64-entry stack of second children */
static int bvh_intersect( const PtSceneDesc* d, v3 o, v3 dir, Hit* hit, TravCount* cnt )
{
    int stack[64];
    int node = 0, top = 0;
    v3 inv   = V( 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z );
    int neg[3] = { inv.x < 0, inv.y < 0, inv.z < 0 };
    float oldMaxT = hit->t;
    for ( ;; )
    {
        const PtBvhNode* n = &d->nodes[node];
        cnt->nodes++;
        if ( ray_aabb( o, inv, neg, n->aabb_min, n->aabb_max, hit->t ) )
        {
            if ( n->num_shapes > 0 )
            {
                for ( int s = n->offset; s < n->offset + n->num_shapes; ++s ) shape_intersect( d, (uint32_t)s, o, dir, hit, cnt );
                if ( top == 0 ) break;
                node = stack[--top];
            }
            else
            {
                stack[top++] = n->offset;
                node         = node + 1;
            }
        }
        else
        {
            if ( top == 0 ) break;
            node = stack[--top];
        }
    }
    return hit->t < oldMaxT;
}

/* BVH::Occluded, src/bvh.cpp:270-310 */
static int bvh_occluded( const PtSceneDesc* d, v3 o, v3 dir, float tMax, TravCount* cnt )
{
    int stack[64];
    int node = 0, top = 0;
    v3 inv   = V( 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z );
    int neg[3] = { inv.x < 0, inv.y < 0, inv.z < 0 };
    for ( ;; )
    {
        const PtBvhNode* n = &d->nodes[node];
        cnt->nodes++;
        if ( ray_aabb( o, inv, neg, n->aabb_min, n->aabb_max, tMax ) )
        {
            if ( n->num_shapes > 0 )
            {
                for ( int s = n->offset; s < n->offset + n->num_shapes; ++s )
                    if ( shape_test( d, (uint32_t)s, o, dir, tMax, cnt ) ) return 1;
                if ( top == 0 ) break;
                node = stack[--top];
            }
            else
            {
                stack[top++] = n->offset;
                node         = node + 1;
            }
        }
        else
        {
            if ( top == 0 ) break;
            node = stack[--top];
        }
    }
    return 0;
}

/* Front-to-back variants (PT_TRAVERSAL_ORDERED of include/pt_api.h) - NOT the reference's order; they exist to check
 * the GPU library's second traversal mode: the child on the side the ray comes from first (the reference stores the
 * split axis, src/bvh.hpp:19, and never reads it).  The reference's answer depends on its order only among hits
 * within rounding of one another, so the closest-hit pass keeps everything within PTO_BAND of the closest distance
 * alive (boxes entered with tMin <= PTO_BAND * t) and, when a second hit lay inside that band, runs the reference's
 * own order afterwards and returns its answer; the counters are the sums of both passes.  A plain stack of far
 * children here; the GPU side follows per-octant links instead, so equal per-ray counters check both.
 * The one exception, as on the GPU (include/pt_api.h, PT_TRAVERSAL_ORDERED): when the reference's answer is a primitive
 * whose computed t lies in front of the primitive's own box by more than the band (an ill-conditioned test) and a sound
 * hit lies between, this order returns the sound hit - it never tests the other primitive. */
static int bvh_intersect_ordered( const PtSceneDesc* d, const uint8_t* sphere_below, v3 o, v3 dir, Hit* hit, TravCount* cnt )
{
    int stack[128];
    int node = 0, top = 0, ambiguous = 0;
    v3 inv   = V( 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z );
    int neg[3] = { inv.x < 0, inv.y < 0, inv.z < 0 };
    const Hit start = *hit;
    for ( ;; )
    {
        const PtBvhNode* n = &d->nodes[node];
        cnt->nodes++;
        float entry;
        /* a box with a sphere inside is entered whatever has been found: the sphere test loses its precision with the
         * distance of the origin (it runs in the sphere's own space) long before the box test does, and a sphere
         * hit the reference computes in front of the sphere's box must be seen here to be told apart */
        const float limit = sphere_below && sphere_below[node] ? INFINITY : hit->t * PTO_BAND;
        if ( ray_aabb_entry( o, inv, neg, n->aabb_min, n->aabb_max, limit, 1, &entry ) )
        {
            if ( n->num_shapes > 0 )
            {
                for ( int s = n->offset; s < n->offset + n->num_shapes; ++s ) shape_intersect_band( d, (uint32_t)s, o, dir, hit, cnt, &ambiguous, entry );
                if ( top == 0 ) break;
                node = stack[--top];
            }
            else if ( neg[n->axis < 3 ? n->axis : 0] )
            {
                stack[top++] = node + 1;
                node         = n->offset;
            }
            else
            {
                stack[top++] = n->offset;
                node         = node + 1;
            }
        }
        else
        {
            if ( top == 0 ) break;
            node = stack[--top];
        }
    }
    if ( ambiguous & 2 )
    {
        /* a hit that cannot be trusted: the reference's order from scratch */
        *hit = start;
        cnt->redone++;
        return bvh_intersect( d, o, dir, hit, cnt );
    }
    if ( ambiguous )
    {
        /* the reference's order, bounded just above the band: farther hits cannot change a decision inside it */
        const float closest = hit->t;
        *hit   = start;
        hit->t = ( closest * PTO_BAND ) * PTO_BAND;
        cnt->redone++;
        bvh_intersect( d, o, dir, hit, cnt );
        if ( hit->prim < 0 )
        {
            *hit = start;
            bvh_intersect( d, o, dir, hit, cnt );
        }
        return hit->t < start.t;
    }
    return hit->t < start.t;
}

static int bvh_occluded_ordered( const PtSceneDesc* d, v3 o, v3 dir, float tMax, TravCount* cnt )
{
    int stack[128];
    int node = 0, top = 0;
    v3 inv   = V( 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z );
    int neg[3] = { inv.x < 0, inv.y < 0, inv.z < 0 };
    for ( ;; )
    {
        const PtBvhNode* n = &d->nodes[node];
        cnt->nodes++;
        if ( ray_aabb( o, inv, neg, n->aabb_min, n->aabb_max, tMax ) )
        {
            if ( n->num_shapes > 0 )
            {
                for ( int s = n->offset; s < n->offset + n->num_shapes; ++s )
                    if ( shape_test( d, (uint32_t)s, o, dir, tMax, cnt ) ) return 1;
                if ( top == 0 ) break;
                node = stack[--top];
            }
            else if ( neg[n->axis < 3 ? n->axis : 0] )
            {
                stack[top++] = node + 1;
                node         = n->offset;
            }
            else
            {
                stack[top++] = n->offset;
                node         = node + 1;
            }
        }
        else
        {
            if ( top == 0 ) break;
            node = stack[--top];
        }
    }
    return 0;
}

static int scene_intersect( const PtoScene* s, v3 o, v3 dir, Hit* hit, TravCount* cnt )
{
    return s->traversal == PT_TRAVERSAL_ORDERED ? bvh_intersect_ordered( &s->d, s->sphere_below, o, dir, hit, cnt ) : bvh_intersect( &s->d, o, dir, hit, cnt );
}

static int scene_occluded_any( const PtoScene* s, v3 o, v3 dir, float tMax, TravCount* cnt )
{
    return s->traversal == PT_TRAVERSAL_ORDERED ? bvh_occluded_ordered( &s->d, o, dir, tMax, cnt ) : bvh_occluded( &s->d, o, dir, tMax, cnt );
}

/* ------------------------------------------------------------------------------------------------ sampling */

/* ConcentricSampleDisk, src/sampling.cpp:15-38 (radius 1) */
static v2 concentric_disk( float u1, float u2 )
{
    v2 r = { 0, 0 };
    u1 = 2 * u1 - 1;
    u2 = 2 * u2 - 1;
    if ( u1 == 0 && u2 == 0 ) return r;
    float theta, rad = 1.0f;
    if ( u1 * u1 > u2 * u2 )
    {
        rad *= u1;
        theta = ( PTO_PI / 4 ) * ( u2 / u1 );
    }
    else
    {
        rad *= u2;
        theta = ( PTO_PI / 2 ) - ( PTO_PI / 4 ) * ( u1 / u2 );
    }
    r.x = rad * cosf( theta );
    r.y = rad * sinf( theta );
    return r;
}

/* CosineSampleHemisphere, src/sampling.cpp:48-53 */
static v3 cosine_hemisphere( float u1, float u2 )
{
    v2 d    = concentric_disk( u1, u2 );
    float z = sqrtf( max0( 1 - d.x * d.x - d.y * d.y ) );
    return V( d.x, d.y, z );
}

/* UniformSampleSphere, src/sampling.cpp:55-61 */
static v3 uniform_sphere( float u1, float u2 )
{
    float z     = 1 - 2 * u1;
    float r     = sqrtf( max0( 1 - z * z ) );
    float theta = 2 * PTO_PI * u2;
    return V( r * cosf( theta ), r * sinf( theta ), z );
}

/* UniformSampleTriangle, src/sampling.cpp:63-67 */
static v2 uniform_triangle( float u1, float u2 )
{
    float su0 = sqrtf( u1 );
    v2 r      = { 1 - su0, u2 * su0 };
    return r;
}

/* ------------------------------------------------------------------------------------------------ shading */

/* Texture::GetPixel, src/resource/texture.cpp:49-66 */
static void texture_fetch( const PtSceneDesc* d, int tex, float u, float v, float rgba[4] )
{
    const PtTexture* t = &d->textures[tex];
    u = fmodf( u, 1 );
    u = u < 0 ? u + 1 : u;
    v = fmodf( v, 1 );
    v = v < 0 ? v + 1 : v;
    int w = (int)( u * t->width );
    int h = (int)( v * t->height );
    if ( t->width - 1 < w ) w = t->width - 1;
    if ( t->height - 1 < h ) h = t->height - 1;
    const uint8_t* px = d->texels + 4 * ( t->texel_offset + (uint64_t)( h * t->width + w ) );
    for ( int c = 0; c < 4; ++c ) rgba[c] = ( 1.0f / 255.0f ) * (float)px[c];
}

/* Skybox::GetPixel( ray ), src/resource/skybox.cpp:84-118 (restated from source text, see pt_oracle.h) */
static void skybox_fetch( const PtSceneDesc* d, v3 v, float rgba[4] )
{
    v3 a = V( fabsf( v.x ), fabsf( v.y ), fabsf( v.z ) );
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
        uy   = -v.z; /* both branches of the reference's conditional are -v.z */
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
    int w = (int)( ux * d->skybox_width );
    int h = (int)( uy * d->skybox_height );
    if ( d->skybox_width - 1 < w ) w = d->skybox_width - 1;
    if ( d->skybox_height - 1 < h ) h = d->skybox_height - 1;
    const float* px = d->skybox + 4 * ( (size_t)face * d->skybox_width * d->skybox_height + (size_t)h * d->skybox_width + w );
    memcpy( rgba, px, 16 );
}

typedef struct
{
    const PtoScene* scene;
    PtRngState rng;
    PtoStats stats;
} Ctx;

static float draw( Ctx* c )
{
    c->stats.draws++;
    return pt_rng_next( &c->rng );
}

static int scene_occluded( Ctx* c, v3 o, v3 dir, float tMax )
{
    TravCount cnt = { 0, 0, 0, 0 };
    int r = scene_occluded_any( c->scene, o, dir, tMax, &cnt );
    c->stats.rays_shadow++;
    c->stats.nodes_shadow += cnt.nodes;
    c->stats.tri_tests_shadow += cnt.tris;
    c->stats.sphere_tests_shadow += cnt.spheres;
    return r;
}

/* Shape::SampleWithRespectToSolidAngle (src/shapes.cpp:11-30) over Triangle / Sphere::SampleWithRespectToArea
 * (src/shapes.cpp:104-115, 42-50).  The two uniforms are function arguments Rand(), Rand(); g++ evaluates
 * them right to left, so the FIRST draw is u2 and the second u1. */
static void sample_shape( Ctx* c, uint32_t prim, v3 p, v3* position, v3* normal, float* pdf )
{
    const PtSceneDesc* d = &c->scene->d;
    float second = draw( c ); /* u2 */
    float first  = draw( c ); /* u1 */
    if ( d->prims[prim].type == PT_PRIM_TRIANGLE )
    {
        size_t i[3];
        triangle_vertices( d, &d->triangles[d->prims[prim].index], i );
        v2 s    = uniform_triangle( first, second );
        float u = s.x, v = s.y, w = 1 - u - v;
        v3 v0 = ld3( d->vertices + 3 * i[0] ), v1 = ld3( d->vertices + 3 * i[1] ), v2_ = ld3( d->vertices + 3 * i[2] );
        *position = add( add( scale( v0, u ), scale( v1, v ) ), scale( v2_, w ) );
        *normal   = normalize( add( add( scale( ld3( d->normals + 3 * i[0] ), u ), scale( ld3( d->normals + 3 * i[1] ), v ) ), scale( ld3( d->normals + 3 * i[2] ), w ) ) );
        float area = 0.5f * length( cross( sub( v1, v0 ), sub( v2_, v0 ) ) ); /* Triangle::Area, src/shapes.cpp:96-102 */
        *pdf       = 1.0f / area;
    }
    else
    {
        const PtSphere* s = &d->spheres[d->prims[prim].index];
        v3 n      = uniform_sphere( first, second );
        *position = add( ld3( s->position ), scale( n, s->radius ) );
        *normal   = n;
        float area = 4 * PTO_PI * s->radius * s->radius; /* Sphere::Area, src/shapes.cpp:37-40 */
        *pdf       = 1.0f / area;
    }
    v3 wi = sub( *position, p );
    if ( length( wi ) == 0 )
        *pdf = 0;
    else
    {
        float r2 = dot( wi, wi );
        wi       = normalize( wi );
        *pdf *= r2 / absdot( *normal, neg( wi ) );
        if ( isinf( *pdf ) ) *pdf = 0.f;
    }
}

/* Light::Sample_Li for the three light types, src/lights.cpp:8-55 */
static v3 sample_light( Ctx* c, const PtLight* l, v3 p, v3* wi, float* pdf )
{
    v3 zero = V( 0, 0, 0 );
    v3 Lemit = ld3( l->Lemit );
    if ( l->type == PT_LIGHT_POINT )
    {
        v3 lp = ld3( l->position );
        *wi   = normalize( sub( lp, p ) );
        *pdf  = 1;
        float dist = length( sub( lp, p ) );
        if ( scene_occluded( c, p, *wi, dist ) ) return zero;
        return divs( Lemit, dist * dist );
    }
    if ( l->type == PT_LIGHT_DIRECTIONAL )
    {
        *wi  = neg( ld3( l->direction ) );
        *pdf = 1;
        if ( scene_occluded( c, p, *wi, FLT_MAX ) ) return zero;
        return Lemit;
    }
    v3 sp, sn;
    float spdf;
    sample_shape( c, (uint32_t)l->prim, p, &sp, &sn, &spdf );
    *wi  = normalize( sub( sp, p ) );
    *pdf = spdf;
    float dist = length( sub( sp, p ) );
    if ( (double)dist < 0.002 || scene_occluded( c, p, *wi, dist ) ) return zero;
    return dot( neg( *wi ), sn ) > 0 ? Lemit : zero;
}

/* Li, src/path_tracer.cpp:106-156, with LDirect / EstimateSingleDirect (:74-104), Material::ComputeBRDF /
 * GetAlbedo (src/resource/material.cpp:34-55) and BRDF::F / Sample_F / Pdf (:15-32) */
static v3 radiance( Ctx* c, v3 o, v3 dir )
{
    const PtSceneDesc* d = &c->scene->d;
    v3 L = V( 0, 0, 0 ), T = V( 1, 1, 1 );
    for ( int bounce = 0; bounce < d->max_depth; ++bounce )
    {
        Hit hit;
        memset( &hit, 0, sizeof( hit ) );
        hit.wo   = neg( dir );
        hit.t    = FLT_MAX; /* Scene::Intersect, src/scene.cpp:370-373 */
        hit.prim = -1;
        TravCount cnt = { 0, 0, 0, 0 };
        int found = scene_intersect( c->scene, o, dir, &hit, &cnt );
        c->stats.rays_closest++;
        c->stats.nodes_closest += cnt.nodes;
        c->stats.tri_tests_closest += cnt.tris;
        c->stats.sphere_tests_closest += cnt.spheres;
        c->stats.rays_redone += cnt.redone;
        if ( !found )
        {
            v3 env;
            if ( d->skybox ) /* Scene::LEnvironment, src/scene.cpp:386-393 */
            {
                float px[4];
                skybox_fetch( d, dir, px );
                env = V( px[0], px[1], px[2] );
            }
            else
                env = ld3( d->background_radiance );
            L = add( L, mul( T, env ) );
            break;
        }
        c->stats.shaded_hits++;
        hit.position = add( hit.position, scale( hit.normal, PTO_EPSILON ) );

        const PtMaterial* m = &d->materials[hit.material];
        if ( bounce == 0 && dot( hit.wo, hit.normal ) > 0 ) L = add( L, mul( T, ld3( m->Ke ) ) );

        v3 Kd = ld3( m->albedo );
        if ( m->albedo_texture >= 0 )
        {
            float px[4];
            texture_fetch( d, m->albedo_texture, hit.uv.x, hit.uv.y, px );
            Kd = mul( Kd, V( px[0], px[1], px[2] ) );
        }
        v3 F = divs( Kd, PTO_PI ); /* BRDF::F */

        v3 Ldirect = V( 0, 0, 0 );
        for ( uint64_t li = 0; li < d->num_lights; ++li )
        {
            const PtLight* l = &d->lights[li];
            v3 Ld = V( 0, 0, 0 );
            for ( int i = 0; i < l->n_samples; ++i )
            {
                v3 wi;
                float pdf;
                v3 Li = sample_light( c, l, hit.position, &wi, &pdf );
                if ( pdf == 0 ) continue; /* EstimateSingleDirect returns 0 */
                Ld = add( Ld, divs( scale( mul( F, Li ), absdot( hit.normal, wi ) ), pdf ) );
            }
            Ldirect = add( Ldirect, divs( Ld, (float)l->n_samples ) );
        }
        L = add( L, mul( T, Ldirect ) );

        float second = draw( c ); /* CosineSampleHemisphere( Rand(), Rand() ): first draw is u2 */
        float first  = draw( c );
        v3 local     = cosine_hemisphere( first, second );
        v3 wi        = add( add( scale( hit.tangent, local.x ), scale( hit.bitangent, local.y ) ), scale( hit.normal, local.z ) );
        int same     = dot( hit.wo, hit.normal ) * dot( wi, hit.normal ) > 0;
        float pdf    = same ? absdot( wi, hit.normal ) / PTO_PI : 0;
        if ( pdf == 0.f || is_zero( F ) ) break;
        T = mul( T, divs( scale( F, absdot( wi, hit.normal ) ), pdf ) );
        if ( is_zero( T ) ) break;
        o   = hit.position;
        dir = wi;
    }
    return L;
}

/* ------------------------------------------------------------------------------------------------ entry points */

int pto_intersect( const PtoScene* s, uint64_t n, const float* origins, const float* dirs, int32_t* prim, float* t, float* u,
                   float* v, float* position, float* normal, float* tangent, float* bitangent, float* uv, int32_t* material,
                   uint32_t* nodes_visited, uint32_t* prim_tests )
{
    if ( !s || !origins || !dirs ) return fail( "pto_intersect: null argument" );
    #pragma omp parallel for schedule( dynamic, 256 )
    for ( int64_t i = 0; i < (int64_t)n; ++i )
    {
        Hit hit;
        memset( &hit, 0, sizeof( hit ) );
        hit.t    = FLT_MAX;
        hit.prim = -1;
        TravCount cnt = { 0, 0, 0, 0 };
        scene_intersect( s, ld3( origins + 3 * i ), ld3( dirs + 3 * i ), &hit, &cnt );
        if ( prim ) prim[i] = hit.prim;
        if ( t ) t[i] = hit.t;
        if ( u ) u[i] = hit.bu;
        if ( v ) v[i] = hit.bv;
        if ( hit.prim >= 0 )
        {
            if ( position ) st3( position + 3 * i, hit.position );
            if ( normal ) st3( normal + 3 * i, hit.normal );
            if ( tangent ) st3( tangent + 3 * i, hit.tangent );
            if ( bitangent ) st3( bitangent + 3 * i, hit.bitangent );
            if ( uv ) { uv[2 * i] = hit.uv.x; uv[2 * i + 1] = hit.uv.y; }
        }
        if ( material ) material[i] = hit.prim >= 0 ? hit.material : -1;
        if ( nodes_visited ) nodes_visited[i] = (uint32_t)cnt.nodes;
        if ( prim_tests ) prim_tests[i] = (uint32_t)( cnt.tris + cnt.spheres );
    }
    return 0;
}

int pto_occluded( const PtoScene* s, uint64_t n, const float* origins, const float* dirs, const float* t_max, uint8_t* occluded,
                  uint32_t* nodes_visited, uint32_t* prim_tests )
{
    if ( !s || !origins || !dirs || !t_max || !occluded ) return fail( "pto_occluded: null argument" );
    #pragma omp parallel for schedule( dynamic, 256 )
    for ( int64_t i = 0; i < (int64_t)n; ++i )
    {
        TravCount cnt = { 0, 0, 0, 0 };
        occluded[i] = (uint8_t)scene_occluded_any( s, ld3( origins + 3 * i ), ld3( dirs + 3 * i ), t_max[i], &cnt );
        if ( nodes_visited ) nodes_visited[i] = (uint32_t)cnt.nodes;
        if ( prim_tests ) prim_tests[i] = (uint32_t)( cnt.tris + cnt.spheres );
    }
    return 0;
}

int pto_li_image( const PtoScene* s, int32_t spp_total, int32_t s0, int32_t s1, uint32_t rng_mode, uint64_t seed, const float* table,
                  uint64_t table_len, int32_t draws_per_sample, int32_t threads, float* rgb_sum, PtoStats* stats )
{
    if ( !s || !rgb_sum ) return fail( "pto_li_image: null argument" );
    if ( rng_mode == 1 && ( !table || table_len == 0 ) ) return fail( "pto_li_image: table mode without a table" );
    const PtCamera* cam = &s->d.camera;
    const int W = s->d.image_width, H = s->d.image_height;
    /* camera frame, src/path_tracer.cpp:168-173 */
    v3 position = ld3( cam->position ), view = ld3( cam->view_dir ), up = ld3( cam->up_dir ), right = ld3( cam->right_dir );
    float halfHeight = tanf( cam->vfov / 2 );
    float halfWidth  = halfHeight * cam->aspect_ratio;
    v3 UL = sub( add( add( position, view ), scale( up, halfHeight ) ), scale( right, halfWidth ) );
    v3 dU = scale( right, 2 * halfWidth / W );
    v3 dV = scale( neg( up ), 2 * halfHeight / H );
    UL    = add( UL, scale( add( dU, dV ), 0.5f ) );

    PtoStats total;
    memset( &total, 0, sizeof( total ) );
#ifdef _OPENMP
    int nthreads = threads > 0 ? threads : omp_get_max_threads();
#else
    int nthreads = 1;
    (void)threads;
#endif
    #pragma omp parallel for schedule( dynamic ) num_threads( nthreads )
    for ( int row = 0; row < H; ++row )
    {
        Ctx c;
        memset( &c, 0, sizeof( c ) );
        c.scene               = s;
        c.rng.mode            = rng_mode;
        c.rng.seed            = seed;
        c.rng.table           = table;
        c.rng.table_len       = table_len;
        c.rng.spp_total       = spp_total;
        c.rng.draws_per_sample = draws_per_sample;
        for ( int col = 0; col < W; ++col )
        {
            v3 imagePlanePos = add( add( UL, scale( dV, (float)row ) ), scale( dU, (float)col ) ); /* :183 */
            v3 sum = V( 0, 0, 0 );
            for ( int smp = s0; smp < s1; ++smp )
            {
                c.rng.pixel  = (uint32_t)( row * W + col );
                c.rng.sample = (uint32_t)smp;
                c.rng.draw   = 0;
                /* AntiAlias::Jitter, src/anti_aliasing.cpp:89-92; RandFloat(l,h) = Rand()*(h-l)+l, random.cpp:18-21.
                 * C++ leaves the order of the two RandFloat calls in `c + R()*dU + R()*dV` unspecified; the
                 * reference as built by g++ 11.4 -O3 (oracle/Makefile.ref) makes the dV call first. */
                float jv = draw( &c ) * ( 0.5f - -0.5f ) + -0.5f;
                float ju = draw( &c ) * ( 0.5f - -0.5f ) + -0.5f;
                v3 aa    = add( add( imagePlanePos, scale( dU, ju ) ), scale( dV, jv ) );
                v3 dir   = normalize( sub( aa, position ) ); /* :189, with cam.position as intended */
                sum      = add( sum, radiance( &c, position, dir ) );
                c.stats.samples++;
            }
            st3( rgb_sum + 3 * ( (size_t)row * W + col ), sum );
        }
        #pragma omp critical
        {
            uint64_t* dst       = (uint64_t*)&total;
            const uint64_t* src = (const uint64_t*)&c.stats;
            for ( size_t k = 0; k < sizeof( PtoStats ) / sizeof( uint64_t ); ++k ) dst[k] += src[k];
        }
    }
    if ( stats ) *stats = total;
    return 0;
}

int pto_max_threads( void )
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Uncharted2TonemapHelper, src/tonemap.cpp:36-45 */
static float u2_helper( float x )
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ( ( x * ( A * x + C * B ) + D * E ) / ( x * ( A * x + B ) + D * F ) ) - E / F;
}

int pto_tonemap( uint64_t n, const float* rgb, float exposure, float gamma, uint32_t op, float* out )
{
    if ( !rgb || !out ) return fail( "pto_tonemap: null argument" );
    float whiteScale = 1.0f / u2_helper( 11.2f ); /* src/tonemap.cpp:50 */
    float invGamma   = 1.0f / gamma;              /* src/tonemap.cpp:6-9: pow( pixel, vec3( 1 ) / vec3( gamma ) ) */
    for ( uint64_t i = 0; i < 3 * n; ++i )
    {
        float c = rgb[i];
        if ( op == PT_TONEMAP_UNCHARTED2 )
            c = u2_helper( exposure * c ) * whiteScale; /* src/tonemap.cpp:47-53 */
        else if ( op == PT_TONEMAP_REINHARD )
        {
            float a = exposure * c; /* src/tonemap.cpp:29-33 */
            c       = a / ( 1.0f + a );
        }
        if ( op != PT_TONEMAP_NONE )
        {
            c = powf( c, invGamma );
            c = c < 0.0f ? 0.0f : c; /* glm::clamp = min( max( x, 0 ), 1 ) */
            c = 1.0f < c ? 1.0f : c;
        }
        out[i] = c;
    }
    return 0;
}

int pto_ray_aabb( uint64_t n, const float* o, const float* inv_dir, const float* bmin, const float* bmax, const float* max_t, uint8_t* hit )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        v3 inv     = ld3( inv_dir + 3 * i );
        int neg[3] = { inv.x < 0, inv.y < 0, inv.z < 0 };
        hit[i]     = (uint8_t)ray_aabb( ld3( o + 3 * i ), inv, neg, bmin + 3 * i, bmax + 3 * i, max_t[i] );
    }
    return 0;
}

int pto_ray_triangle( uint64_t n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, const float* max_t,
                      uint8_t* hit, float* t, float* u, float* v )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        float tt = 0, uu = 0, vv = 0;
        hit[i] = (uint8_t)ray_triangle( ld3( o + 3 * i ), ld3( d + 3 * i ), ld3( v0 + 3 * i ), ld3( v1 + 3 * i ), ld3( v2 + 3 * i ), &tt, &uu, &vv, max_t[i] );
        t[i] = tt; u[i] = uu; v[i] = vv;
    }
    return 0;
}

int pto_ray_sphere( uint64_t n, const float* o, const float* d, const float* centre, const float* radius, const float* max_t, uint8_t* hit, float* t )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        float tt = 0;
        hit[i] = (uint8_t)ray_sphere( ld3( o + 3 * i ), ld3( d + 3 * i ), ld3( centre + 3 * i ), radius[i], &tt, max_t[i] );
        t[i]   = tt;
    }
    return 0;
}

int pto_sampling( uint64_t n, const float* u1, const float* u2, float* cosine, float* disk, float* sphere, float* triangle )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        st3( cosine + 3 * i, cosine_hemisphere( u1[i], u2[i] ) );
        v2 k = concentric_disk( u1[i], u2[i] );
        disk[2 * i] = k.x; disk[2 * i + 1] = k.y;
        st3( sphere + 3 * i, uniform_sphere( u1[i], u2[i] ) );
        v2 tr = uniform_triangle( u1[i], u2[i] );
        triangle[2 * i] = tr.x; triangle[2 * i + 1] = tr.y;
    }
    return 0;
}

int pto_texture_fetch( const PtoScene* s, int32_t texture, uint64_t n, const float* u, const float* v, float* rgba )
{
    if ( !s || texture < 0 || (uint64_t)texture >= s->d.num_textures ) return fail( "pto_texture_fetch: bad texture" );
    for ( uint64_t i = 0; i < n; ++i ) texture_fetch( &s->d, texture, u[i], v[i], rgba + 4 * i );
    return 0;
}

int pto_skybox_fetch( const PtoScene* s, uint64_t n, const float* dirs, float* rgba )
{
    if ( !s || !s->d.skybox ) return fail( "pto_skybox_fetch: scene has no skybox" );
    for ( uint64_t i = 0; i < n; ++i ) skybox_fetch( &s->d, ld3( dirs + 3 * i ), rgba + 4 * i );
    return 0;
}
