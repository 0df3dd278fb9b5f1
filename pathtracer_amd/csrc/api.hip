// C-ABI of the path-tracing core (include/pt_api.h): scene validation and upload, the render loop that
// sequences the wavefront kernels, readback, batch ray queries.
#include "pt_api.h"
#include "layout_build.h"
#include "pt_device.h"
#include "wide_bvh.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <queue>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

using namespace ptd;

// Staging memory of pt_trace_closest_static, the entry point that allocates nothing: one array per GPU, part of the
// library's code object, alive until the process ends.  Per ray: origin and direction (24 bytes) in, primitive, t,
// u, v and the two counters (24 bytes) out; eight sections of kStaticRays elements each.
constexpr uint64_t kStaticRays = PT_TRACE_STATIC_MAX_RAYS;
static __device__ __align__( 256 ) unsigned char scratch[kStaticRays * 48];

namespace
{

thread_local std::string g_error;

// Device address of `scratch` on every GPU this process has uploaded a scene to.  The array's name is a host-side
// handle, not an address: only what hipGetSymbolAddress returns, with that GPU current, may go to a kernel or a copy.
constexpr int kMaxDevices = 64;
std::mutex g_scratchMutex;
unsigned char* g_scratchPtr[kMaxDevices] = {};

// Must run with `device` current.  Allocates nothing: the array exists as soon as the code object is loaded.
hipError_t ScratchAddress( int device, unsigned char** out )
{
    if ( device < 0 || device >= kMaxDevices ) return hipErrorInvalidDevice;
    std::lock_guard< std::mutex > lock( g_scratchMutex );
    if ( !g_scratchPtr[device] )
    {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress( &p, HIP_SYMBOL( scratch ) );
        if ( e != hipSuccess ) return e;
        g_scratchPtr[device] = static_cast< unsigned char* >( p );
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

int Fail( int code, const char* fmt, ... )
{
    char buf[512];
    va_list ap;
    va_start( ap, fmt );
    vsnprintf( buf, sizeof( buf ), fmt, ap );
    va_end( ap );
    g_error = buf;
    return code;
}

} // namespace

namespace ptd
{
// for the other source files of the library: sets what pt_last_error returns on this thread
int SetError( int code, const char* message )
{
    g_error = message;
    return code;
}
} // namespace ptd

namespace
{

// The C-ABI never lets an exception out (std::bad_alloc from the host-side staging vectors is the one that can
// happen): every entry point is a function-try-block that ends here.
int HostFailure( const char* where )
{
    try
    {
        throw;
    }
    catch ( const std::bad_alloc& )
    {
        return Fail( PT_ERR_DEVICE, "%s: out of host memory", where );
    }
    catch ( const std::exception& e )
    {
        return Fail( PT_ERR_DEVICE, "%s: %s", where, e.what() );
    }
    catch ( ... )
    {
        return Fail( PT_ERR_DEVICE, "%s: unknown exception", where );
    }
}

#define HIP_TRY( expr )                                                                                                  \
    do                                                                                                                   \
    {                                                                                                                    \
        hipError_t e_ = ( expr );                                                                                        \
        if ( e_ != hipSuccess )                                                                                          \
        {                                                                                                                \
            (void)hipGetLastError(); /* reported here; a later call must not find it again */                           \
            return Fail( PT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString( e_ ) );                              \
        }                                                                                                                \
    } while ( 0 )

// Device allocation that frees itself.
struct DeviceBuffer
{
    void* ptr    = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer( const DeviceBuffer& ) = delete;
    DeviceBuffer& operator=( const DeviceBuffer& ) = delete;
    ~DeviceBuffer() { Release(); }
    void Release()
    {
        if ( ptr ) (void)hipFree( ptr );
        ptr   = nullptr;
        bytes = 0;
    }
    hipError_t Allocate( size_t n )
    {
        Release();
        if ( n == 0 ) return hipSuccess;
        hipError_t e = hipMalloc( &ptr, n );
        if ( e == hipSuccess ) bytes = n;
        return e;
    }
    hipError_t Upload( const void* src, size_t n )
    {
        hipError_t e = Allocate( n );
        if ( e != hipSuccess || n == 0 ) return e;
        return hipMemcpy( ptr, src, n, hipMemcpyHostToDevice );
    }
    template < typename T >
    T* As() const { return static_cast< T* >( ptr ); }
};

struct f3h
{
    float x, y, z;
};
f3h operator-( f3h a, f3h b ) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
f3h Cross( f3h x, f3h y ) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
float Dot( f3h a, f3h b ) { return ( a.x * b.x + a.y * b.y ) + a.z * b.z; }
f3h Load3( const float* p ) { return { p[0], p[1], p[2] }; }

constexpr float kPiHost = 3.14159265358979323846f;

// The per-node and per-primitive loops of an upload, spread over host threads once they are long enough to pay for
// the threads (a 20 M triangle scene: 3.7 s of record building on one thread).  body( begin, end ) must not throw and
// must touch only what belongs to its range; the ranges are handed out in chunks, smallest index first.
template < typename Body >
void ParallelFor( uint64_t n, const Body& body )
{
    constexpr uint64_t kChunk = 1u << 16;
    unsigned threads = std::thread::hardware_concurrency();
    if ( threads > 16 ) threads = 16; // a share of a large machine, not all of it
    if ( threads < 2 || n < 4 * kChunk )
    {
        if ( n ) body( uint64_t( 0 ), n );
        return;
    }
    std::atomic< uint64_t > next{ 0 };
    auto work = [&] {
        for ( ;; )
        {
            const uint64_t begin = next.fetch_add( kChunk );
            if ( begin >= n ) return;
            body( begin, begin + kChunk < n ? begin + kChunk : n );
        }
    };
    std::vector< std::thread > pool;
    pool.reserve( threads - 1 );
    for ( unsigned t = 1; t < threads; ++t ) pool.emplace_back( work );
    work();
    for ( std::thread& t : pool ) t.join();
}

// The tree cut into its top and the subtrees below it, for passes in which a node depends on its parent or on its
// children only: the nodes of a subtree are a contiguous range [root, escape[root]) of the reference's pre-order,
// parents before children.  `upper` (ascending: a parent's index is below its children's) is split largest subtree first.
struct TreeCut
{
    std::vector< uint32_t > upper, roots;
};

TreeCut CutTree( const PtSceneDesc* d, const std::vector< uint32_t >& escape )
{
    TreeCut cut;
    const uint32_t N = static_cast< uint32_t >( d->num_nodes );
    std::priority_queue< std::pair< uint32_t, uint32_t > > open; // (nodes in the subtree, its root)
    open.push( { N, 0u } );
    while ( !open.empty() && open.size() + cut.roots.size() < 1024 )
    {
        const uint32_t size = open.top().first, x = open.top().second;
        open.pop();
        if ( d->nodes[x].num_shapes > 0 || size < 8192 )
        {
            cut.roots.push_back( x );
            continue;
        }
        cut.upper.push_back( x );
        const uint32_t a = x + 1, b = static_cast< uint32_t >( d->nodes[x].offset );
        open.push( { escape[a] - a, a } );
        open.push( { escape[b] - b, b } );
    }
    for ( ; !open.empty(); open.pop() ) cut.roots.push_back( open.top().second );
    std::sort( cut.upper.begin(), cut.upper.end() );
    return cut;
}

// body( root ) for every subtree of the cut, on several host threads when the tree is large
template < typename Body >
void ForEachSubtree( const TreeCut& cut, uint64_t numNodes, const Body& body )
{
    std::atomic< size_t > next{ 0 };
    auto work = [&] {
        for ( size_t k = next.fetch_add( 1 ); k < cut.roots.size(); k = next.fetch_add( 1 ) ) body( cut.roots[k] );
    };
    unsigned threads = std::thread::hardware_concurrency();
    if ( threads > 16 ) threads = 16;
    if ( numNodes < ( 1u << 18 ) || threads < 2 ) threads = 1;
    std::vector< std::thread > pool;
    for ( unsigned t = 1; t < threads; ++t ) pool.emplace_back( work );
    work();
    for ( std::thread& t : pool ) t.join();
}

} // namespace

struct PtScene
{
    int device = 0;
    hipStream_t stream = nullptr;
    int gridBlocks = 2048;
    int extendBlocks = 1024, connectBlocks = 1024; // resident grids of the traversal kernels
    int wideBlocks = 512, wideThreads = 1024, wideLdsStack = 8; // the wide closest-hit kernel: resident grid, block size, stack entries it keeps in LDS
    int wideShadowBlocks = 512, wideShadowThreads = 768;       // the wide any-hit kernel
    uint32_t topNodes = 0;                          // nodes at the front of the device array in breadth-first order
    bool directSum = false;                         // one light with one sample: ConnectKernel adds the direct light, no ResolveKernel
    bool oneDirectionalLight = false;               // that light is directional; towardsLight = -direction (wi of src/lights.cpp:30)
    float towardsLight[3] = { 0, 0, 0 };

    // scene
    DeviceBuffer wideNodes, widePrims, wideSpill, lightWideNodes;
    DeviceBuffer nodes, orderedNodes, shadowNodes, shadowTopWords, prims, triShade, spheres, materials, textures, texels, lights, skybox;
    SceneView view{};
    PtCamera camera{};
    int width = 0, height = 0, maxDepth = 0;
    uint64_t bytesPerPathFixed = 0;

    // path state
    DeviceBuffer rayO0, rayO1, rayD0, rayD1, thr0, thr1, rad, hit, shadowO, shDir, shEst, queue0, queue1, shadowPaths, shadowRays, sphereQueue, redo, counts, cursors;
    PathBuffers paths{};
    DeviceBuffer sampleTable;

    // film and outputs
    DeviceBuffer film, stats, linearOut, tonemappedOut;
    bool rendered = false;

    // host-side statistics
    uint64_t samples = 0;
    uint64_t launches[PT_K_COUNT] = {};
    double kernelMs[PT_K_COUNT]   = {};
    double renderMs               = 0;

    struct TimedSpan
    {
        int kernel;
        hipEvent_t start, stop;
    };
    std::vector< hipEvent_t > eventPool;
    size_t eventsUsed = 0;
    std::vector< TimedSpan > spans;

    ~PtScene()
    {
        for ( hipEvent_t e : eventPool ) (void)hipEventDestroy( e );
        if ( stream ) (void)hipStreamDestroy( stream );
    }
};

namespace
{

// ------------------------------------------------------------------------------------------------ validation

int Validate( const PtSceneDesc* d, std::vector< uint32_t >& escape )
{
    if ( d->struct_size != sizeof( PtSceneDesc ) ) return Fail( PT_ERR_INVALID_ARG, "PtSceneDesc::struct_size is %u, expected %zu", d->struct_size, sizeof( PtSceneDesc ) );
    if ( d->version != PT_API_VERSION ) return Fail( PT_ERR_INVALID_ARG, "PtSceneDesc::version is %u, this library implements %u", d->version, PT_API_VERSION );
    if ( !d->nodes || d->num_nodes == 0 || !d->prims || d->num_prims == 0 ) return Fail( PT_ERR_INVALID_SCENE, "scene has no BVH nodes or no primitives" );
    if ( d->num_prims > kMaxPrims ) return Fail( PT_ERR_INVALID_SCENE, "more than %u primitives are not supported", kMaxPrims );
    if ( d->num_nodes > kMaxNodes ) return Fail( PT_ERR_INVALID_SCENE, "more than %u BVH nodes are not supported", kMaxNodes );
    if ( d->num_nodes >= 0xFFFFFFFFull ) return Fail( PT_ERR_INVALID_SCENE, "too many BVH nodes" );
    if ( d->image_width <= 0 || d->image_height <= 0 ) return Fail( PT_ERR_INVALID_SCENE, "image size %d x %d", d->image_width, d->image_height );
    if ( static_cast< uint64_t >( d->image_width ) * d->image_height > 0x7fffffffull ) return Fail( PT_ERR_INVALID_SCENE, "image too large" );
    if ( d->max_depth < 0 ) return Fail( PT_ERR_INVALID_SCENE, "max_depth %d", d->max_depth );
    if ( ( d->num_triangles && !d->triangles ) || ( d->num_spheres && !d->spheres ) || ( d->num_meshes && !d->meshes ) ||
         ( d->num_materials && !d->materials ) || ( d->num_textures && !d->textures ) || ( d->num_lights && !d->lights ) ||
         ( d->num_texels && !d->texels ) || ( d->num_vertices && ( !d->vertices || !d->normals || !d->tangents || !d->uvs ) ) )
        return Fail( PT_ERR_INVALID_ARG, "an array pointer is null while its count is not zero" );

    for ( uint64_t i = 0; i < d->num_textures; ++i )
    {
        const PtTexture& t = d->textures[i];
        if ( t.width <= 0 || t.height <= 0 || t.texel_offset + static_cast< uint64_t >( t.width ) * t.height > d->num_texels )
            return Fail( PT_ERR_INVALID_SCENE, "texture %llu (%d x %d at %llu) lies outside the texel pool", (unsigned long long)i, t.width, t.height, (unsigned long long)t.texel_offset );
    }
    for ( uint64_t i = 0; i < d->num_materials; ++i )
        if ( d->materials[i].albedo_texture < -1 || d->materials[i].albedo_texture >= static_cast< int64_t >( d->num_textures ) )
            return Fail( PT_ERR_INVALID_SCENE, "material %llu refers to texture %d", (unsigned long long)i, d->materials[i].albedo_texture );
    for ( uint64_t i = 0; i < d->num_meshes; ++i )
    {
        const PtMesh& m = d->meshes[i];
        if ( static_cast< uint64_t >( m.vertex_offset ) + m.num_vertices > d->num_vertices ) return Fail( PT_ERR_INVALID_SCENE, "mesh %llu exceeds the vertex arrays", (unsigned long long)i );
        if ( m.material < 0 || m.material >= static_cast< int64_t >( d->num_materials ) ) return Fail( PT_ERR_INVALID_SCENE, "mesh %llu refers to material %d", (unsigned long long)i, m.material );
    }
    for ( uint64_t i = 0; i < d->num_triangles; ++i )
    {
        const PtTriangle& t = d->triangles[i];
        if ( t.mesh >= d->num_meshes ) return Fail( PT_ERR_INVALID_SCENE, "triangle %llu refers to mesh %u", (unsigned long long)i, t.mesh );
        const uint32_t nv = d->meshes[t.mesh].num_vertices;
        if ( t.i0 >= nv || t.i1 >= nv || t.i2 >= nv ) return Fail( PT_ERR_INVALID_SCENE, "triangle %llu has a vertex index outside its mesh", (unsigned long long)i );
    }
    for ( uint64_t i = 0; i < d->num_spheres; ++i )
        if ( d->spheres[i].material < 0 || d->spheres[i].material >= static_cast< int64_t >( d->num_materials ) )
            return Fail( PT_ERR_INVALID_SCENE, "sphere %llu refers to material %d", (unsigned long long)i, d->spheres[i].material );
    for ( uint64_t i = 0; i < d->num_prims; ++i )
    {
        const PtPrim& p = d->prims[i];
        if ( !( ( p.type == PT_PRIM_TRIANGLE && p.index < d->num_triangles ) || ( p.type == PT_PRIM_SPHERE && p.index < d->num_spheres ) ) )
            return Fail( PT_ERR_INVALID_SCENE, "primitive %llu (type %u, index %u) is out of range", (unsigned long long)i, p.type, p.index );
    }
    for ( uint64_t i = 0; i < d->num_lights; ++i )
    {
        const PtLight& l = d->lights[i];
        if ( l.type > PT_LIGHT_AREA ) return Fail( PT_ERR_INVALID_SCENE, "light %llu has unknown type %u", (unsigned long long)i, l.type );
        if ( l.n_samples < 1 ) return Fail( PT_ERR_INVALID_SCENE, "light %llu has n_samples %d (must be >= 1)", (unsigned long long)i, l.n_samples );
        if ( l.type == PT_LIGHT_AREA && ( l.prim < 0 || static_cast< uint64_t >( l.prim ) >= d->num_prims ) )
            return Fail( PT_ERR_INVALID_SCENE, "area light %llu refers to primitive %d", (unsigned long long)i, l.prim );
    }
    if ( d->skybox && ( d->skybox_width <= 0 || d->skybox_height <= 0 ) ) return Fail( PT_ERR_INVALID_SCENE, "skybox size %d x %d", d->skybox_width, d->skybox_height );

    // BVH topology: pre-order, first child at index + 1, second child inside the parent's span.  The escape
    // index of a node is the first node after its subtree; a forward pass fills it in.
    const uint32_t N = static_cast< uint32_t >( d->num_nodes );
    const uint32_t unset = 0xFFFFFFFFu;
    escape.assign( N, unset );
    escape[0] = N;
    for ( uint32_t i = 0; i < N; ++i )
    {
        const PtBvhNode& n = d->nodes[i];
        if ( escape[i] == unset ) return Fail( PT_ERR_INVALID_SCENE, "BVH node %u is not reachable from the root", i );
        if ( n.num_shapes > 0 )
        {
            if ( n.num_shapes > kMaxLeafShapes ) return Fail( PT_ERR_INVALID_SCENE, "BVH leaf %u holds %u shapes; at most %d are supported", i, n.num_shapes, kMaxLeafShapes );
            if ( n.offset < 0 || static_cast< uint64_t >( n.offset ) + n.num_shapes > d->num_prims ) return Fail( PT_ERR_INVALID_SCENE, "BVH leaf %u covers primitives outside the list", i );
            if ( escape[i] != i + 1 ) return Fail( PT_ERR_INVALID_SCENE, "BVH node %u: array is not in depth-first pre-order", i );
        }
        else
        {
            const int64_t second = n.offset;
            if ( second <= static_cast< int64_t >( i ) + 1 || second >= static_cast< int64_t >( escape[i] ) || second >= N )
                return Fail( PT_ERR_INVALID_SCENE, "BVH node %u: second child %lld is outside its subtree", i, (long long)second );
            if ( escape[i + 1] != unset || escape[second] != unset ) return Fail( PT_ERR_INVALID_SCENE, "BVH node %u: a child has two parents", i );
            escape[i + 1]  = static_cast< uint32_t >( second );
            escape[second] = escape[i];
        }
    }
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------ upload

const float* TriVertex( const PtSceneDesc* d, const float* attribute, const PtTriangle& t, uint32_t index, int stride )
{
    return attribute + static_cast< size_t >( stride ) * ( d->meshes[t.mesh].vertex_offset + index );
}

int Upload( PtScene* s, const PtSceneDesc* d, const std::vector< uint32_t >& escape )
{
    const uint32_t N = static_cast< uint32_t >( d->num_nodes );
    const uint64_t P = d->num_prims;
    // PTAMD_UPLOAD_TIMING in the environment: the wall time of each part of the upload on stderr
    const bool timing = std::getenv( "PTAMD_UPLOAD_TIMING" ) != nullptr;
    auto phaseStart   = std::chrono::steady_clock::now();
    auto phase = [&]( const char* name ) {
        if ( !timing ) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf( stderr, "[pt_upload_scene] %-28s %8.1f ms\n", name, std::chrono::duration< double, std::milli >( now - phaseStart ).count() );
        phaseStart = now;
    };

    // Device order of the nodes: the top of the tree first (contiguous: the traversal kernels copy its beginning
    // into LDS), sibling pairs in the order of their parents' box area, everything below in the reference's pre-order.  The traversal follows stored
    // links, so the visit sequence is the reference's whatever the order in memory.
    uint32_t top = d->top_nodes ? d->top_nodes : kTopNodes;
    if ( top > N ) top = N;
    // The records are written on the device from the description as it is (layout_build.h); PT_SCENE_HOST_LAYOUTS keeps
    // the host loops below, which produce the same bytes
    const bool hostLayouts = ( d->flags & PT_SCENE_HOST_LAYOUTS ) != 0;
    std::vector< uint32_t > place; // reference index -> device index (host path)
    std::vector< uint32_t > bfs; // the top of the tree: the root, then sibling pairs
    {
        bfs.reserve( top );
        if ( top > 0 ) bfs.push_back( 0 );
        if ( !( d->flags & PT_SCENE_TOP_BREADTH_FIRST ) )
        {
            // largest box first: a node is visited when its parent's box is hit, and a bigger box is hit by more rays
            // (against plain breadth-first order, PT_SCENE_TOP_BREADTH_FIRST: closest-hit kernel -0.5 %)
            auto area = [&]( uint32_t i ) {
                const PtBvhNode& n = d->nodes[i];
                const float dx = n.aabb_max[0] - n.aabb_min[0], dy = n.aabb_max[1] - n.aabb_min[1], dz = n.aabb_max[2] - n.aabb_min[2];
                return dx * dy + dy * dz + dz * dx;
            };
            std::priority_queue< std::pair< float, uint32_t > > open; // inner nodes whose children are not placed yet
            if ( top > 0 && d->nodes[0].num_shapes == 0 ) open.push( { area( 0 ), 0u } );
            while ( !open.empty() && bfs.size() + 2 <= top )
            {
                const uint32_t x = open.top().second;
                open.pop();
                const uint32_t a = x + 1, b = static_cast< uint32_t >( d->nodes[x].offset );
                bfs.push_back( a );
                bfs.push_back( b );
                if ( d->nodes[a].num_shapes == 0 ) open.push( { area( a ), a } );
                if ( d->nodes[b].num_shapes == 0 ) open.push( { area( b ), b } );
            }
        }
        else
        for ( size_t head = 0; head < bfs.size() && bfs.size() < top; ++head )
        {
            const PtBvhNode& n = d->nodes[bfs[head]];
            if ( n.num_shapes > 0 ) continue;
            if ( bfs.size() + 2 > top ) break; // siblings stay together
            bfs.push_back( bfs[head] + 1 );
            bfs.push_back( static_cast< uint32_t >( n.offset ) );
        }
        if ( hostLayouts )
        {
            place.assign( N + 1, 0xFFFFFFFFu );
            uint32_t next = 0;
            for ( uint32_t i : bfs ) place[i] = next++;
            for ( uint32_t i = 0; i < N; ++i )
                if ( place[i] == 0xFFFFFFFFu ) place[i] = next++;
            place[N] = N; // the end of the traversal
        }
        s->topNodes = static_cast< uint32_t >( bfs.size() );
    }
    // the top of the tree by reference index, for the searches below and on the device
    std::vector< std::pair< uint32_t, uint32_t > > topByIndex( bfs.size() ); // (reference index, position in the top)
    for ( uint32_t k = 0; k < bfs.size(); ++k ) topByIndex[k] = { bfs[k], k };
    std::sort( topByIndex.begin(), topByIndex.end() );
    auto inTop = [&]( uint32_t i ) {
        const auto it = std::lower_bound( topByIndex.begin(), topByIndex.end(), std::make_pair( i, 0u ) );
        return it != topByIndex.end() && it->first == i;
    };

    // which subtrees hold a sphere (the front-to-back layouts enter their boxes whatever has been found, see kernels.hip);
    // children lie behind their parent in the reference's pre-order, so one backward pass does it
    std::vector< uint8_t > sphereBelow;
    if ( d->num_spheres > 0 )
    {
        sphereBelow.assign( N, 0 );
        for ( uint32_t i = N; i-- > 0; )
        {
            const PtBvhNode& n = d->nodes[i];
            if ( n.num_shapes > 0 )
                for ( uint32_t k = 0; k < n.num_shapes; ++k ) sphereBelow[i] |= d->prims[static_cast< uint64_t >( n.offset ) + k].type == PT_PRIM_SPHERE;
            else
                sphereBelow[i] = sphereBelow[i + 1] | sphereBelow[static_cast< uint32_t >( n.offset )];
        }
    }

    // the description on the device, for the kernels that write the records; released when the upload returns
    DeviceBuffer rawNodes, rawEscape, rawPrims, rawTriangles, rawMeshes, rawVertices, rawNormals, rawTangents, rawUvs, rawSpheres, rawSphereBelow;
    DeviceBuffer dTopSorted, dTopRank; // the top of the tree by reference index, and where each of its nodes is stored
    LayoutInputs in;
    if ( !hostLayouts )
    {
        HIP_TRY( rawNodes.Upload( d->nodes, 32ull * N ) );
        HIP_TRY( rawEscape.Upload( escape.data(), 4ull * N ) );
        HIP_TRY( rawPrims.Upload( d->prims, sizeof( PtPrim ) * P ) );
        HIP_TRY( rawTriangles.Upload( d->triangles, sizeof( PtTriangle ) * d->num_triangles ) );
        HIP_TRY( rawMeshes.Upload( d->meshes, sizeof( PtMesh ) * d->num_meshes ) );
        HIP_TRY( rawVertices.Upload( d->vertices, 12ull * d->num_vertices ) );
        HIP_TRY( rawNormals.Upload( d->normals, 12ull * d->num_vertices ) );
        HIP_TRY( rawTangents.Upload( d->tangents, 12ull * d->num_vertices ) );
        HIP_TRY( rawUvs.Upload( d->uvs, 8ull * d->num_vertices ) );
        HIP_TRY( rawSpheres.Upload( d->spheres, sizeof( PtSphere ) * d->num_spheres ) );
        if ( !sphereBelow.empty() ) HIP_TRY( rawSphereBelow.Upload( sphereBelow.data(), N ) );
        in.nodes = rawNodes.As< PtBvhNode >();       in.numNodes = N;
        in.escape = rawEscape.As< uint32_t >();
        in.prims = rawPrims.As< PtPrim >();          in.numPrims = P;
        in.triangles = rawTriangles.As< PtTriangle >();
        in.meshes = rawMeshes.As< PtMesh >();
        in.vertices = rawVertices.As< float >();     in.normals = rawNormals.As< float >();
        in.tangents = rawTangents.As< float >();     in.uvs = rawUvs.As< float >();
        in.spheres = rawSpheres.As< PtSphere >();
        in.sphereBelow = rawSphereBelow.As< uint8_t >();
        phase( "description to the device" );

        std::vector< uint32_t > topSorted( bfs.size() ), topRank( bfs.size() );
        for ( size_t k = 0; k < bfs.size(); ++k )
        {
            topSorted[k] = topByIndex[k].first;
            topRank[k]   = topByIndex[k].second;
        }
        HIP_TRY( dTopSorted.Upload( topSorted.data(), 4 * topSorted.size() ) );
        HIP_TRY( dTopRank.Upload( topRank.data(), 4 * topRank.size() ) );
        HIP_TRY( s->nodes.Allocate( 32ull * N ) );
        BuildReferenceNodes( nullptr, in, dTopSorted.As< uint32_t >(), dTopRank.As< uint32_t >(), static_cast< uint32_t >( bfs.size() ), s->nodes.As< float4 >() );
        HIP_TRY( hipGetLastError() );
    }
    else
    {
    std::unique_ptr< float[] > nodes( new float[8ull * N] ); // every word is written below
    ParallelFor( N, [&]( uint64_t begin, uint64_t end ) {
        for ( uint64_t i = begin; i < end; ++i )
        {
            const PtBvhNode& n = d->nodes[i];
            float* o = &nodes[8ull * place[i]];
            o[0] = n.aabb_min[0]; o[1] = n.aabb_min[1]; o[2] = n.aabb_max[0]; o[3] = n.aabb_max[1];
            o[4] = n.aabb_min[2]; o[5] = n.aabb_max[2];
            // link word: where to go after this node's subtree, top bit set for a leaf; second word: the leaf's
            // primitives, or the first child of an inner node
            uint32_t link = place[escape[i]], word = place[i + 1];
            if ( n.num_shapes > 0 )
            {
                link |= kLeafFlag;
                word = static_cast< uint32_t >( n.offset ) | ( static_cast< uint32_t >( n.num_shapes - 1 ) << 28 );
            }
            std::memcpy( &o[6], &link, 4 );
            std::memcpy( &o[7], &word, 4 );
        }
    } );
    HIP_TRY( s->nodes.Upload( nodes.get(), 32ull * N ) );
    }
    phase( "reference-order nodes" );

    // Second layout, for front-to-back traversal (kernels.hip, TraceClosestOrdered): 64-byte records, the root at 0,
    // record 1 unused, then sibling pairs at even indices - the top of the tree in the same order as above, below it
    // every pair followed by the first child's subtree, then the second child's.  Each record carries, per octant of
    // ray directions, the node that follows its subtree when every inner node's near child is visited first.
    uint32_t numOrdered = 0, topOrdered = 0;
    if ( !( d->flags & PT_SCENE_NO_ORDERED_LAYOUT ) && !hostLayouts )
    {
        // the inner nodes of the top whose children lie below it, in ascending order: each starts a run of pairs, one per
        // inner node of its subtree (a subtree of n nodes has ( n - 1 ) / 2 of them)
        std::vector< uint32_t > roots;
        for ( const auto& t : topByIndex )
            if ( d->nodes[t.first].num_shapes == 0 && !inTop( t.first + 1 ) ) roots.push_back( t.first );
        topOrdered = static_cast< uint32_t >( bfs.size() ) + 1u;
        std::vector< uint32_t > rootBase( roots.size() );
        uint32_t next = topOrdered;
        for ( size_t j = 0; j < roots.size(); ++j )
        {
            rootBase[j] = next;
            next += escape[roots[j]] - roots[j] - 1u; // two records per inner node
        }
        numOrdered = next; // N + 1
        DeviceBuffer dTopOrder, dRoots, dRootBase, dAt, dParent, dInnerPrefix, dBlockSums;
        HIP_TRY( dTopOrder.Upload( bfs.data(), 4 * bfs.size() ) );
        HIP_TRY( dRoots.Upload( roots.data(), 4 * roots.size() ) );
        HIP_TRY( dRootBase.Upload( rootBase.data(), 4 * rootBase.size() ) );
        HIP_TRY( dAt.Allocate( 4ull * N ) );
        HIP_TRY( dParent.Allocate( 4ull * N ) );
        HIP_TRY( dInnerPrefix.Allocate( 4ull * ( N + 1 ) ) );
        HIP_TRY( dBlockSums.Allocate( 4ull * ( LayoutScanBlocks( N ) + 1 ) ) );
        HIP_TRY( s->orderedNodes.Allocate( 64ull * numOrdered ) );
        BuildOrderedNodes( nullptr, in, dTopOrder.As< uint32_t >(), static_cast< uint32_t >( bfs.size() ), dRoots.As< uint32_t >(), dRootBase.As< uint32_t >(),
                           static_cast< uint32_t >( roots.size() ), numOrdered, dAt.As< uint32_t >(), dParent.As< uint32_t >(), dInnerPrefix.As< uint32_t >(),
                           dBlockSums.As< uint32_t >(), s->orderedNodes.As< uint4 >() );
        HIP_TRY( hipGetLastError() );
        HIP_TRY( hipDeviceSynchronize() );
    }
    else if ( !( d->flags & PT_SCENE_NO_ORDERED_LAYOUT ) )
    {
        const uint32_t unset = 0xFFFFFFFFu;
        std::vector< uint32_t > at( N, unset ); // reference index -> record
        at[0] = 0;
        uint32_t next = 2;
        for ( size_t i = 1; i + 1 < bfs.size(); i += 2 )
        {
            at[bfs[i]]     = next;
            at[bfs[i + 1]] = next + 1;
            next += 2;
        }
        topOrdered = next;
        std::vector< uint32_t > stack;
        for ( uint32_t i = 0; i < N; ++i )
        {
            if ( at[i] == unset || d->nodes[i].num_shapes > 0 || at[i + 1] != unset ) continue;
            stack.push_back( i );
            while ( !stack.empty() )
            {
                const uint32_t x = stack.back();
                stack.pop_back();
                if ( d->nodes[x].num_shapes > 0 ) continue;
                const uint32_t a = x + 1, b = static_cast< uint32_t >( d->nodes[x].offset );
                if ( at[a] == unset )
                {
                    at[a] = next;
                    at[b] = next + 1;
                    next += 2;
                }
                stack.push_back( b );
                stack.push_back( a );
            }
        }
        numOrdered = next; // N + 1
        // every word of a record is written below - the box and the two words by its own node, the links by its parent -
        // except the unused record 1
        std::unique_ptr< uint32_t[] > rec( new uint32_t[16ull * numOrdered] );
        std::memset( &rec[16], 0, 64 );
        for ( int o = 0; o < 8; ++o ) rec[8 + o] = numOrdered; // after the root's subtree: the end
        auto fill = [&]( uint32_t i ) {                        // a node's parent must have been filled before it
            const PtBvhNode& n = d->nodes[i];
            uint32_t* r = &rec[16ull * at[i]];
            std::memcpy( &r[0], &n.aabb_min[0], 4 ); std::memcpy( &r[1], &n.aabb_min[1], 4 );
            std::memcpy( &r[2], &n.aabb_max[0], 4 ); std::memcpy( &r[3], &n.aabb_max[1], 4 );
            std::memcpy( &r[4], &n.aabb_min[2], 4 ); std::memcpy( &r[5], &n.aabb_max[2], 4 );
            if ( n.num_shapes > 0 )
            {
                r[6] = static_cast< uint32_t >( n.offset ) | ( static_cast< uint32_t >( n.num_shapes - 1 ) << 28 );
                r[7] = kLeafFlag | ( !sphereBelow.empty() && sphereBelow[i] ? kSphereBelowFlag : 0u );
                return;
            }
            const uint32_t a = i + 1, b = static_cast< uint32_t >( n.offset );
            const uint32_t axis = n.axis < 3 ? n.axis : 0; // any choice of the near child gives the same hits
            r[6] = at[a];
            r[7] = ( 2u + axis ) | ( !sphereBelow.empty() && sphereBelow[i] ? kSphereBelowFlag : 0u );
            uint32_t* ra = &rec[16ull * at[a]];
            uint32_t* rb = &rec[16ull * at[b]];
            for ( uint32_t o = 0; o < 8; ++o )
            {
                const bool secondIsNear = ( o >> axis ) & 1u;
                ( secondIsNear ? rb : ra )[8 + o] = secondIsNear ? at[a] : at[b]; // after the near child's subtree: its sibling
                ( secondIsNear ? ra : rb )[8 + o] = r[8 + o];                      // after the far child's: what follows the parent
            }
        };
        // top of the tree first (a node's links derive from its parent's), then the subtrees below it side by side
        const TreeCut cut = CutTree( d, escape );
        for ( uint32_t x : cut.upper ) fill( x );
        ForEachSubtree( cut, N, [&]( uint32_t root ) {
            for ( uint32_t i = root, end = escape[root]; i < end; ++i ) fill( i );
        } );
        HIP_TRY( s->orderedNodes.Upload( rec.get(), 64ull * numOrdered ) );
    }
    phase( "front-to-back nodes" );

    // Third layout, for the shadow rays of a scene whose one light is directional (kernels.hip, ConnectParallelKernel):
    // every such ray has the light's direction, so seen along it a node is a rectangle, and the ray can only enter the
    // node if its point lies in the rectangle and the node reaches beyond the ray's start.  16-byte records
    // {x0 | width << 16, y0 | height << 16, far, word}: the rectangle on a 16-bit grid over the root's, widened by what
    // the rounding of the ray's own projection can amount to; far = the box's far plane, in the rays' sense, along the
    // world axis they mostly travel on.  The records are outer bounds only: a hit
    // found through them counts once the reference's own test of its leaf's box has passed (the box travels in the
    // primitive record).  Order: the first numShadowTop records are the top of the tree (as in the other layouts) and
    // are read from LDS with a second word each; the rest follows in the reference's pre-order, where an inner
    // node's first child is the next record and a leaf is followed by the next record too - so one word serves:
    // word = link for an inner node (where to go when its box is missed), leaf flag | leaf word for a leaf.  A
    // leaf of the lower part whose successor lies in the top part gets a record of its own behind it that no ray enters
    // (wmax = -inf) and that links there.  Top records: inner {link, first child}, leaf {flag | leaf word, link}.
    uint32_t numShadow = 0, shadowTop = 0;
    ShadowAxes lightAxes{};  // kept for the wide tree in the light's frame, which is made once the wide layout is there
    ShadowGrid lightGrid{};
    bool lightGridMade = false;
    if ( !( d->flags & PT_SCENE_NO_PARALLEL_SHADOWS ) && d->num_lights == 1 && d->lights[0].type == PT_LIGHT_DIRECTIONAL )
    {
        const double w[3] = { -static_cast< double >( d->lights[0].direction[0] ), -static_cast< double >( d->lights[0].direction[1] ),
                              -static_cast< double >( d->lights[0].direction[2] ) }; // wi of src/lights.cpp:30
        const double wLen = std::sqrt( w[0] * w[0] + w[1] * w[1] + w[2] * w[2] );
        float fu[3] = { 0, 0, 0 }, fv[3] = { 0, 0, 0 };
        const int zeros = ( w[0] == 0 ) + ( w[1] == 0 ) + ( w[2] == 0 );
        bool build = std::isfinite( wLen ) && wLen > 0;
        int depthAxis = 0; // the world axis the rays mostly travel along
        for ( int a = 1; a < 3; ++a )
            if ( std::fabs( w[a] ) > std::fabs( w[depthAxis] ) ) depthAxis = a;
        // a record's word keeps bit 31 for the leaf flag, which leaves three bits for a leaf's shape count
        // (the kernels that make the bounds look for such a leaf themselves)
        for ( uint32_t i = 0; i < N && build && hostLayouts; ++i ) build = d->nodes[i].num_shapes <= 8;
        if ( build && zeros == 2 )
        {
            // along an axis: the other two axes as they are, and the projections are exact
            const int k = w[0] != 0 ? 0 : w[1] != 0 ? 1 : 2;
            fu[( k + 1 ) % 3] = 1.0f;
            fv[( k + 2 ) % 3] = 1.0f;
        }
        else if ( build )
        {
            int k = 0; // the axis the direction leans on least
            for ( int a = 1; a < 3; ++a )
                if ( std::fabs( w[a] ) < std::fabs( w[k] ) ) k = a;
            double e[3] = { 0, 0, 0 };
            e[k] = 1;
            double u[3] = { w[1] * e[2] - w[2] * e[1], w[2] * e[0] - w[0] * e[2], w[0] * e[1] - w[1] * e[0] };
            const double uLen = std::sqrt( u[0] * u[0] + u[1] * u[1] + u[2] * u[2] );
            for ( int a = 0; a < 3; ++a ) u[a] /= uLen;
            double v[3] = { w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0] };
            for ( int a = 0; a < 3; ++a )
            {
                fu[a] = static_cast< float >( u[a] );
                fv[a] = static_cast< float >( v[a] / wLen );
            }
        }
        // what the device's single-precision projection of a point of the scene can be off by
        double maxAbs = 0;
        for ( int a = 0; a < 3; ++a ) maxAbs = std::max( { maxAbs, std::fabs( static_cast< double >( d->nodes[0].aabb_min[a] ) ), std::fabs( static_cast< double >( d->nodes[0].aabb_max[a] ) ) } );
        const double projectionError = zeros == 2 ? 0.0 : std::ldexp( maxAbs, -20 );
        // Bounds of every node in the light's frame, from the geometry below it rather than from its box (the rectangle
        // around a box seen at an angle is far larger than the one around its triangles): leaves from their primitives,
        // inner nodes from their children, children before parents.  {u0, u1, v0, v1} per node, rounded outwards.
        std::unique_ptr< float[] > frame( build && hostLayouts ? new float[4ull * N] : nullptr );
        DeviceBuffer dFrame;
        float rootFrame[4] = { 0, 0, 0, 0 };
        auto down = []( double x ) { float f = static_cast< float >( x ); return static_cast< double >( f ) > x ? std::nextafter( f, -std::numeric_limits< float >::infinity() ) : f; };
        auto up   = []( double x ) { float f = static_cast< float >( x ); return static_cast< double >( f ) < x ? std::nextafter( f, std::numeric_limits< float >::infinity() ) : f; };
        auto bound = [&]( uint32_t i ) {
            const PtBvhNode& n = d->nodes[i];
            float* f = &frame[4ull * i];
            if ( n.num_shapes == 0 )
            {
                const float *x = &frame[4ull * ( i + 1 )], *y = &frame[4ull * static_cast< uint32_t >( n.offset )];
                f[0] = std::min( x[0], y[0] ); f[1] = std::max( x[1], y[1] );
                f[2] = std::min( x[2], y[2] ); f[3] = std::max( x[3], y[3] );
                return;
            }
            const double inf = std::numeric_limits< double >::infinity();
            double lo[2] = { inf, inf }, hi[2] = { -inf, -inf };
            const float* axes[2] = { fu, fv };
            for ( uint32_t k = 0; k < n.num_shapes; ++k )
            {
                const PtPrim& prim = d->prims[static_cast< uint64_t >( n.offset ) + k];
                if ( prim.type == PT_PRIM_TRIANGLE )
                {
                    const PtTriangle& t = d->triangles[prim.index];
                    const uint32_t idx[3] = { t.i0, t.i1, t.i2 };
                    for ( int c = 0; c < 3; ++c )
                    {
                        const float* p = TriVertex( d, d->vertices, t, idx[c], 3 );
                        for ( int a = 0; a < 2; ++a )
                        {
                            const double q = static_cast< double >( p[0] ) * axes[a][0] + static_cast< double >( p[1] ) * axes[a][1] + static_cast< double >( p[2] ) * axes[a][2];
                            lo[a] = std::min( lo[a], q );
                            hi[a] = std::max( hi[a], q );
                        }
                    }
                }
                else
                {
                    const PtSphere& sp = d->spheres[prim.index];
                    for ( int a = 0; a < 2; ++a )
                    {
                        const double q = static_cast< double >( sp.position[0] ) * axes[a][0] + static_cast< double >( sp.position[1] ) * axes[a][1] + static_cast< double >( sp.position[2] ) * axes[a][2];
                        const double len = std::sqrt( static_cast< double >( axes[a][0] ) * axes[a][0] + static_cast< double >( axes[a][1] ) * axes[a][1] + static_cast< double >( axes[a][2] ) * axes[a][2] );
                        const double r = std::fabs( static_cast< double >( sp.radius ) ) * len * ( 1.0 + 1e-6 );
                        lo[a] = std::min( lo[a], q - r );
                        hi[a] = std::max( hi[a], q + r );
                    }
                }
            }
            f[0] = down( lo[0] ); f[1] = up( hi[0] );
            f[2] = down( lo[1] ); f[3] = up( hi[1] );
        };
        if ( build && !hostLayouts )
        {
            DeviceBuffer dParent, dArrivals;
            ShadowAxes axes;
            for ( int a = 0; a < 3; ++a )
            {
                axes.u[a] = fu[a];
                axes.v[a] = fv[a];
            }
            DeviceBuffer dVerdict;
            HIP_TRY( dFrame.Allocate( 16ull * N ) );
            HIP_TRY( dParent.Allocate( 4ull * N ) );
            HIP_TRY( dArrivals.Allocate( 4ull * N ) );
            HIP_TRY( dVerdict.Allocate( 4 ) );
            // the fast climb first; should its check ever fail, once more with full fences (PTAMD_SHADOW_FULL_FENCES: at once)
            uint32_t verdict = kShadowFramesWrong;
            for ( int attempt = std::getenv( "PTAMD_SHADOW_FULL_FENCES" ) ? 1 : 0; attempt < 2 && ( verdict & kShadowFramesWrong ); ++attempt )
            {
                BuildShadowFrames( nullptr, in, axes, attempt == 1, dParent.As< uint32_t >(), dArrivals.As< uint32_t >(), dFrame.As< float >(), dVerdict.As< uint32_t >() );
                HIP_TRY( hipGetLastError() );
                HIP_TRY( hipMemcpy( &verdict, dVerdict.ptr, 4, hipMemcpyDeviceToHost ) );
            }
            if ( verdict & kShadowFramesWrong ) return Fail( PT_ERR_DEVICE, "pt_upload_scene: the bounds in the light's frame failed their check" );
            if ( verdict & kShadowLeafTooLarge ) build = false;
            HIP_TRY( hipMemcpy( rootFrame, dFrame.ptr, 16, hipMemcpyDeviceToHost ) );
        }
        else if ( build )
        {
            const TreeCut cut = CutTree( d, escape );
            ForEachSubtree( cut, N, [&]( uint32_t root ) {
                for ( uint32_t i = escape[root]; i-- > root; ) bound( i );
            } );
            for ( size_t k = cut.upper.size(); k-- > 0; ) bound( cut.upper[k] );
            std::memcpy( rootFrame, frame.get(), 16 );
        }
        double origin[2] = { 0, 0 }, cell[2] = { 1, 1 };
        float invCell[2] = { 1, 1 }, originF[2] = { 0, 0 };
        double slack[2] = { 1, 1 }; // in cells
        for ( int a = 0; a < 2 && build; ++a )
        {
            const double lo = rootFrame[2 * a], hi = rootFrame[2 * a + 1];
            build = std::isfinite( lo ) && std::isfinite( hi );
            if ( !build ) break;
            cell[a]    = std::max( { ( hi - lo ) / 64000.0, std::ldexp( maxAbs, -18 ), 1e-30 } );
            invCell[a] = static_cast< float >( 1.0 / cell[a] );
            cell[a]    = 1.0 / static_cast< double >( invCell[a] ); // the cell the device's multiplication implies
            // The device computes floor( ( p - origin ) * invCell ) in single precision.  Its p is off by projectionError, the
            // subtraction by an ulp of a number of the scene's size (at most 1 / 8 of a cell, see the cell's floor above),
            // the product by an ulp of a number below 2^16; and a ray that the reference's test counts as a hit may pass
            // a rounding error outside the triangle.  A cell q then covers every point the device maps to q if the
            // rectangle [lo, hi] is widened by `slack` cells before it is rounded to cells.
            slack[a]   = ( projectionError + std::ldexp( maxAbs, -21 ) ) / cell[a] + 0.1;
            const int border = 2 + static_cast< int >( std::ceil( slack[a] ) );
            originF[a] = static_cast< float >( lo - border * cell[a] );
            origin[a]  = originF[a];
            build      = std::isfinite( originF[a] ) && std::isfinite( invCell[a] ) && ( hi - origin[a] ) / cell[a] + border < 65534.0;
        }
        if ( build ) shadowTop = static_cast< uint32_t >( std::min< size_t >( bfs.size(), kShadowTopNodes ) );
        if ( build && !hostLayouts )
        {
            DeviceBuffer dBefore, dBlockSums;
            HIP_TRY( dBefore.Allocate( 4ull * ( N + 1 ) ) );
            HIP_TRY( dBlockSums.Allocate( 4ull * ( LayoutScanBlocks( N ) + 1 ) ) );
            CountShadowRecords( nullptr, in, dTopSorted.As< uint32_t >(), dTopRank.As< uint32_t >(), static_cast< uint32_t >( bfs.size() ), shadowTop,
                                dBefore.As< uint32_t >(), dBlockSums.As< uint32_t >() );
            HIP_TRY( hipGetLastError() );
            uint32_t behindTheTop = 0;
            HIP_TRY( hipMemcpy( &behindTheTop, dBefore.As< uint32_t >() + N, 4, hipMemcpyDeviceToHost ) );
            numShadow = shadowTop + behindTheTop;
            ShadowGrid grid;
            for ( int a = 0; a < 2; ++a )
            {
                grid.origin[a] = origin[a];
                grid.cell[a]   = cell[a];
                grid.slack[a]  = slack[a];
            }
            grid.depthAxis       = static_cast< uint32_t >( depthAxis );
            grid.towardsPositive = w[depthAxis] > 0;
            grid.shadowTop       = shadowTop;
            grid.numRecords      = numShadow;
            lightGrid            = grid;
            lightGridMade        = true;
            for ( int a = 0; a < 3; ++a )
            {
                lightAxes.u[a] = fu[a];
                lightAxes.v[a] = fv[a];
            }
            HIP_TRY( s->shadowNodes.Allocate( 16ull * numShadow ) );
            HIP_TRY( s->shadowTopWords.Allocate( 4ull * shadowTop ) );
            BuildShadowRecords( nullptr, in, dFrame.As< float >(), grid, dTopSorted.As< uint32_t >(), dTopRank.As< uint32_t >(), static_cast< uint32_t >( bfs.size() ),
                                dBefore.As< uint32_t >(), s->shadowNodes.As< uint4 >(), s->shadowTopWords.As< uint32_t >() );
            HIP_TRY( hipGetLastError() );
            HIP_TRY( hipDeviceSynchronize() ); // the scratch buffers go out of scope here
        }
        else if ( build )
        {
            const uint32_t unset = 0xFFFFFFFFu;
            std::vector< uint32_t > sp( N + 1, unset ); // reference index -> record
            for ( uint32_t k = 0; k < shadowTop; ++k ) sp[bfs[k]] = k;
            std::vector< std::pair< uint32_t, uint32_t > > jumps; // (record, reference index of its target)
            uint32_t next = shadowTop;
            for ( uint32_t i = 0; i < N; ++i )
            {
                if ( sp[i] != unset ) continue;
                sp[i] = next++;
                if ( d->nodes[i].num_shapes > 0 && escape[i] < N && sp[escape[i]] != unset && sp[escape[i]] < shadowTop ) jumps.push_back( { next++, escape[i] } );
            }
            numShadow = next;
            sp[N]     = numShadow; // the end of the traversal
            std::unique_ptr< float[] > rec( new float[4ull * numShadow] );
            std::vector< uint32_t > topWords( shadowTop, 0u );
            ParallelFor( N, [&]( uint64_t begin, uint64_t end ) {
                for ( uint64_t i = begin; i < end; ++i )
                {
                    const PtBvhNode& n = d->nodes[i];
                    uint32_t q[2];
                    const float* f = &frame[4ull * i];
                    for ( int a = 0; a < 2; ++a )
                    {
                        const double lo = f[2 * a], hi = f[2 * a + 1];
                        const double qlo = std::min( 65534.0, std::max( 1.0, std::floor( ( lo - origin[a] ) / cell[a] - slack[a] ) ) );
                        const double qhi = std::min( 65534.0, std::max( qlo, std::floor( ( hi - origin[a] ) / cell[a] + slack[a] ) ) );
                        q[a] = static_cast< uint32_t >( qlo ) | static_cast< uint32_t >( qhi - qlo ) << 16;
                    }
                    // how far the box reaches in the rays' sense along the world axis they mostly travel on: its far
                    // plane there, which is the reference's own "the box lies ahead of the ray" test on that axis, and exact
                    const float wmax = w[depthAxis] > 0 ? n.aabb_max[depthAxis] : -n.aabb_min[depthAxis];
                    const bool leaf     = n.num_shapes > 0;
                    const uint32_t link = sp[escape[i]];
                    const uint32_t word = leaf ? kLeafFlag | static_cast< uint32_t >( n.offset ) | ( static_cast< uint32_t >( n.num_shapes - 1 ) << 28 ) : link;
                    float* r = &rec[4ull * sp[i]];
                    std::memcpy( &r[0], &q[0], 4 );
                    std::memcpy( &r[1], &q[1], 4 );
                    r[2] = wmax;
                    std::memcpy( &r[3], &word, 4 );
                    if ( sp[i] < shadowTop ) topWords[sp[i]] = leaf ? link : sp[i + 1];
                }
            } );
            for ( const auto& j : jumps )
            {
                float* r = &rec[4ull * j.first];
                const uint32_t one = 1u, link = sp[j.second];
                std::memcpy( &r[0], &one, 4 );
                std::memcpy( &r[1], &one, 4 );
                r[2] = -std::numeric_limits< float >::infinity();
                std::memcpy( &r[3], &link, 4 );
            }
            HIP_TRY( s->shadowNodes.Upload( rec.get(), 16ull * numShadow ) );
            HIP_TRY( s->shadowTopWords.Upload( topWords.data(), 4ull * shadowTop ) );
        }
        if ( build )
        {
            for ( int a = 0; a < 3; ++a )
            {
                s->view.lightU[a] = fu[a];
                s->view.lightV[a] = fv[a];
            }
            for ( int a = 0; a < 2; ++a )
            {
                s->view.shadowOrigin[a]  = originF[a];
                s->view.shadowInvCell[a] = invCell[a];
            }
            s->view.shadowDepthAxis = static_cast< uint32_t >( depthAxis );
            s->view.shadowDepthSign = w[depthAxis] < 0 ? -1.0f : 1.0f;
        }
    }

    phase( "directional-shadow nodes" );
    // 100 bytes of attributes in a 128-byte slot: two 64-byte lines per record
    std::unique_ptr< float[] > prims, shade;
    if ( !hostLayouts )
    {
        HIP_TRY( s->prims.Allocate( 64ull * P ) );
        HIP_TRY( s->triShade.Allocate( 128ull * P ) );
        BuildPrimRecords( nullptr, in, s->prims.As< float4 >(), s->triShade.As< float4 >() );
        HIP_TRY( hipGetLastError() );
    }
    else
    {
    prims.reset( new float[16ull * P] );
    shade.reset( new float[32ull * P] );
    ParallelFor( P, [&]( uint64_t begin, uint64_t end ) {
    for ( uint64_t i = begin; i < end; ++i )
    {
        float* q = &prims[16 * i];
        std::memset( q, 0, 64 );
        std::memset( &shade[32 * i], 0, 128 );
        if ( d->prims[i].type == PT_PRIM_TRIANGLE )
        {
            const PtTriangle& t = d->triangles[d->prims[i].index];
            f3h v0 = Load3( TriVertex( d, d->vertices, t, t.i0, 3 ) );
            f3h e1 = Load3( TriVertex( d, d->vertices, t, t.i1, 3 ) ) - v0; // v0v1 of src/intersection_tests.cpp:37
            f3h e2 = Load3( TriVertex( d, d->vertices, t, t.i2, 3 ) ) - v0; // v0v2 of src/intersection_tests.cpp:38
            q[0] = v0.x; q[1] = v0.y; q[2] = v0.z;
            uint32_t tag = kTagTriangle;
            std::memcpy( &q[3], &tag, 4 );
            q[4] = e1.x; q[5] = e1.y; q[6] = e1.z;
            q[8] = e2.x; q[9] = e2.y; q[10] = e2.z;
            float* a = &shade[32 * i];
            const uint32_t idx[3] = { t.i0, t.i1, t.i2 };
            for ( int k = 0; k < 3; ++k )
            {
                std::memcpy( a + 3 * k, TriVertex( d, d->normals, t, idx[k], 3 ), 12 );
                std::memcpy( a + 9 + 3 * k, TriVertex( d, d->tangents, t, idx[k], 3 ), 12 );
                std::memcpy( a + 18 + 2 * k, TriVertex( d, d->uvs, t, idx[k], 2 ), 8 );
            }
            int32_t material = d->meshes[t.mesh].material;
            std::memcpy( a + 24, &material, 4 );
        }
        else
        {
            const PtSphere& sp = d->spheres[d->prims[i].index];
            q[0] = sp.position[0]; q[1] = sp.position[1]; q[2] = sp.position[2];
            uint32_t tag = kTagSphere, index = d->prims[i].index;
            std::memcpy( &q[3], &tag, 4 );
            q[4] = sp.radius;
            std::memcpy( &q[5], &index, 4 );
        }
    }
    } );
    // every primitive record carries the box of its leaf in the words the tests above leave free (7, 11, 12 - 15)
    ParallelFor( N, [&]( uint64_t begin, uint64_t end ) {
        for ( uint64_t i = begin; i < end; ++i )
        {
            const PtBvhNode& n = d->nodes[i];
            for ( uint32_t k = 0; k < n.num_shapes; ++k )
            {
                float* q = &prims[16ull * ( static_cast< uint64_t >( n.offset ) + k )];
                q[7] = n.aabb_min[0]; q[11] = n.aabb_min[1]; q[12] = n.aabb_min[2];
                q[13] = n.aabb_max[0]; q[14] = n.aabb_max[1]; q[15] = n.aabb_max[2];
            }
        }
    } );
    HIP_TRY( s->prims.Upload( prims.get(), 64ull * P ) );
    HIP_TRY( s->triShade.Upload( shade.get(), 128ull * P ) );
    shade.reset();
    }
    phase( "primitive records" );

    // Fourth layout (wide_bvh.h): the tree collapsed to eight children per node, and the primitive records once more
    // in that tree's order with the reference's index in place of the tag.  A tree it cannot express (a bound that is
    // not finite, a box outside its parent's) simply has none.
    uint32_t numWide = 0, wideDepth = 0;
    if ( !( d->flags & PT_SCENE_NO_WIDE_LAYOUT ) )
    {
        WideLayout wide;
        WideDeviceLayout onDevice;
        // below the top of the tree the collapse runs in kernels too; if that cannot be (device memory), on host threads
        if ( !hostLayouts && BuildWideLayoutOnDevice( d, escape, sphereBelow.empty() ? nullptr : sphereBelow.data(), in.nodes, in.prims, in.sphereBelow, onDevice ) )
        {
            numWide   = onDevice.numNodes;
            wideDepth = onDevice.depth;
            s->wideNodes.Release();
            s->wideNodes.ptr   = onDevice.nodes;
            s->wideNodes.bytes = onDevice.nodeBytes;
            DeviceBuffer dOrder;
            dOrder.ptr   = onDevice.primOrder;
            dOrder.bytes = 4ull * P;
            HIP_TRY( s->widePrims.Allocate( 64ull * P ) );
            GatherWidePrims( nullptr, in, s->prims.As< float4 >(), dOrder.As< uint32_t >(), s->widePrims.As< float4 >() );
            HIP_TRY( hipGetLastError() );
            HIP_TRY( hipDeviceSynchronize() );
            const uint64_t lanes = std::max< uint64_t >( static_cast< uint64_t >( s->wideBlocks ) * s->wideThreads, static_cast< uint64_t >( s->wideShadowBlocks ) * s->wideShadowThreads );
            if ( wideDepth > static_cast< uint32_t >( s->wideLdsStack ) ) HIP_TRY( s->wideSpill.Allocate( 8ull * ( wideDepth - s->wideLdsStack ) * lanes ) );
        }
        else if ( BuildWideLayout( d, escape, sphereBelow.empty() ? nullptr : sphereBelow.data(), wide ) )
        {
            numWide   = static_cast< uint32_t >( wide.nodes.size() / kWideNodeWords );
            wideDepth = wide.depth;
            HIP_TRY( s->wideNodes.Upload( wide.nodes.data(), wide.nodes.size() * 4 ) );
            std::vector< uint32_t >().swap( wide.nodes );
            if ( !hostLayouts )
            {
                DeviceBuffer dOrder;
                HIP_TRY( dOrder.Upload( wide.primOrder.data(), 4ull * P ) );
                HIP_TRY( s->widePrims.Allocate( 64ull * P ) );
                GatherWidePrims( nullptr, in, s->prims.As< float4 >(), dOrder.As< uint32_t >(), s->widePrims.As< float4 >() );
                HIP_TRY( hipGetLastError() );
                HIP_TRY( hipDeviceSynchronize() );
            }
            else
            {
            std::unique_ptr< float[] > ordered( new float[16ull * P] );
            ParallelFor( P, [&]( uint64_t begin, uint64_t end ) {
                for ( uint64_t k = begin; k < end; ++k )
                {
                    const uint32_t i = wide.primOrder[k];
                    float* q = &ordered[16 * k];
                    std::memcpy( q, &prims[16ull * i], 64 );
                    const uint32_t tag = i | ( d->prims[i].type == PT_PRIM_SPHERE ? 0x80000000u : 0u );
                    std::memcpy( &q[3], &tag, 4 );
                }
            } );
            HIP_TRY( s->widePrims.Upload( ordered.get(), 64ull * P ) );
            }
            // stack entries beyond the ones the kernels keep in LDS: one slot per level and thread of the larger of the two grids
            const uint64_t lanes = std::max< uint64_t >( static_cast< uint64_t >( s->wideBlocks ) * s->wideThreads, static_cast< uint64_t >( s->wideShadowBlocks ) * s->wideShadowThreads );
            if ( wideDepth > static_cast< uint32_t >( s->wideLdsStack ) ) HIP_TRY( s->wideSpill.Allocate( 8ull * ( wideDepth - s->wideLdsStack ) * lanes ) );
        }
    }
    prims.reset();
    phase( "wide nodes and primitives" );

    // The wide nodes once more in the frame of the one directional light (ConnectLightWideKernel): made by kernels from the
    // wide layout and the grid of the third layout.  A scene uploaded with PT_SCENE_HOST_LAYOUTS has none (its shadow rays
    // take ConnectWideKernel), nor has one whose bounds fail their check.
    if ( lightGridMade && numWide != 0 && s->widePrims.ptr )
    {
        DeviceBuffer primRect, primFar, parent, arrivals, nodeBounds, verdictWord;
        HIP_TRY( primRect.Allocate( 8ull * P ) );
        HIP_TRY( primFar.Allocate( 4ull * P ) );
        HIP_TRY( parent.Allocate( 4ull * numWide ) );
        HIP_TRY( arrivals.Allocate( 4ull * numWide ) );
        HIP_TRY( nodeBounds.Allocate( 16ull * numWide ) );
        HIP_TRY( verdictWord.Allocate( 4 ) );
        HIP_TRY( s->lightWideNodes.Allocate( 64ull * numWide ) );
        BuildLightWideNodes( nullptr, in, lightAxes, lightGrid, s->wideNodes.ptr, numWide, s->widePrims.As< float4 >(), primRect.As< uint2 >(), primFar.As< float >(),
                             parent.As< uint32_t >(), arrivals.As< uint32_t >(), nodeBounds.As< uint32_t >(), verdictWord.As< uint32_t >(), s->lightWideNodes.As< uint4 >() );
        HIP_TRY( hipGetLastError() );
        uint32_t verdict = 0;
        HIP_TRY( hipMemcpy( &verdict, verdictWord.ptr, 4, hipMemcpyDeviceToHost ) );
        if ( verdict != 0 ) s->lightWideNodes.Release();
        phase( "wide nodes in the light's frame" );
    }

    std::vector< float > spheres( 24ull * d->num_spheres, 0.0f );
    for ( uint64_t i = 0; i < d->num_spheres; ++i )
    {
        const PtSphere& sp = d->spheres[i];
        float* o = &spheres[24 * i];
        std::memcpy( o, sp.world_to_local, 64 );
        o[16] = sp.position[0]; o[17] = sp.position[1]; o[18] = sp.position[2]; o[19] = sp.radius;
        std::memcpy( &o[20], &sp.material, 4 );
    }
    HIP_TRY( s->spheres.Upload( spheres.data(), spheres.size() * 4 ) );

    std::vector< float > materials( 8ull * d->num_materials, 0.0f );
    for ( uint64_t i = 0; i < d->num_materials; ++i )
    {
        const PtMaterial& m = d->materials[i];
        float* o = &materials[8 * i];
        o[0] = m.albedo[0]; o[1] = m.albedo[1]; o[2] = m.albedo[2];
        std::memcpy( &o[3], &m.albedo_texture, 4 );
        o[4] = m.Ke[0]; o[5] = m.Ke[1]; o[6] = m.Ke[2];
    }
    HIP_TRY( s->materials.Upload( materials.data(), materials.size() * 4 ) );

    std::vector< DevTexture > textures( d->num_textures );
    for ( uint64_t i = 0; i < d->num_textures; ++i )
        textures[i] = DevTexture{ d->textures[i].width, d->textures[i].height, static_cast< uint32_t >( d->textures[i].texel_offset ),
                                  static_cast< uint32_t >( d->textures[i].texel_offset >> 32 ) };
    HIP_TRY( s->textures.Upload( textures.data(), textures.size() * sizeof( DevTexture ) ) );
    HIP_TRY( s->texels.Upload( d->texels, d->num_texels * 4 ) );

    std::vector< DevLight > lights( d->num_lights );
    int slots = 0;
    uint32_t areaSamples = 0;
    for ( uint64_t i = 0; i < d->num_lights; ++i )
    {
        const PtLight& l = d->lights[i];
        DevLight o;
        std::memset( &o, 0, sizeof( o ) );
        o.type     = l.type;
        o.nSamples = l.n_samples;
        std::memcpy( o.Lemit, l.Lemit, 12 );
        if ( l.type == PT_LIGHT_POINT ) std::memcpy( o.a, l.position, 12 );
        if ( l.type == PT_LIGHT_DIRECTIONAL ) std::memcpy( o.a, l.direction, 12 );
        if ( l.type == PT_LIGHT_AREA )
        {
            o.drawOffset = areaSamples;
            areaSamples += static_cast< uint32_t >( l.n_samples );
            const PtPrim& p = d->prims[l.prim];
            o.shapeType = p.type;
            if ( p.type == PT_PRIM_TRIANGLE )
            {
                const PtTriangle& t = d->triangles[p.index];
                const uint32_t idx[3] = { t.i0, t.i1, t.i2 };
                float* vs[3] = { o.v0, o.v1, o.v2 };
                float* ns[3] = { o.n0, o.n1, o.n2 };
                for ( int k = 0; k < 3; ++k )
                {
                    std::memcpy( vs[k], TriVertex( d, d->vertices, t, idx[k], 3 ), 12 );
                    std::memcpy( ns[k], TriVertex( d, d->normals, t, idx[k], 3 ), 12 );
                }
                // Triangle::Area (src/shapes.cpp:96-102) and the area pdf 1 / Area (src/shapes.cpp:113)
                f3h c      = Cross( Load3( o.v1 ) - Load3( o.v0 ), Load3( o.v2 ) - Load3( o.v0 ) );
                float area = 0.5f * std::sqrt( Dot( c, c ) );
                o.invArea  = 1.0f / area;
            }
            else
            {
                const PtSphere& sp = d->spheres[p.index];
                std::memcpy( o.a, sp.position, 12 );
                o.radius   = sp.radius;
                float area = 4 * kPiHost * sp.radius * sp.radius; // Sphere::Area (src/shapes.cpp:37-40)
                o.invArea  = 1.0f / area;
            }
        }
        slots += l.n_samples;
        lights[i] = o;
    }
    HIP_TRY( s->lights.Upload( lights.data(), lights.size() * sizeof( DevLight ) ) );
    if ( d->skybox ) HIP_TRY( s->skybox.Upload( d->skybox, 6ull * d->skybox_width * d->skybox_height * 16 ) );

    SceneView& v    = s->view;
    v.nodes         = s->nodes.As< float4 >();
    v.orderedNodes  = s->orderedNodes.As< float4 >();
    v.shadowNodes   = s->shadowNodes.As< float4 >();
    v.shadowTopWords = s->shadowTopWords.As< uint32_t >();
    v.numShadowNodes = numShadow;
    v.numShadowTop  = shadowTop;
    v.numOrderedNodes = numOrdered;
    v.numTopOrdered = topOrdered;
    v.wideNodes     = s->wideNodes.As< float4 >();
    v.widePrims     = s->widePrims.As< float4 >();
    v.numWideNodes  = numWide;
    v.lightWideNodes = s->lightWideNodes.As< float4 >();
    v.wideDepth     = wideDepth;
    v.prims         = s->prims.As< float4 >();
    v.triShade      = s->triShade.As< float4 >();
    v.spheres       = s->spheres.As< float4 >();
    v.materials     = s->materials.As< float4 >();
    v.textures      = s->textures.As< DevTexture >();
    v.texels        = s->texels.As< uint32_t >();
    v.lights        = s->lights.As< DevLight >();
    v.skybox        = s->skybox.As< float4 >();
    v.numNodes      = N;
    v.numTopNodes   = s->topNodes;
    v.numPrims      = static_cast< uint32_t >( P );
    v.numSpheres    = static_cast< uint32_t >( d->num_spheres );
    v.numLights     = static_cast< uint32_t >( d->num_lights );
    v.shadowSlots   = slots;
    s->directSum    = d->num_lights == 1 && d->lights[0].n_samples == 1 && !( d->flags & PT_SCENE_NO_DIRECT_SUM );
    s->oneDirectionalLight = d->num_lights == 1 && d->lights[0].type == PT_LIGHT_DIRECTIONAL;
    for ( int a = 0; a < 3 && s->oneDirectionalLight; ++a ) s->towardsLight[a] = -d->lights[0].direction[a];
    v.drawsPerBounce = static_cast< int32_t >( 2 * areaSamples + 2 );
    v.skyboxWidth   = d->skybox_width;
    v.skyboxHeight  = d->skybox_height;
    std::memcpy( v.background, d->background_radiance, 12 );

    s->camera   = d->camera;
    s->width    = d->image_width;
    s->height   = d->image_height;
    s->maxDepth = d->max_depth;
    // ray origin, direction and throughput double buffered and the shadow origin (12 B each), hit, radiance (16 B each),
    // four queues (4 B each), per shadow slot a 32 B record and a 4 B queue entry
    s->bytesPerPathFixed = 7 * 12 + 2 * 16 + 5 * 4 + 36ull * static_cast< uint64_t >( slots > 0 ? slots : 1 );

    const size_t pixels = static_cast< size_t >( s->width ) * s->height;
    HIP_TRY( s->film.Allocate( pixels * 16 ) );
    HIP_TRY( hipMemset( s->film.ptr, 0, pixels * 16 ) );
    HIP_TRY( s->stats.Allocate( kStatsWords * 8 ) );
    HIP_TRY( hipMemset( s->stats.ptr, 0, kStatsWords * 8 ) );
    // per bounce: the two unit cursors of the traversal kernels and the length of the redo queue
    HIP_TRY( s->cursors.Allocate( 3 * 4 * static_cast< size_t >( s->maxDepth > 0 ? s->maxDepth : 1 ) ) );
    HIP_TRY( hipDeviceSynchronize() ); // the kernels that read the description's device copy are done before it is freed
    phase( "materials, textures, film" );
    return PT_OK;
}

int EnsurePathBuffers( PtScene* s, uint64_t capacity )
{
    if ( s->paths.capacity >= capacity ) return PT_OK;
    // Growing frees every buffer before it allocates the larger one.  Should an allocation below fail, the scene is
    // left with no path buffers at all (capacity 0), so that the next call allocates from scratch instead of
    // launching on freed memory.
    s->paths = PathBuffers{};
    const uint64_t slots = s->view.shadowSlots > 0 ? s->view.shadowSlots : 1;
    // one queue segment of kUnitPaths entries per unit of paths, one length per bounce, queue and unit
    const uint64_t units   = ( capacity + kUnitPaths - 1 ) / kUnitPaths;
    const uint64_t entries = units * kUnitPaths;
    HIP_TRY( s->rayO0.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->rayO1.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->rayD0.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->rayD1.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->thr0.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->thr1.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->hit.Allocate( entries * 16 ) );
    HIP_TRY( s->rad.Allocate( capacity * 16 ) );
    HIP_TRY( s->shadowO.Allocate( entries * sizeof( Packed3 ) ) );
    HIP_TRY( s->shDir.Allocate( entries * slots * 16 ) );
    HIP_TRY( s->shEst.Allocate( entries * slots * 16 ) );
    HIP_TRY( s->counts.Allocate( kCountsPerBounce * 4 * static_cast< size_t >( s->maxDepth > 0 ? s->maxDepth : 1 ) * units ) );
    HIP_TRY( s->queue0.Allocate( entries * 4 ) );
    HIP_TRY( s->queue1.Allocate( entries * 4 ) );
    HIP_TRY( s->shadowPaths.Allocate( entries * 4 ) );
    HIP_TRY( s->shadowRays.Allocate( entries * slots * 4 ) );
    if ( s->view.numSpheres ) HIP_TRY( s->sphereQueue.Allocate( entries * 4 ) );
    if ( s->orderedNodes.ptr || s->wideNodes.ptr ) HIP_TRY( s->redo.Allocate( entries * 4 ) );
    PathBuffers& p = s->paths;
    p.rayO[0]     = s->rayO0.As< Packed3 >();
    p.rayO[1]     = s->rayO1.As< Packed3 >();
    p.rayD[0]     = s->rayD0.As< Packed3 >();
    p.rayD[1]     = s->rayD1.As< Packed3 >();
    p.shadowO     = s->shadowO.As< Packed3 >();
    p.thr[0]      = s->thr0.As< Packed3 >();
    p.thr[1]      = s->thr1.As< Packed3 >();
    p.rad         = s->rad.As< float4 >();
    p.hit         = s->hit.As< float4 >();
    p.shDir       = s->shDir.As< float4 >();
    p.shEst       = s->shEst.As< float4 >();
    p.queue[0]    = s->queue0.As< uint32_t >();
    p.queue[1]    = s->queue1.As< uint32_t >();
    p.shadowPaths = s->shadowPaths.As< uint32_t >();
    p.shadowRays  = s->shadowRays.As< uint32_t >();
    p.sphereQueue = s->sphereQueue.As< uint32_t >();
    p.counts      = s->counts.As< uint32_t >();
    p.cursors     = s->cursors.As< uint32_t >();
    p.redo        = s->redo.As< uint32_t >();
    p.redoCount   = p.cursors + 2 * static_cast< size_t >( s->maxDepth > 0 ? s->maxDepth : 1 );
    p.capacity    = capacity;
    p.shadowSlots = static_cast< uint32_t >( slots );
    p.directSum   = s->directSum ? 1u : 0u;
    return PT_OK;
}

// Camera frame of src/path_tracer.cpp:168-173, in host float arithmetic (tanf from the host's libm).
CameraFrame MakeCameraFrame( const PtCamera& cam, int width, int height )
{
    auto add = []( f3h a, f3h b ) { return f3h{ a.x + b.x, a.y + b.y, a.z + b.z }; };
    auto mul = []( f3h a, float k ) { return f3h{ a.x * k, a.y * k, a.z * k }; };
    f3h position = Load3( cam.position ), view = Load3( cam.view_dir ), up = Load3( cam.up_dir ), right = Load3( cam.right_dir );
    float halfHeight = std::tan( cam.vfov / 2 );
    float halfWidth  = halfHeight * cam.aspect_ratio;
    f3h UL = add( add( position, view ), mul( up, halfHeight ) ) - mul( right, halfWidth );
    f3h dU = mul( right, 2 * halfWidth / width );
    f3h dV = mul( f3h{ -up.x, -up.y, -up.z }, 2 * halfHeight / height );
    UL     = add( UL, mul( add( dU, dV ), 0.5f ) );
    CameraFrame f;
    std::memcpy( f.position, &position, 12 );
    std::memcpy( f.UL, &UL, 12 );
    std::memcpy( f.dU, &dU, 12 );
    std::memcpy( f.dV, &dV, 12 );
    return f;
}

struct SpanTimer
{
    PtScene* s;
    hipStream_t stream;
    bool enabled;
    int Begin( int kernel )
    {
        s->launches[kernel]++;
        if ( !enabled ) return -1;
        while ( s->eventPool.size() < s->eventsUsed + 2 )
        {
            hipEvent_t e;
            if ( hipEventCreate( &e ) != hipSuccess ) return -1;
            s->eventPool.push_back( e );
        }
        PtScene::TimedSpan span{ kernel, s->eventPool[s->eventsUsed], s->eventPool[s->eventsUsed + 1] };
        s->eventsUsed += 2;
        (void)hipEventRecord( span.start, stream );
        s->spans.push_back( span );
        return static_cast< int >( s->spans.size() ) - 1;
    }
    void End( int idx )
    {
        if ( idx >= 0 ) (void)hipEventRecord( s->spans[idx].stop, stream );
    }
};

} // namespace

extern "C" {

const char* pt_last_error( void ) { return g_error.c_str(); }

int pt_upload_scene( const PtSceneDesc* desc, PtScene** out )
try
{
    if ( !desc || !out ) return Fail( PT_ERR_INVALID_ARG, "pt_upload_scene: null argument" );
    *out = nullptr;
    std::vector< uint32_t > escape;
    int rc = Validate( desc, escape );
    if ( rc != PT_OK ) return rc;

    int deviceCount = 0;
    HIP_TRY( hipGetDeviceCount( &deviceCount ) );
    if ( desc->device < 0 || desc->device >= deviceCount ) return Fail( PT_ERR_DEVICE, "device %d requested, %d GPU(s) visible", desc->device, deviceCount );
    HIP_TRY( hipSetDevice( desc->device ) );
    {
        // set-up for pt_trace_closest_static: this GPU's address of the staging array.  Should the lookup fail, that entry
        // point will say so when it is called; nothing else needs the array.
        unsigned char* staging = nullptr;
        if ( ScratchAddress( desc->device, &staging ) != hipSuccess ) (void)hipGetLastError();
    }

    std::unique_ptr< PtScene > owner( new PtScene ); // freed on every early return and if anything below throws
    PtScene* s = owner.get();
    s->device = desc->device;
    hipDeviceProp_t prop;
    if ( hipGetDeviceProperties( &prop, desc->device ) == hipSuccess && prop.multiProcessorCount > 0 ) s->gridBlocks = prop.multiProcessorCount * 8;
    {
        int extendPerCu = 0, connectPerCu = 0;
        TraversalOccupancy( desc->num_spheres > 0, &extendPerCu, &connectPerCu );
        const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        s->extendBlocks  = extendPerCu * cus;
        s->connectBlocks = connectPerCu * cus;
        int widePerCu = 1;
        WideOccupancy( desc->num_spheres > 0, &widePerCu, &s->wideThreads, &s->wideLdsStack );
        s->wideBlocks = widePerCu * cus;
        int shadowPerCu = 1;
        WideShadowOccupancy( &shadowPerCu, &s->wideShadowThreads );
        s->wideShadowBlocks = shadowPerCu * cus;
    }
    hipError_t e = hipStreamCreate( &s->stream );
    if ( e != hipSuccess ) return Fail( PT_ERR_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString( e ) );
    rc = Upload( s, desc, escape );
    if ( rc != PT_OK ) return rc;
    *out = owner.release();
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_upload_scene" );
}

int pt_destroy_scene( PtScene* scene )
try
{
    if ( !scene ) return PT_OK;
    (void)hipSetDevice( scene->device );
    (void)hipDeviceSynchronize();
    delete scene;
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_destroy_scene" );
}

int pt_clear_film( PtScene* s )
try
{
    if ( !s ) return Fail( PT_ERR_INVALID_ARG, "pt_clear_film: null scene" );
    HIP_TRY( hipSetDevice( s->device ) );
    HIP_TRY( hipMemsetAsync( s->film.ptr, 0, s->film.bytes, s->stream ) );
    HIP_TRY( hipMemsetAsync( s->stats.ptr, 0, s->stats.bytes, s->stream ) );
    HIP_TRY( hipStreamSynchronize( s->stream ) );
    s->samples = 0;
    std::memset( s->launches, 0, sizeof( s->launches ) );
    std::memset( s->kernelMs, 0, sizeof( s->kernelMs ) );
    s->renderMs = 0;
    s->rendered = false;
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_clear_film" );
}

int pt_render( PtScene* s, const PtRenderParams* p )
try
{
    if ( !s || !p ) return Fail( PT_ERR_INVALID_ARG, "pt_render: null argument" );
    if ( p->struct_size != sizeof( PtRenderParams ) ) return Fail( PT_ERR_INVALID_ARG, "PtRenderParams::struct_size is %u, expected %zu", p->struct_size, sizeof( PtRenderParams ) );
    if ( p->sample_begin < 0 || p->sample_end < p->sample_begin || p->spp_total < 1 ) return Fail( PT_ERR_INVALID_ARG, "pt_render: bad sample range [%d, %d) of %d", p->sample_begin, p->sample_end, p->spp_total );
    if ( p->rng_mode > PT_RNG_TABLE ) return Fail( PT_ERR_INVALID_ARG, "pt_render: unknown rng_mode %u", p->rng_mode );
    if ( p->rng_mode == PT_RNG_TABLE && ( !p->sample_table || p->sample_table_len == 0 || p->table_draws_per_sample < 1 ) )
        return Fail( PT_ERR_INVALID_ARG, "pt_render: PT_RNG_TABLE needs a table and table_draws_per_sample" );
    HIP_TRY( hipSetDevice( s->device ) );
    hipStream_t stream = p->stream ? static_cast< hipStream_t >( p->stream ) : s->stream;
    const auto wallStart = std::chrono::steady_clock::now();

    if ( !( p->flags & PT_RENDER_ACCUMULATE ) )
    {
        int rc = pt_clear_film( s );
        if ( rc != PT_OK ) return rc;
    }

    const uint64_t pixels = static_cast< uint64_t >( s->width ) * s->height;
    const int totalSamples = p->sample_end - p->sample_begin;
    // Samples of every pixel in flight at once: as many as fit 256 Mi paths, 64 GiB of path state (and half of
    // the memory that is free when the buffers have to grow) and 32-bit shadow record indices.  Large passes
    // matter: every traversal launch ends with waves draining their last, longest rays, a fixed cost per launch
    // (measured at 1080p: 2 / 8 / 32 / 128 samples per pass = 1.59 / 2.44 / 2.76 / 2.90 Grays/s).
    uint64_t perPass = p->samples_per_pass > 0 ? static_cast< uint64_t >( p->samples_per_pass ) : 0;
    if ( perPass == 0 )
    {
        const uint64_t slots = s->view.shadowSlots > 0 ? s->view.shadowSlots : 1;
        uint64_t budget = 64ull << 30;
        size_t freeBytes = 0, totalBytes = 0;
        if ( hipMemGetInfo( &freeBytes, &totalBytes ) == hipSuccess )
            budget = std::min< uint64_t >( budget, std::max< uint64_t >( freeBytes / 2, s->paths.capacity * s->bytesPerPathFixed ) );
        uint64_t maxPaths = std::min< uint64_t >( 256ull << 20, budget / s->bytesPerPathFixed );
        maxPaths = std::min< uint64_t >( maxPaths, 0xfffffffeull / slots - kUnitPaths );
        perPass = std::max< uint64_t >( 1, maxPaths / pixels );
    }
    if ( perPass > static_cast< uint64_t >( totalSamples ) ) perPass = totalSamples > 0 ? totalSamples : 1;
    {
        const uint64_t slots = s->view.shadowSlots > 0 ? s->view.shadowSlots : 1;
        if ( ( perPass * pixels + kUnitPaths ) * slots > 0xfffffffeull )
            return Fail( PT_ERR_INVALID_ARG, "pt_render: %llu samples per pass of %llu pixels with %llu shadow rays per hit exceeds 2^32 shadow records",
                         (unsigned long long)perPass, (unsigned long long)pixels, (unsigned long long)slots );
    }
    if ( perPass * pixels > 0x7fffffffull ) return Fail( PT_ERR_INVALID_ARG, "pt_render: %llu samples per pass of %llu pixels exceeds 2^31 paths", (unsigned long long)perPass, (unsigned long long)pixels );
    if ( totalSamples > 0 )
    {
        // A pass size the library chose itself is halved until its buffers fit: the memory that was free a moment ago
        // may have gone to another scene on this GPU (several host threads, several processes).
        for ( ;; )
        {
            const int rc = EnsurePathBuffers( s, perPass * pixels );
            if ( rc == PT_OK ) break;
            if ( p->samples_per_pass > 0 || perPass == 1 ) return rc;
            (void)hipGetLastError();
            perPass = ( perPass + 1 ) / 2;
        }
    }

    RngParams rng{};
    rng.mode       = p->rng_mode;
    rng.seedLo     = static_cast< uint32_t >( p->seed );
    rng.seedHi     = static_cast< uint32_t >( p->seed >> 32 );
    rng.sppTotal   = p->spp_total;
    rng.tableDraws = p->table_draws_per_sample;
    if ( p->rng_mode == PT_RNG_TABLE )
    {
        HIP_TRY( s->sampleTable.Upload( p->sample_table, p->sample_table_len * 4 ) );
        rng.table    = s->sampleTable.As< float >();
        rng.tableLen = p->sample_table_len;
    }

    const CameraFrame cam = MakeCameraFrame( s->camera, s->width, s->height );
    const bool wideShadows = ( p->flags & PT_RENDER_WIDE_SHADOWS ) != 0;
    const bool wide    = ( p->flags & PT_RENDER_WIDE ) != 0;
    const bool ordered = ( p->flags & PT_RENDER_ORDERED ) != 0 || wide;
    if ( ( wide || wideShadows ) && !s->wideNodes.ptr ) return Fail( PT_ERR_STATE, "pt_render: PT_RENDER_WIDE / PT_RENDER_WIDE_SHADOWS, but the scene has no wide layout (PT_SCENE_NO_WIDE_LAYOUT, or a tree it cannot express)" );
    if ( ordered && !wide && !s->orderedNodes.ptr ) return Fail( PT_ERR_STATE, "pt_render: PT_RENDER_ORDERED, but the scene was uploaded with PT_SCENE_NO_ORDERED_LAYOUT" );
    LaunchCfg cfg{ stream, s->gridBlocks, s->extendBlocks, s->connectBlocks, ordered };
    cfg.wide       = wide;
    cfg.wideShadows = wideShadows;
    cfg.wideShadowBlocks = s->wideShadowBlocks;
    cfg.wideBlocks = s->wideBlocks;
    cfg.wideSpill  = s->wideSpill.As< uint2 >();
    SpanTimer timer{ s, stream, ( p->flags & PT_RENDER_PROFILE ) != 0 };
    uint64_t* stats = s->stats.As< uint64_t >();
    PathBuffers pb = s->paths;

    for ( int begin = p->sample_begin; begin < p->sample_end; begin += static_cast< int >( perPass ) )
    {
        const uint32_t inPass   = static_cast< uint32_t >( std::min< int >( static_cast< int >( perPass ), p->sample_end - begin ) );
        const uint32_t numPaths = inPass * static_cast< uint32_t >( pixels );
        pb.units = ( numPaths + kUnitPaths - 1 ) / kUnitPaths;
        HIP_TRY( hipMemsetAsync( s->cursors.ptr, 0, s->cursors.bytes, stream ) );

        std::memcpy( pb.origin0, cam.position, sizeof( pb.origin0 ) );
        pb.constantShadowDir = ( wideShadows && s->directSum && s->oneDirectionalLight ) ? 1u : 0u;
        std::memcpy( pb.shadowDir, s->towardsLight, sizeof( pb.shadowDir ) );
        int span = timer.Begin( PT_K_RAYGEN );
        LaunchRaygen( cfg, cam, rng, pb, s->width, s->height, static_cast< uint32_t >( begin ), numPaths );
        timer.End( span );

        if ( s->maxDepth == 0 ) HIP_TRY( hipMemsetAsync( s->rad.ptr, 0, static_cast< size_t >( numPaths ) * 16, stream ) );
        for ( int bounce = 0; bounce < s->maxDepth; ++bounce )
        {
            span = timer.Begin( PT_K_EXTEND );
            LaunchExtend( cfg, s->view, pb, bounce, numPaths, stats );
            timer.End( span );
            span = timer.Begin( PT_K_SHADE );
            LaunchShade( cfg, s->view, rng, pb, bounce, s->maxDepth, s->width, s->height, static_cast< uint32_t >( begin ), numPaths, stats );
            timer.End( span );
            if ( s->view.numLights > 0 )
            {
                span = timer.Begin( PT_K_CONNECT );
                LaunchConnect( cfg, s->view, pb, bounce, stats );
                timer.End( span );
                if ( !pb.directSum )
                {
                    span = timer.Begin( PT_K_RESOLVE );
                    LaunchResolve( cfg, s->view, pb, bounce );
                    timer.End( span );
                }
            }
        }
        span = timer.Begin( PT_K_FILM );
        LaunchFilm( cfg, pb, s->film.As< float4 >(), static_cast< uint32_t >( s->width ), static_cast< uint32_t >( s->height ), inPass );
        timer.End( span );
        HIP_TRY( hipGetLastError() );
        s->samples += numPaths;
    }

    HIP_TRY( hipStreamSynchronize( stream ) );
    for ( const PtScene::TimedSpan& span : s->spans )
    {
        float ms = 0;
        if ( hipEventElapsedTime( &ms, span.start, span.stop ) == hipSuccess ) s->kernelMs[span.kernel] += ms;
    }
    s->spans.clear();
    s->eventsUsed = 0;
    s->rendered   = true;
    s->renderMs += std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - wallStart ).count();
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_render" );
}

int pt_readback( PtScene* s, uint32_t tonemap, float* rgb_linear, float* rgb_tonemapped, PtStats* stats )
try
{
    if ( !s ) return Fail( PT_ERR_INVALID_ARG, "pt_readback: null scene" );
    if ( tonemap > PT_TONEMAP_NONE ) return Fail( PT_ERR_INVALID_ARG, "pt_readback: unknown tonemap operator %u", tonemap );
    if ( !s->rendered && ( rgb_linear || rgb_tonemapped ) ) return Fail( PT_ERR_STATE, "pt_readback: nothing has been rendered into the film" );
    HIP_TRY( hipSetDevice( s->device ) );
    const size_t pixels = static_cast< size_t >( s->width ) * s->height;
    if ( rgb_linear || rgb_tonemapped )
    {
        if ( !s->linearOut.ptr ) HIP_TRY( s->linearOut.Allocate( pixels * 12 ) );
        if ( !s->tonemappedOut.ptr ) HIP_TRY( s->tonemappedOut.Allocate( pixels * 12 ) );
        const LaunchCfg cfg{ s->stream, s->gridBlocks, s->extendBlocks, s->connectBlocks };
        s->launches[PT_K_TONEMAP]++;
        LaunchTonemap( cfg, s->film.As< float4 >(), rgb_linear ? s->linearOut.As< float >() : nullptr,
                       rgb_tonemapped ? s->tonemappedOut.As< float >() : nullptr, static_cast< uint32_t >( pixels ), s->camera.exposure,
                       s->camera.gamma, tonemap );
        HIP_TRY( hipGetLastError() );
        if ( rgb_linear ) HIP_TRY( hipMemcpyAsync( rgb_linear, s->linearOut.ptr, pixels * 12, hipMemcpyDeviceToHost, s->stream ) );
        if ( rgb_tonemapped ) HIP_TRY( hipMemcpyAsync( rgb_tonemapped, s->tonemappedOut.ptr, pixels * 12, hipMemcpyDeviceToHost, s->stream ) );
        HIP_TRY( hipStreamSynchronize( s->stream ) );
    }
    if ( stats )
    {
        std::vector< uint64_t > dev( kStatsWords );
        HIP_TRY( hipMemcpy( dev.data(), s->stats.ptr, kStatsWords * 8, hipMemcpyDeviceToHost ) );
        for ( uint32_t k = 0; k < kShadedSlots; ++k ) dev[ST_SHADED] += dev[ST_COUNT + static_cast< size_t >( k ) * kShadedStride]; // see kShadedSlots
        std::memset( stats, 0, sizeof( *stats ) );
        stats->samples              = s->samples;
        stats->rays_closest         = dev[ST_RAYS_CLOSEST];
        stats->rays_shadow          = dev[ST_RAYS_SHADOW];
        stats->nodes_closest        = dev[ST_NODES_CLOSEST];
        stats->nodes_shadow         = dev[ST_NODES_SHADOW];
        stats->tri_tests_closest    = dev[ST_TRI_CLOSEST];
        stats->tri_tests_shadow     = dev[ST_TRI_SHADOW];
        stats->sphere_tests_closest = dev[ST_SPH_CLOSEST];
        stats->sphere_tests_shadow  = dev[ST_SPH_SHADOW];
        stats->shaded_hits          = dev[ST_SHADED];
        stats->node_steps_closest   = dev[ST_STEPS_CLOSEST];
        stats->node_steps_shadow    = dev[ST_STEPS_SHADOW];
        stats->rays_redone          = dev[ST_RAYS_REDONE];
        std::memcpy( stats->launches, s->launches, sizeof( s->launches ) );
        std::memcpy( stats->kernel_ms, s->kernelMs, sizeof( s->kernelMs ) );
        stats->render_ms = s->renderMs;
    }
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_readback" );
}

int pt_read_layout( PtScene* s, uint32_t which, void* dst, uint64_t capacity, uint64_t* bytes )
try
{
    if ( !s ) return Fail( PT_ERR_INVALID_ARG, "pt_read_layout: null scene" );
    const DeviceBuffer* all[] = { &s->nodes, &s->orderedNodes, &s->prims, &s->triShade, &s->wideNodes, &s->widePrims, &s->shadowNodes, &s->shadowTopWords, &s->lightWideNodes };
    if ( which >= sizeof( all ) / sizeof( all[0] ) ) return Fail( PT_ERR_INVALID_ARG, "pt_read_layout: no layout %u", which );
    const DeviceBuffer& b = *all[which];
    if ( bytes ) *bytes = b.bytes;
    if ( !dst || b.bytes == 0 ) return PT_OK;
    if ( capacity < b.bytes ) return Fail( PT_ERR_INVALID_ARG, "pt_read_layout: %llu bytes needed, %llu given", (unsigned long long)b.bytes, (unsigned long long)capacity );
    HIP_TRY( hipSetDevice( s->device ) );
    HIP_TRY( hipMemcpy( dst, b.ptr, b.bytes, hipMemcpyDeviceToHost ) );
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_read_layout" );
}

int pt_film_device_ptr( PtScene* s, void** device_ptr, uint64_t* bytes )
try
{
    if ( !s || !device_ptr ) return Fail( PT_ERR_INVALID_ARG, "pt_film_device_ptr: null argument" );
    *device_ptr = s->film.ptr;
    if ( bytes ) *bytes = s->film.bytes;
    s->rendered = true; // the caller may fill the film itself (merged films of several GPUs)
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_film_device_ptr" );
}

int pt_merge_films_ex( PtScene* dst, PtScene* const* sources, uint32_t num_sources, uint32_t flags )
try
{
    if ( !dst || ( num_sources && !sources ) ) return Fail( PT_ERR_INVALID_ARG, "pt_merge_films: null argument" );
    if ( flags & ~static_cast< uint32_t >( PT_MERGE_STAGE_ALWAYS ) ) return Fail( PT_ERR_INVALID_ARG, "pt_merge_films: unknown flags %u", flags );
    for ( uint32_t i = 0; i < num_sources; ++i )
    {
        const PtScene* src = sources[i];
        if ( !src || src == dst ) return Fail( PT_ERR_INVALID_ARG, "pt_merge_films: source %u is null or the destination itself", i );
        if ( src->width != dst->width || src->height != dst->height ) return Fail( PT_ERR_INVALID_ARG, "pt_merge_films: source %u is %d x %d, the destination %d x %d", i, src->width, src->height, dst->width, dst->height );
    }
    // whatever happens below, the calling thread is left on the destination's GPU
    struct OnDevice
    {
        int device;
        ~OnDevice() { (void)hipSetDevice( device ); }
    } restore{ dst->device };
    const LaunchCfg cfg{ dst->stream, dst->gridBlocks, dst->extendBlocks, dst->connectBlocks, false };
    const uint32_t pixels = static_cast< uint32_t >( dst->width ) * static_cast< uint32_t >( dst->height );
    DeviceBuffer staging; // for a source GPU whose memory this GPU cannot map (or PT_MERGE_STAGE_ALWAYS)
    for ( uint32_t i = 0; i < num_sources; ++i ) // in the order given: the sum does not depend on timing
    {
        PtScene* src = sources[i];
        HIP_TRY( hipSetDevice( src->device ) );
        HIP_TRY( hipDeviceSynchronize() ); // its film is complete, on whichever stream it was rendered (PtRenderParams::stream)
        HIP_TRY( hipSetDevice( dst->device ) );
        const float4* other = src->film.As< float4 >();
        bool staged = ( flags & PT_MERGE_STAGE_ALWAYS ) != 0;
        if ( !staged && src->device != dst->device )
        {
            int mappable = 0;
            HIP_TRY( hipDeviceCanAccessPeer( &mappable, dst->device, src->device ) );
            if ( mappable )
            {
                const hipError_t e = hipDeviceEnablePeerAccess( src->device, 0 );
                if ( e != hipSuccess ) (void)hipGetLastError(); // already enabled, or refused: either way not an error of this call
                if ( e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled ) mappable = 0;
            }
            staged = !mappable;
        }
        if ( staged )
        {
            if ( !staging.ptr ) HIP_TRY( staging.Allocate( dst->film.bytes ) );
            HIP_TRY( hipMemcpyPeerAsync( staging.ptr, dst->device, src->film.ptr, src->device, dst->film.bytes, dst->stream ) );
            other = staging.As< float4 >();
        }
        LaunchAddFilm( cfg, dst->film.As< float4 >(), other, pixels );
        HIP_TRY( hipGetLastError() );
        if ( staged ) HIP_TRY( hipStreamSynchronize( dst->stream ) ); // before the next source overwrites the buffer
    }
    HIP_TRY( hipStreamSynchronize( dst->stream ) );
    dst->rendered = true;
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_merge_films" );
}

int pt_merge_films( PtScene* dst, PtScene* const* sources, uint32_t num_sources ) { return pt_merge_films_ex( dst, sources, num_sources, 0u ); }

int pt_download_film( PtScene* s, float* film_rgba )
try
{
    if ( !s || !film_rgba ) return Fail( PT_ERR_INVALID_ARG, "pt_download_film: null argument" );
    HIP_TRY( hipSetDevice( s->device ) );
    HIP_TRY( hipMemcpyAsync( film_rgba, s->film.ptr, s->film.bytes, hipMemcpyDeviceToHost, s->stream ) );
    HIP_TRY( hipStreamSynchronize( s->stream ) );
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_download_film" );
}

int pt_upload_film( PtScene* s, const float* film_rgba )
try
{
    if ( !s || !film_rgba ) return Fail( PT_ERR_INVALID_ARG, "pt_upload_film: null argument" );
    HIP_TRY( hipSetDevice( s->device ) );
    HIP_TRY( hipMemcpyAsync( s->film.ptr, film_rgba, s->film.bytes, hipMemcpyHostToDevice, s->stream ) );
    HIP_TRY( hipStreamSynchronize( s->stream ) );
    s->rendered = true;
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_upload_film" );
}

int pt_trace_closest_mode( PtScene* s, uint32_t traversal, uint64_t n, const float* origins, const float* directions, int32_t* prim, float* t,
                           float* u, float* v, uint32_t* nodes_visited, uint32_t* prim_tests )
try
{
    if ( !s || ( n && ( !origins || !directions ) ) ) return Fail( PT_ERR_INVALID_ARG, "pt_trace_closest: null argument" );
    if ( traversal > PT_TRAVERSAL_WIDE ) return Fail( PT_ERR_INVALID_ARG, "pt_trace_closest: unknown traversal order %u", traversal );
    if ( traversal == PT_TRAVERSAL_ORDERED && !s->orderedNodes.ptr ) return Fail( PT_ERR_STATE, "pt_trace_closest: PT_TRAVERSAL_ORDERED, but the scene was uploaded with PT_SCENE_NO_ORDERED_LAYOUT" );
    if ( traversal == PT_TRAVERSAL_WIDE && !s->wideNodes.ptr ) return Fail( PT_ERR_STATE, "pt_trace_closest: PT_TRAVERSAL_WIDE, but the scene has no wide layout" );
    if ( n == 0 ) return PT_OK;
    HIP_TRY( hipSetDevice( s->device ) );
    DeviceBuffer dO, dD, dPrim, dT, dU, dV, dN, dP;
    HIP_TRY( dO.Upload( origins, n * 12 ) );
    HIP_TRY( dD.Upload( directions, n * 12 ) );
    HIP_TRY( dPrim.Allocate( n * 4 ) );
    HIP_TRY( dT.Allocate( n * 4 ) );
    HIP_TRY( dU.Allocate( n * 4 ) );
    HIP_TRY( dV.Allocate( n * 4 ) );
    HIP_TRY( dN.Allocate( n * 4 ) );
    HIP_TRY( dP.Allocate( n * 4 ) );
    LaunchCfg cfg{ s->stream, s->gridBlocks, s->extendBlocks, s->connectBlocks, traversal == PT_TRAVERSAL_ORDERED };
    cfg.wide = traversal == PT_TRAVERSAL_WIDE;
    LaunchTraceClosest( cfg, s->view, n, dO.As< float >(), dD.As< float >(), dPrim.As< int32_t >(), dT.As< float >(), dU.As< float >(),
                        dV.As< float >(), dN.As< uint32_t >(), dP.As< uint32_t >() );
    HIP_TRY( hipGetLastError() );
    HIP_TRY( hipStreamSynchronize( s->stream ) );
    if ( prim ) HIP_TRY( hipMemcpy( prim, dPrim.ptr, n * 4, hipMemcpyDeviceToHost ) );
    if ( t ) HIP_TRY( hipMemcpy( t, dT.ptr, n * 4, hipMemcpyDeviceToHost ) );
    if ( u ) HIP_TRY( hipMemcpy( u, dU.ptr, n * 4, hipMemcpyDeviceToHost ) );
    if ( v ) HIP_TRY( hipMemcpy( v, dV.ptr, n * 4, hipMemcpyDeviceToHost ) );
    if ( nodes_visited ) HIP_TRY( hipMemcpy( nodes_visited, dN.ptr, n * 4, hipMemcpyDeviceToHost ) );
    if ( prim_tests ) HIP_TRY( hipMemcpy( prim_tests, dP.ptr, n * 4, hipMemcpyDeviceToHost ) );
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_trace_closest" );
}

int pt_trace_closest( PtScene* s, uint64_t n, const float* origins, const float* directions, int32_t* prim, float* t, float* u, float* v,
                      uint32_t* nodes_visited, uint32_t* prim_tests )
{
    return pt_trace_closest_mode( s, PT_TRAVERSAL_REFERENCE, n, origins, directions, prim, t, u, v, nodes_visited, prim_tests );
}

int pt_trace_closest_static( PtScene* s, uint64_t n, const float* origins, const float* directions, int32_t* prim, float* t, float* u, float* v,
                             uint32_t* nodes_visited, uint32_t* prim_tests )
try
{
    if ( !s || ( n && ( !origins || !directions ) ) ) return Fail( PT_ERR_INVALID_ARG, "pt_trace_closest_static: null argument" );
    if ( n > kStaticRays )
        return Fail( PT_ERR_INVALID_ARG, "pt_trace_closest_static: %llu rays, the staging array holds %llu (PT_TRACE_STATIC_MAX_RAYS); split the batch or use pt_trace_closest",
                     (unsigned long long)n, (unsigned long long)kStaticRays );
    if ( n == 0 ) return PT_OK;
    HIP_TRY( hipSetDevice( s->device ) );
    unsigned char* base = nullptr;
    HIP_TRY( ScratchAddress( s->device, &base ) ); // found at pt_upload_scene as a rule; a table lookup here
    float* dO       = reinterpret_cast< float* >( base );
    float* dD       = reinterpret_cast< float* >( base + kStaticRays * 12 );
    int32_t* dPrim  = reinterpret_cast< int32_t* >( base + kStaticRays * 24 );
    float* dT       = reinterpret_cast< float* >( base + kStaticRays * 28 );
    float* dU       = reinterpret_cast< float* >( base + kStaticRays * 32 );
    float* dV       = reinterpret_cast< float* >( base + kStaticRays * 36 );
    uint32_t* dN    = reinterpret_cast< uint32_t* >( base + kStaticRays * 40 );
    uint32_t* dP    = reinterpret_cast< uint32_t* >( base + kStaticRays * 44 );
    HIP_TRY( hipMemcpyAsync( dO, origins, n * 12, hipMemcpyHostToDevice, s->stream ) );
    HIP_TRY( hipMemcpyAsync( dD, directions, n * 12, hipMemcpyHostToDevice, s->stream ) );
    const LaunchCfg cfg{ s->stream, s->gridBlocks, s->extendBlocks, s->connectBlocks, false };
    LaunchTraceClosest( cfg, s->view, n, dO, dD, dPrim, dT, dU, dV, dN, dP ); // every output element below n is written
    HIP_TRY( hipGetLastError() );
    if ( prim ) HIP_TRY( hipMemcpyAsync( prim, dPrim, n * 4, hipMemcpyDeviceToHost, s->stream ) );
    if ( t ) HIP_TRY( hipMemcpyAsync( t, dT, n * 4, hipMemcpyDeviceToHost, s->stream ) );
    if ( u ) HIP_TRY( hipMemcpyAsync( u, dU, n * 4, hipMemcpyDeviceToHost, s->stream ) );
    if ( v ) HIP_TRY( hipMemcpyAsync( v, dV, n * 4, hipMemcpyDeviceToHost, s->stream ) );
    if ( nodes_visited ) HIP_TRY( hipMemcpyAsync( nodes_visited, dN, n * 4, hipMemcpyDeviceToHost, s->stream ) );
    if ( prim_tests ) HIP_TRY( hipMemcpyAsync( prim_tests, dP, n * 4, hipMemcpyDeviceToHost, s->stream ) );
    HIP_TRY( hipStreamSynchronize( s->stream ) );
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_trace_closest_static" );
}

int pt_trace_any_mode( PtScene* s, uint32_t traversal, uint64_t n, const float* origins, const float* directions, const float* t_max,
                       uint8_t* occluded, uint32_t* nodes_visited, uint32_t* prim_tests )
try
{
    if ( !s || ( n && ( !origins || !directions || !t_max || !occluded ) ) ) return Fail( PT_ERR_INVALID_ARG, "pt_trace_any: null argument" );
    if ( traversal > PT_TRAVERSAL_WIDE ) return Fail( PT_ERR_INVALID_ARG, "pt_trace_any: unknown traversal order %u", traversal );
    if ( traversal == PT_TRAVERSAL_ORDERED && !s->orderedNodes.ptr ) return Fail( PT_ERR_STATE, "pt_trace_any: PT_TRAVERSAL_ORDERED, but the scene was uploaded with PT_SCENE_NO_ORDERED_LAYOUT" );
    if ( traversal == PT_TRAVERSAL_WIDE && !s->wideNodes.ptr ) return Fail( PT_ERR_STATE, "pt_trace_any: PT_TRAVERSAL_WIDE, but the scene has no wide layout" );
    if ( n == 0 ) return PT_OK;
    HIP_TRY( hipSetDevice( s->device ) );
    DeviceBuffer dO, dD, dT, dOcc, dN, dP;
    HIP_TRY( dO.Upload( origins, n * 12 ) );
    HIP_TRY( dD.Upload( directions, n * 12 ) );
    HIP_TRY( dT.Upload( t_max, n * 4 ) );
    HIP_TRY( dOcc.Allocate( n ) );
    HIP_TRY( dN.Allocate( n * 4 ) );
    HIP_TRY( dP.Allocate( n * 4 ) );
    LaunchCfg cfg{ s->stream, s->gridBlocks, s->extendBlocks, s->connectBlocks, traversal == PT_TRAVERSAL_ORDERED };
    cfg.wide = traversal == PT_TRAVERSAL_WIDE;
    LaunchTraceAny( cfg, s->view, n, dO.As< float >(), dD.As< float >(), dT.As< float >(), dOcc.As< uint8_t >(), dN.As< uint32_t >(), dP.As< uint32_t >() );
    HIP_TRY( hipGetLastError() );
    HIP_TRY( hipStreamSynchronize( s->stream ) );
    HIP_TRY( hipMemcpy( occluded, dOcc.ptr, n, hipMemcpyDeviceToHost ) );
    if ( nodes_visited ) HIP_TRY( hipMemcpy( nodes_visited, dN.ptr, n * 4, hipMemcpyDeviceToHost ) );
    if ( prim_tests ) HIP_TRY( hipMemcpy( prim_tests, dP.ptr, n * 4, hipMemcpyDeviceToHost ) );
    return PT_OK;
}
catch ( ... )
{
    return HostFailure( "pt_trace_any" );
}

int pt_trace_any( PtScene* s, uint64_t n, const float* origins, const float* directions, const float* t_max, uint8_t* occluded,
                  uint32_t* nodes_visited, uint32_t* prim_tests )
{
    return pt_trace_any_mode( s, PT_TRAVERSAL_REFERENCE, n, origins, directions, t_max, occluded, nodes_visited, prim_tests );
}

} // extern "C"
