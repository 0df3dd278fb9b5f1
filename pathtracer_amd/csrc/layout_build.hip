// Kernels that write the device records of a scene from its description (see layout_build.h).  Each mirrors a loop of
// Upload in api.hip word for word; the host loops stay as the checker (PT_SCENE_HOST_LAYOUTS).
#include "layout_build.h"
#include "pt_device.h"

namespace ptd
{

namespace
{

constexpr int kBlock = 256;
constexpr uint32_t kScanItems = 16;                  // nodes per thread of the prefix count
constexpr uint32_t kScanTile  = kBlock * kScanItems; // nodes per block
constexpr uint32_t kUnset     = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t GlobalIndex() { return blockIdx.x * kBlock + threadIdx.x; }

inline uint32_t BlocksFor( uint64_t n ) { return static_cast< uint32_t >( ( n + kBlock - 1 ) / kBlock ); }

// Largest k with sorted[k] <= x, or count if there is none ( sorted ascending )
__device__ __forceinline__ uint32_t LastNotAbove( const uint32_t* __restrict__ sorted, uint32_t count, uint32_t x )
{
    uint32_t lo = 0, hi = count; // first element above x
    while ( lo < hi )
    {
        const uint32_t mid = ( lo + hi ) >> 1;
        if ( sorted[mid] <= x )
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo == 0 ? count : lo - 1;
}

// ---- first layout

// Device index of reference node i: the top of the tree first, in the given order, everything else behind it in the
// reference's order
__device__ __forceinline__ uint32_t Place( const uint32_t* __restrict__ topSorted, const uint32_t* __restrict__ topRank, uint32_t topCount, uint32_t numNodes, uint32_t i )
{
    if ( i >= numNodes ) return numNodes; // the end of the traversal
    const uint32_t k = LastNotAbove( topSorted, topCount, i );
    if ( k != topCount && topSorted[k] == i ) return topRank[k];
    const uint32_t before = k == topCount ? 0u : k + 1u; // top nodes with a smaller index
    return topCount + i - before;
}

__global__ __launch_bounds__( kBlock ) void ReferenceNodesKernel( LayoutInputs in, const uint32_t* __restrict__ topSorted, const uint32_t* __restrict__ topRank,
                                                                  uint32_t topCount, float4* __restrict__ out )
{
    const uint32_t i = GlobalIndex();
    if ( i >= in.numNodes ) return;
    const PtBvhNode n = in.nodes[i];
    uint32_t link = Place( topSorted, topRank, topCount, in.numNodes, in.escape[i] ), word;
    if ( n.num_shapes > 0 )
    {
        link |= kLeafFlag;
        word = static_cast< uint32_t >( n.offset ) | ( static_cast< uint32_t >( n.num_shapes - 1 ) << 28 );
    }
    else
        word = Place( topSorted, topRank, topCount, in.numNodes, i + 1 );
    float4* o = out + 2ull * Place( topSorted, topRank, topCount, in.numNodes, i );
    o[0] = make_float4( n.aabb_min[0], n.aabb_min[1], n.aabb_max[0], n.aabb_max[1] );
    o[1] = make_float4( n.aabb_min[2], n.aabb_max[2], __uint_as_float( link ), __uint_as_float( word ) );
}

// ---- second layout

// innerPrefix[i] = inner nodes with a reference index below i, i = 0 .. numNodes: block totals, their scan, the rest
__device__ __forceinline__ uint32_t InnerFlag( const LayoutInputs& in, uint32_t i ) { return i < in.numNodes && in.nodes[i].num_shapes == 0 ? 1u : 0u; }

__global__ __launch_bounds__( kBlock ) void InnerCountKernel( LayoutInputs in, uint32_t* __restrict__ blockSums )
{
    __shared__ uint32_t total;
    if ( threadIdx.x == 0 ) total = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t mine = 0;
    for ( uint32_t k = 0; k < kScanItems; ++k ) mine += InnerFlag( in, first + k );
    atomicAdd( &total, mine );
    __syncthreads();
    if ( threadIdx.x == 0 ) blockSums[blockIdx.x] = total;
}

__global__ __launch_bounds__( 1 ) void ScanBlockSumsKernel( uint32_t* __restrict__ blockSums, uint32_t count )
{
    // a few thousand entries at most: one thread
    uint32_t run = 0;
    for ( uint32_t k = 0; k < count; ++k )
    {
        const uint32_t v = blockSums[k];
        blockSums[k] = run;
        run += v;
    }
    blockSums[count] = run;
}

__global__ __launch_bounds__( kBlock ) void InnerPrefixKernel( LayoutInputs in, const uint32_t* __restrict__ blockSums, uint32_t* __restrict__ innerPrefix )
{
    __shared__ uint32_t sums[kBlock];
    const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t flags = 0, mine = 0;
    for ( uint32_t k = 0; k < kScanItems; ++k )
    {
        const uint32_t f = InnerFlag( in, first + k );
        flags |= f << k;
        mine += f;
    }
    sums[threadIdx.x] = mine;
    __syncthreads();
    for ( uint32_t step = 1; step < kBlock; step <<= 1 )
    {
        const uint32_t add = threadIdx.x >= step ? sums[threadIdx.x - step] : 0u;
        __syncthreads();
        sums[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = blockSums[blockIdx.x] + sums[threadIdx.x] - mine;
    for ( uint32_t k = 0; k < kScanItems; ++k )
    {
        if ( first + k <= in.numNodes ) innerPrefix[first + k] = run;
        run += ( flags >> k ) & 1u;
    }
}

__global__ __launch_bounds__( kBlock ) void TopRecordsKernel( const uint32_t* __restrict__ topOrder, uint32_t topCount, uint32_t* __restrict__ at )
{
    // the root at 0, record 1 unused, then the sibling pairs of the top in their order
    const uint32_t k = GlobalIndex();
    if ( k < topCount ) at[topOrder[k]] = k == 0 ? 0u : k + 1u;
}

// Below the top a pair of children is followed by the first child's subtree, then the second's: the pairs of a subtree
// come in the pre-order of their parents, so the position of a pair is the number of inner nodes before its parent
__global__ __launch_bounds__( kBlock ) void PairRecordsKernel( LayoutInputs in, const uint32_t* __restrict__ roots, const uint32_t* __restrict__ rootBase, uint32_t numRoots,
                                                               const uint32_t* __restrict__ innerPrefix, uint32_t* __restrict__ at, uint32_t* __restrict__ parent )
{
    const uint32_t x = GlobalIndex();
    if ( x >= in.numNodes ) return;
    if ( x == 0 ) parent[0] = kUnset;
    const PtBvhNode n = in.nodes[x];
    if ( n.num_shapes > 0 ) return;
    const uint32_t a = x + 1, b = static_cast< uint32_t >( n.offset );
    parent[a] = x;
    parent[b] = x;
    if ( at[a] != kUnset ) return; // a pair of the top (only this thread ever writes at[a])
    const uint32_t j    = LastNotAbove( roots, numRoots, x );
    const uint32_t pair = rootBase[j] + 2u * ( innerPrefix[x] - innerPrefix[roots[j]] );
    at[a] = pair;
    at[b] = pair + 1u;
}

__global__ __launch_bounds__( kBlock ) void OrderedNodesKernel( LayoutInputs in, const uint32_t* __restrict__ at, const uint32_t* __restrict__ parent, uint32_t numOrdered,
                                                                uint4* __restrict__ out )
{
    const uint32_t i = GlobalIndex();
    if ( i == 0 ) out[4] = out[5] = out[6] = out[7] = make_uint4( 0u, 0u, 0u, 0u ); // record 1
    if ( i >= in.numNodes ) return;
    const PtBvhNode n = in.nodes[i];
    const uint32_t below = in.sphereBelow && in.sphereBelow[i] ? kSphereBelowFlag : 0u;
    uint32_t w6, w7;
    if ( n.num_shapes > 0 )
    {
        w6 = static_cast< uint32_t >( n.offset ) | ( static_cast< uint32_t >( n.num_shapes - 1 ) << 28 );
        w7 = kLeafFlag | below;
    }
    else
    {
        w6 = at[i + 1];
        w7 = ( 2u + ( n.axis < 3 ? n.axis : 0u ) ) | below;
    }
    // Per octant of ray directions the record that follows this node's subtree when near children go first: the sibling
    // of the first node on the way up that is its parent's near child, or the end.  Half of the octants are settled at
    // every level whose split axis is new, so the walk is short.
    uint32_t links[8];
    for ( int o = 0; o < 8; ++o ) links[o] = numOrdered;
    uint32_t open = 0xFFu;
    for ( uint32_t y = i; open != 0 && y != 0; )
    {
        const uint32_t p   = parent[y];
        const PtBvhNode pn = in.nodes[p];
        const uint32_t axis = pn.axis < 3 ? pn.axis : 0u;
        const bool first    = y == p + 1;
        const uint32_t lowSide = axis == 0 ? 0x55u : axis == 1 ? 0x33u : 0x0Fu; // octants whose bit `axis` is clear: the first child is the near one
        const uint32_t mine    = open & ( first ? lowSide : ~lowSide );
        if ( mine )
        {
            const uint32_t sibling = at[first ? static_cast< uint32_t >( pn.offset ) : p + 1];
            for ( int o = 0; o < 8; ++o )
                if ( ( mine >> o ) & 1u ) links[o] = sibling;
            open &= ~mine;
        }
        y = p;
    }
    uint4* r = out + 4ull * at[i];
    r[0] = make_uint4( __float_as_uint( n.aabb_min[0] ), __float_as_uint( n.aabb_min[1] ), __float_as_uint( n.aabb_max[0] ), __float_as_uint( n.aabb_max[1] ) );
    r[1] = make_uint4( __float_as_uint( n.aabb_min[2] ), __float_as_uint( n.aabb_max[2] ), w6, w7 );
    r[2] = make_uint4( links[0], links[1], links[2], links[3] );
    r[3] = make_uint4( links[4], links[5], links[6], links[7] );
}

// ---- primitives

__device__ __forceinline__ const float* VertexOf( const float* attribute, const PtMesh& m, uint32_t index, int stride )
{
    return attribute + static_cast< size_t >( stride ) * ( m.vertex_offset + index );
}

__global__ __launch_bounds__( kBlock ) void PrimRecordsKernel( LayoutInputs in, float4* __restrict__ prims, float4* __restrict__ shade )
{
    const uint64_t i = static_cast< uint64_t >( blockIdx.x ) * kBlock + threadIdx.x;
    if ( i >= in.numPrims ) return;
    const PtPrim p = in.prims[i];
    float q[16], a[32];
    for ( int k = 0; k < 16; ++k ) q[k] = 0.0f;
    for ( int k = 0; k < 32; ++k ) a[k] = 0.0f;
    if ( p.type == PT_PRIM_TRIANGLE )
    {
        const PtTriangle t = in.triangles[p.index];
        const PtMesh m     = in.meshes[t.mesh];
        const uint32_t idx[3] = { t.i0, t.i1, t.i2 };
        const float *v0 = VertexOf( in.vertices, m, t.i0, 3 ), *v1 = VertexOf( in.vertices, m, t.i1, 3 ), *v2 = VertexOf( in.vertices, m, t.i2, 3 );
        q[0] = v0[0]; q[1] = v0[1]; q[2] = v0[2];
        q[3] = __uint_as_float( kTagTriangle );
        q[4] = v1[0] - v0[0]; q[5] = v1[1] - v0[1]; q[6]  = v1[2] - v0[2]; // v0v1 of src/intersection_tests.cpp:37
        q[8] = v2[0] - v0[0]; q[9] = v2[1] - v0[1]; q[10] = v2[2] - v0[2]; // v0v2 of src/intersection_tests.cpp:38
        for ( int k = 0; k < 3; ++k )
        {
            const float *nn = VertexOf( in.normals, m, idx[k], 3 ), *tt = VertexOf( in.tangents, m, idx[k], 3 ), *uv = VertexOf( in.uvs, m, idx[k], 2 );
            a[3 * k] = nn[0]; a[3 * k + 1] = nn[1]; a[3 * k + 2] = nn[2];
            a[9 + 3 * k] = tt[0]; a[10 + 3 * k] = tt[1]; a[11 + 3 * k] = tt[2];
            a[18 + 2 * k] = uv[0]; a[19 + 2 * k] = uv[1];
        }
        a[24] = __int_as_float( m.material );
    }
    else
    {
        const PtSphere sp = in.spheres[p.index];
        q[0] = sp.position[0]; q[1] = sp.position[1]; q[2] = sp.position[2];
        q[3] = __uint_as_float( kTagSphere );
        q[4] = sp.radius;
        q[5] = __uint_as_float( p.index );
    }
    // words 7, 11 and 12 - 15 become the leaf's box (LeafBoxKernel, which runs after this one)
    float4* qo = prims + 4 * i;
    for ( int k = 0; k < 4; ++k ) qo[k] = make_float4( q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3] );
    float4* ao = shade + 8 * i;
    for ( int k = 0; k < 8; ++k ) ao[k] = make_float4( a[4 * k], a[4 * k + 1], a[4 * k + 2], a[4 * k + 3] );
}

__global__ __launch_bounds__( kBlock ) void LeafBoxKernel( LayoutInputs in, float4* __restrict__ prims )
{
    const uint32_t i = GlobalIndex();
    if ( i >= in.numNodes ) return;
    const PtBvhNode n = in.nodes[i];
    for ( uint32_t k = 0; k < n.num_shapes; ++k )
    {
        float* q = reinterpret_cast< float* >( prims + 4ull * ( static_cast< uint64_t >( n.offset ) + k ) );
        q[7]  = n.aabb_min[0]; q[11] = n.aabb_min[1]; q[12] = n.aabb_min[2];
        q[13] = n.aabb_max[0]; q[14] = n.aabb_max[1]; q[15] = n.aabb_max[2];
    }
}

__global__ __launch_bounds__( kBlock ) void WidePrimsKernel( LayoutInputs in, const float4* __restrict__ prims, const uint32_t* __restrict__ primOrder, float4* __restrict__ widePrims )
{
    const uint64_t k = static_cast< uint64_t >( blockIdx.x ) * kBlock + threadIdx.x;
    if ( k >= in.numPrims ) return;
    const uint32_t i = primOrder[k];
    const float4* q  = prims + 4ull * i;
    float4 q0 = q[0];
    q0.w = __uint_as_float( i | ( in.prims[i].type == PT_PRIM_SPHERE ? 0x80000000u : 0u ) );
    float4* o = widePrims + 4 * k;
    o[0] = q0;
    o[1] = q[1];
    o[2] = q[2];
    o[3] = q[3];
}

// ---- third layout: the light-space records of a scene lit by one directional light

__device__ __forceinline__ float RoundedDown( double x )
{
    float f = static_cast< float >( x );
    if ( static_cast< double >( f ) > x ) f = nextafterf( f, -INFINITY );
    return f;
}
__device__ __forceinline__ float RoundedUp( double x )
{
    float f = static_cast< float >( x );
    if ( static_cast< double >( f ) < x ) f = nextafterf( f, INFINITY );
    return f;
}

__device__ __forceinline__ float LoadShared( const float* p ) { return __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); }

__global__ __launch_bounds__( kBlock ) void ParentsKernel( LayoutInputs in, uint32_t* __restrict__ parent, uint32_t* __restrict__ verdict )
{
    const uint32_t x = GlobalIndex();
    if ( x >= in.numNodes ) return;
    if ( x == 0 ) parent[0] = kUnset;
    const PtBvhNode n = in.nodes[x];
    if ( n.num_shapes > 8 ) atomicOr( verdict, kShadowLeafTooLarge );
    if ( n.num_shapes > 0 ) return;
    parent[x + 1] = x;
    parent[static_cast< uint32_t >( n.offset )] = x;
}

// lo, hi <- joined with the extent of one primitive along the two axes of the light's frame, in double precision (the
// products of two floats are exact there, so the sums are the host's whether fused or not)
__device__ __forceinline__ void JoinPrimitive( const LayoutInputs& in, const ShadowAxes& axes, PtPrim prim, double lo[2], double hi[2] )
{
    if ( prim.type == PT_PRIM_TRIANGLE )
    {
        const PtTriangle t = in.triangles[prim.index];
        const PtMesh m     = in.meshes[t.mesh];
        const uint32_t idx[3] = { t.i0, t.i1, t.i2 };
        for ( int c = 0; c < 3; ++c )
        {
            const float* p = VertexOf( in.vertices, m, idx[c], 3 );
            for ( int a = 0; a < 2; ++a )
            {
                const float* ax = a ? axes.v : axes.u;
                const double q = static_cast< double >( p[0] ) * ax[0] + static_cast< double >( p[1] ) * ax[1] + static_cast< double >( p[2] ) * ax[2];
                lo[a] = q < lo[a] ? q : lo[a];
                hi[a] = hi[a] < q ? q : hi[a];
            }
        }
    }
    else
    {
        const PtSphere sp = in.spheres[prim.index];
        for ( int a = 0; a < 2; ++a )
        {
            const float* ax = a ? axes.v : axes.u;
            const double q   = static_cast< double >( sp.position[0] ) * ax[0] + static_cast< double >( sp.position[1] ) * ax[1] + static_cast< double >( sp.position[2] ) * ax[2];
            const double len = sqrt( static_cast< double >( ax[0] ) * ax[0] + static_cast< double >( ax[1] ) * ax[1] + static_cast< double >( ax[2] ) * ax[2] );
            const double r   = fabs( static_cast< double >( sp.radius ) ) * len * ( 1.0 + 1e-6 );
            lo[a] = q - r < lo[a] ? q - r : lo[a];
            hi[a] = hi[a] < q + r ? q + r : hi[a];
        }
    }
}

// Bounds of every node in the light's frame {u0, u1, v0, v1}, from the geometry below it: a leaf from its primitives, then
// up the tree - the second child to arrive at a parent finds its sibling's bounds written and goes on, the first leaves.
// The bounds travel between the two through agent-scope atomic stores and loads, which go past this XCD's L2 to memory, and
// the arrival counter orders them: the stores are acknowledged before the counter is touched (that wait is all a release
// of such stores amounts to), the loads are issued after its answer.  kFullFences = false leaves it at that; true puts
// agent-scope fences around the counter, which write back and invalidate the XCD's whole L2 twice per node and level
// (40 M nodes: 211 ms against 3 ms).  FramesCheckKernel proves the result either way.
template < bool kFullFences >
__global__ __launch_bounds__( kBlock ) void ShadowFramesKernel( LayoutInputs in, ShadowAxes axes, const uint32_t* __restrict__ parent, uint32_t* __restrict__ arrivals,
                                                                float* __restrict__ frame )
{
    const uint32_t i = GlobalIndex();
    if ( i >= in.numNodes ) return;
    const PtBvhNode n = in.nodes[i];
    if ( n.num_shapes == 0 ) return;
    double lo[2] = { INFINITY, INFINITY }, hi[2] = { -INFINITY, -INFINITY };
    for ( uint32_t k = 0; k < n.num_shapes; ++k ) JoinPrimitive( in, axes, in.prims[static_cast< uint64_t >( n.offset ) + k], lo, hi );
    float f[4] = { RoundedDown( lo[0] ), RoundedUp( hi[0] ), RoundedDown( lo[1] ), RoundedUp( hi[1] ) };
    for ( uint32_t y = i;; )
    {
        float* mine = frame + 4ull * y;
        for ( int k = 0; k < 4; ++k ) __hip_atomic_store( mine + k, f[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        const uint32_t p = parent[y];
        if ( p == kUnset ) break;
        if constexpr ( kFullFences )
            __threadfence();
        else
            __builtin_amdgcn_fence( __ATOMIC_RELEASE, "workgroup" ); // waits for the stores' acknowledgements
        if ( atomicAdd( arrivals + p, 1u ) == 0 ) break; // the sibling is still on its way: it will do the parent
        if constexpr ( kFullFences )
            __threadfence();
        else
            __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "workgroup" );
        const uint32_t sibling = y == p + 1 ? static_cast< uint32_t >( in.nodes[p].offset ) : p + 1;
        const float* other = frame + 4ull * sibling;
        // min( first child, second child ) as the host writes it: the second argument only where it is smaller
        const float a0 = LoadShared( other ), a1 = LoadShared( other + 1 ), a2 = LoadShared( other + 2 ), a3 = LoadShared( other + 3 );
        const bool iAmFirst = y == p + 1;
        const float x0 = iAmFirst ? f[0] : a0, y0 = iAmFirst ? a0 : f[0], x1 = iAmFirst ? f[1] : a1, y1 = iAmFirst ? a1 : f[1];
        const float x2 = iAmFirst ? f[2] : a2, y2 = iAmFirst ? a2 : f[2], x3 = iAmFirst ? f[3] : a3, y3 = iAmFirst ? a3 : f[3];
        f[0] = y0 < x0 ? y0 : x0; // std::min( x, y )
        f[1] = x1 < y1 ? y1 : x1; // std::max( x, y )
        f[2] = y2 < x2 ? y2 : x2;
        f[3] = x3 < y3 ? y3 : x3;
        y = p;
    }
}

// Every inner node's bounds are its children's, joined: with that and the leaves' own bounds the whole array is right.
// Runs as a launch of its own, after every write of ShadowFramesKernel has landed.
__global__ __launch_bounds__( kBlock ) void FramesCheckKernel( LayoutInputs in, const float4* __restrict__ frame, uint32_t* __restrict__ verdict )
{
    const uint32_t i = GlobalIndex();
    if ( i >= in.numNodes ) return;
    const PtBvhNode n = in.nodes[i];
    if ( n.num_shapes > 0 ) return;
    const float4 x = frame[i + 1], y = frame[static_cast< uint32_t >( n.offset )], f = frame[i];
    const float f0 = y.x < x.x ? y.x : x.x, f1 = x.y < y.y ? y.y : x.y, f2 = y.z < x.z ? y.z : x.z, f3 = x.w < y.w ? y.w : x.w;
    const bool same = __float_as_uint( f0 ) == __float_as_uint( f.x ) && __float_as_uint( f1 ) == __float_as_uint( f.y ) &&
                      __float_as_uint( f2 ) == __float_as_uint( f.z ) && __float_as_uint( f3 ) == __float_as_uint( f.w );
    if ( !same ) atomicOr( verdict, kShadowFramesWrong );
}

// Is reference node i one of the first shadowTop nodes of the top of the tree, and if so which
__device__ __forceinline__ bool ShadowTopRank( const uint32_t* __restrict__ topSorted, const uint32_t* __restrict__ topRank, uint32_t topCount, uint32_t shadowTop, uint32_t i,
                                               uint32_t* rank )
{
    const uint32_t k = LastNotAbove( topSorted, topCount, i );
    if ( k == topCount || topSorted[k] != i || topRank[k] >= shadowTop ) return false;
    *rank = topRank[k];
    return true;
}

// Records a node takes in the part behind the top: none if it is in the top, else one, and one more for a leaf whose
// successor lies in the top (the record that links back there)
__global__ __launch_bounds__( kBlock ) void ShadowSlotsKernel( LayoutInputs in, const uint32_t* __restrict__ topSorted, const uint32_t* __restrict__ topRank, uint32_t topCount,
                                                               uint32_t shadowTop, uint32_t* __restrict__ slots )
{
    const uint32_t i = GlobalIndex();
    if ( i > in.numNodes ) return;
    uint32_t rank, count = 0;
    if ( i < in.numNodes && !ShadowTopRank( topSorted, topRank, topCount, shadowTop, i, &rank ) )
    {
        count = 1;
        const uint32_t next = in.escape[i];
        if ( in.nodes[i].num_shapes > 0 && next < in.numNodes && ShadowTopRank( topSorted, topRank, topCount, shadowTop, next, &rank ) ) count = 2;
    }
    slots[i] = count;
}

__global__ __launch_bounds__( kBlock ) void SumTilesKernel( const uint32_t* __restrict__ values, uint32_t count, uint32_t* __restrict__ blockSums )
{
    __shared__ uint32_t total;
    if ( threadIdx.x == 0 ) total = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t mine = 0;
    for ( uint32_t k = 0; k < kScanItems; ++k ) mine += first + k < count ? values[first + k] : 0u;
    atomicAdd( &total, mine );
    __syncthreads();
    if ( threadIdx.x == 0 ) blockSums[blockIdx.x] = total;
}

__global__ __launch_bounds__( kBlock ) void PrefixTilesKernel( uint32_t* __restrict__ values, uint32_t count, const uint32_t* __restrict__ blockSums )
{
    // values[i] <- sum of the values before i
    __shared__ uint32_t sums[kBlock];
    const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], mine = 0;
    for ( uint32_t k = 0; k < kScanItems; ++k )
    {
        v[k] = first + k < count ? values[first + k] : 0u;
        mine += v[k];
    }
    sums[threadIdx.x] = mine;
    __syncthreads();
    for ( uint32_t step = 1; step < kBlock; step <<= 1 )
    {
        const uint32_t add = threadIdx.x >= step ? sums[threadIdx.x - step] : 0u;
        __syncthreads();
        sums[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = blockSums[blockIdx.x] + sums[threadIdx.x] - mine;
    for ( uint32_t k = 0; k < kScanItems; ++k )
    {
        if ( first + k < count ) values[first + k] = run;
        run += v[k];
    }
}

// The rectangle {u0, u1, v0, v1} on the grid, widened by the slack: q[a] = first cell | ( last cell - first cell ) << 16 along axis a
__device__ __forceinline__ void CellsOfFrame( const float* f, const ShadowGrid& grid, uint32_t q[2] )
{
    for ( int a = 0; a < 2; ++a )
    {
        const double lo = f[2 * a], hi = f[2 * a + 1];
        // std::min( 65534.0, std::max( 1.0, . ) ) and std::min( 65534.0, std::max( qlo, . ) ), comparisons in their order
        double qlo = floor( ( lo - grid.origin[a] ) / grid.cell[a] - grid.slack[a] );
        qlo = 1.0 < qlo ? qlo : 1.0;
        qlo = qlo < 65534.0 ? qlo : 65534.0;
        double qhi = floor( ( hi - grid.origin[a] ) / grid.cell[a] + grid.slack[a] );
        qhi = qlo < qhi ? qhi : qlo;
        qhi = qhi < 65534.0 ? qhi : 65534.0;
        q[a] = static_cast< uint32_t >( qlo ) | static_cast< uint32_t >( qhi - qlo ) << 16;
    }
}

__global__ __launch_bounds__( kBlock ) void ShadowRecordsKernel( LayoutInputs in, const float* __restrict__ frame, ShadowGrid grid, const uint32_t* __restrict__ topSorted,
                                                                 const uint32_t* __restrict__ topRank, uint32_t topCount, const uint32_t* __restrict__ before,
                                                                 uint4* __restrict__ rec, uint32_t* __restrict__ topWords )
{
    const uint32_t i = GlobalIndex();
    if ( i >= in.numNodes ) return;
    auto recordOf = [&]( uint32_t j ) {
        uint32_t rank;
        if ( j >= in.numNodes ) return grid.numRecords; // the end of the traversal
        return ShadowTopRank( topSorted, topRank, topCount, grid.shadowTop, j, &rank ) ? rank : grid.shadowTop + before[j];
    };
    const PtBvhNode n = in.nodes[i];
    uint32_t q[2];
    CellsOfFrame( frame + 4ull * i, grid, q );
    const float wmax = grid.towardsPositive ? n.aabb_max[grid.depthAxis] : -n.aabb_min[grid.depthAxis];
    const bool leaf     = n.num_shapes > 0;
    const uint32_t next = in.escape[i];
    const uint32_t link = recordOf( next );
    const uint32_t word = leaf ? kLeafFlag | static_cast< uint32_t >( n.offset ) | ( static_cast< uint32_t >( n.num_shapes - 1 ) << 28 ) : link;
    const uint32_t at   = recordOf( i );
    rec[at] = make_uint4( q[0], q[1], __float_as_uint( wmax ), word );
    if ( at < grid.shadowTop )
        topWords[at] = leaf ? link : recordOf( i + 1 );
    else if ( leaf && next < in.numNodes && link < grid.shadowTop )
        rec[at + 1] = make_uint4( 1u, 1u, __float_as_uint( -INFINITY ), link ); // no ray enters it; it links back into the top
}

// ---- the wide tree in the light's frame (the records ConnectLightWideKernel reads, see kernels.hip)
//
// Same nodes, same children, same indices as the wide layout (wide_bvh.h); what a record holds of a child is the rectangle
// around its geometry on the light's grid and how far it reaches along the rays' main axis - outer bounds both, joined
// upwards from the primitives: a primitive's rectangle is CellsOfFrame of its own extent, its reach the far plane of its
// leaf's box (what the third layout keeps per leaf).

struct LightBounds
{
    uint32_t x0, x1, y0, y1; // cells
    float farMax, farMin;    // largest and smallest reach of a child
};

__device__ __forceinline__ void Join( LightBounds& b, uint32_t rx, uint32_t ry, float farMax, float farMin )
{
    const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
    b.x0 = x0 < b.x0 ? x0 : b.x0; b.x1 = b.x1 < x1 ? x1 : b.x1;
    b.y0 = y0 < b.y0 ? y0 : b.y0; b.y1 = b.y1 < y1 ? y1 : b.y1;
    b.farMax = b.farMax < farMax ? farMax : b.farMax;
    b.farMin = farMin < b.farMin ? farMin : b.farMin;
}

struct WideChildren
{
    uint32_t imask, lmask, childBase, primBase;
};
__device__ __forceinline__ WideChildren ChildrenOf( const uint32_t* __restrict__ wideNodes, uint32_t n )
{
    const uint32_t* w = wideNodes + 20ull * n;
    return WideChildren{ w[3] >> 24, w[6] & 0xFFu, w[4], w[5] };
}

__device__ __forceinline__ uint32_t LoadSharedWord( const uint32_t* p ) { return __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); }

// A node's bounds: {x0 | x1 << 16, y0 | y1 << 16, farMax, farMin}.  kShared: the children's bounds were written by other
// threads of this launch (LightClimbKernel)
template < bool kShared >
__device__ __forceinline__ uint4 BoundsOfNode( const WideChildren& c, const uint2* __restrict__ primRect, const float* __restrict__ primFar, const uint32_t* nodeBounds )
{
    LightBounds b{ 0xFFFFu, 0u, 0xFFFFu, 0u, -INFINITY, INFINITY };
    for ( uint32_t k = 0, count = __popc( c.lmask ); k < count; ++k )
    {
        const uint2 r = primRect[c.primBase + k];
        const float f = primFar[c.primBase + k];
        Join( b, r.x, r.y, f, f );
    }
    for ( uint32_t k = 0, count = __popc( c.imask ); k < count; ++k )
    {
        const uint32_t* w = nodeBounds + 4ull * ( c.childBase + k );
        uint32_t rx, ry, fmax;
        if constexpr ( kShared )
        {
            rx   = LoadSharedWord( w );
            ry   = LoadSharedWord( w + 1 );
            fmax = LoadSharedWord( w + 2 );
        }
        else
        {
            rx   = w[0];
            ry   = w[1];
            fmax = w[2];
        }
        Join( b, rx, ry, __uint_as_float( fmax ), __uint_as_float( fmax ) ); // a child reaches as far as its farthest child
    }
    return make_uint4( b.x0 | b.x1 << 16, b.y0 | b.y1 << 16, __float_as_uint( b.farMax ), __float_as_uint( b.farMin ) );
}

// One thread per primitive of the wide layout's copy: word 3 of its record names it in the reference's order, words 7, 11,
// 12 - 15 are its leaf's box
__global__ __launch_bounds__( kBlock ) void LightPrimBoundsKernel( LayoutInputs in, ShadowAxes axes, ShadowGrid grid, const float4* __restrict__ widePrims,
                                                                   uint2* __restrict__ primRect, float* __restrict__ primFar )
{
    const uint32_t j = GlobalIndex();
    if ( j >= in.numPrims ) return;
    const float4 q0 = widePrims[4ull * j], q1 = widePrims[4ull * j + 1], q2 = widePrims[4ull * j + 2], q3 = widePrims[4ull * j + 3];
    double lo[2] = { INFINITY, INFINITY }, hi[2] = { -INFINITY, -INFINITY };
    JoinPrimitive( in, axes, in.prims[__float_as_uint( q0.w ) & 0x7FFFFFFFu], lo, hi );
    const float f[4] = { RoundedDown( lo[0] ), RoundedUp( hi[0] ), RoundedDown( lo[1] ), RoundedUp( hi[1] ) };
    uint32_t q[2];
    CellsOfFrame( f, grid, q );
    primRect[j] = make_uint2( ( q[0] & 0xFFFFu ) | ( ( q[0] & 0xFFFFu ) + ( q[0] >> 16 ) ) << 16, ( q[1] & 0xFFFFu ) | ( ( q[1] & 0xFFFFu ) + ( q[1] >> 16 ) ) << 16 );
    const float boxLo[3] = { q1.w, q2.w, q3.x }, boxHi[3] = { q3.y, q3.z, q3.w };
    primFar[j] = grid.towardsPositive ? boxHi[grid.depthAxis] : -boxLo[grid.depthAxis];
}

__global__ __launch_bounds__( kBlock ) void WideParentsKernel( const uint32_t* __restrict__ wideNodes, uint32_t numNodes, uint32_t* __restrict__ parent )
{
    const uint32_t n = GlobalIndex();
    if ( n >= numNodes ) return;
    if ( n == 0 ) parent[0] = kUnset;
    const WideChildren c = ChildrenOf( wideNodes, n );
    for ( uint32_t k = 0, count = __popc( c.imask ); k < count; ++k ) parent[c.childBase + k] = n;
}

// Bounds of every wide node, joined upwards: the nodes without inner children start, and whoever arrives last at a parent
// does the parent (the protocol of ShadowFramesKernel< false >; LightBoundsCheckKernel proves the result)
__global__ __launch_bounds__( kBlock ) void LightClimbKernel( const uint32_t* __restrict__ wideNodes, uint32_t numNodes, const uint32_t* __restrict__ parent,
                                                              uint32_t* __restrict__ arrivals, const uint2* __restrict__ primRect, const float* __restrict__ primFar,
                                                              uint32_t* nodeBounds )
{
    const uint32_t i = GlobalIndex();
    if ( i >= numNodes ) return;
    WideChildren c = ChildrenOf( wideNodes, i );
    if ( c.imask != 0 ) return;
    for ( uint32_t n = i;; )
    {
        const uint4 b = BoundsOfNode< true >( c, primRect, primFar, nodeBounds );
        uint32_t* mine = nodeBounds + 4ull * n;
        __hip_atomic_store( mine, b.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        __hip_atomic_store( mine + 1, b.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        __hip_atomic_store( mine + 2, b.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        __hip_atomic_store( mine + 3, b.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        const uint32_t p = parent[n];
        if ( p == kUnset ) break;
        c = ChildrenOf( wideNodes, p );
        __builtin_amdgcn_fence( __ATOMIC_RELEASE, "workgroup" ); // waits for the stores' acknowledgements
        if ( atomicAdd( arrivals + p, 1u ) + 1u != static_cast< uint32_t >( __popc( c.imask ) ) ) break; // others are still on their way
        __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "workgroup" );
        n = p;
    }
}

__global__ __launch_bounds__( kBlock ) void LightBoundsCheckKernel( const uint32_t* __restrict__ wideNodes, uint32_t numNodes, const uint2* __restrict__ primRect,
                                                                    const float* __restrict__ primFar, const uint32_t* __restrict__ nodeBounds, uint32_t* __restrict__ verdict )
{
    const uint32_t n = GlobalIndex();
    if ( n >= numNodes ) return;
    const uint4 b = BoundsOfNode< false >( ChildrenOf( wideNodes, n ), primRect, primFar, nodeBounds );
    const uint32_t* w = nodeBounds + 4ull * n;
    if ( b.x != w[0] || b.y != w[1] || b.z != w[2] || b.w != w[3] ) atomicOr( verdict, kShadowFramesWrong );
}

// The 64-byte record of a wide node in the light's frame:
//   0      first cell of the node's rectangle along u | along v << 16
//   1      shiftU | shiftV << 8 | imask << 16 | lmask << 24: a child's rectangle is kept in units of 2^shift cells from there
//   2, 3   index of the first inner child, of the first primitive (as in the wide layout)
//   4, 5   d0, 1 / step: a child reaches to d0 + byte * step along the rays' main axis, rounded up
//   6, 7   reach[8];  8, 9  first u[8];  10, 11  last u[8];  12, 13  first v[8];  14, 15  last v[8] - one byte per slot; an
//          empty slot has first = 255, last = 0
__global__ __launch_bounds__( kBlock ) void LightWideNodesKernel( const uint32_t* __restrict__ wideNodes, uint32_t numNodes, const uint2* __restrict__ primRect,
                                                                  const float* __restrict__ primFar, const uint32_t* __restrict__ nodeBounds, uint4* __restrict__ out )
{
    const uint32_t n = GlobalIndex();
    if ( n >= numNodes ) return;
    const WideChildren c = ChildrenOf( wideNodes, n );
    const uint32_t* w = nodeBounds + 4ull * n;
    const uint32_t ox = w[0] & 0xFFFFu, oy = w[1] & 0xFFFFu, ex = ( w[0] >> 16 ) - ox, ey = ( w[1] >> 16 ) - oy;
    const uint32_t shiftU = ex > 255u ? 24u - __clz( ex ) : 0u, shiftV = ey > 255u ? 24u - __clz( ey ) : 0u; // extent >> shift <= 255
    const double d0 = __uint_as_float( w[3] ), range = static_cast< double >( __uint_as_float( w[2] ) ) - d0;
    int e = 0;
    if ( range > 0.0 )
    {
        (void)frexp( range / 255.0, &e ); // range / 255 = m * 2^e with m < 1: 2^e is a step that 255 of reach over the range
        e = e < -126 ? -126 : e;
    }
    const double step = ldexp( 1.0, e );
    uint32_t bytes[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; // reach, first u, last u, first v, last v: two words each
    for ( uint32_t s = 0; s < 8; ++s )
    {
        uint32_t reach = 0, u0 = 255, u1 = 0, v0 = 255, v1 = 0;
        const uint32_t bit = 1u << s;
        if ( ( c.imask | c.lmask ) & bit )
        {
            uint32_t rx, ry;
            float far;
            if ( c.imask & bit )
            {
                const uint32_t* cw = nodeBounds + 4ull * ( c.childBase + __popc( c.imask & ( bit - 1u ) ) );
                rx = cw[0]; ry = cw[1]; far = __uint_as_float( cw[2] );
            }
            else
            {
                const uint32_t j = c.primBase + __popc( c.lmask & ( bit - 1u ) );
                rx = primRect[j].x; ry = primRect[j].y; far = primFar[j];
            }
            u0 = ( ( rx & 0xFFFFu ) - ox ) >> shiftU; u1 = ( ( rx >> 16 ) - ox ) >> shiftU;
            v0 = ( ( ry & 0xFFFFu ) - oy ) >> shiftV; v1 = ( ( ry >> 16 ) - oy ) >> shiftV;
            reach = static_cast< uint32_t >( ceil( ( static_cast< double >( far ) - d0 ) / step ) ); // <= 255 by the choice of the step
        }
        const uint32_t at = s >> 2, sh = 8u * ( s & 3u );
        bytes[at] |= reach << sh;
        bytes[2 + at] |= u0 << sh; bytes[4 + at] |= u1 << sh;
        bytes[6 + at] |= v0 << sh; bytes[8 + at] |= v1 << sh;
    }
    uint4* o = out + 4ull * n;
    o[0] = make_uint4( ox | oy << 16, shiftU | shiftV << 8 | c.imask << 16 | c.lmask << 24, c.childBase, c.primBase );
    o[1] = make_uint4( __float_as_uint( static_cast< float >( d0 ) ), __float_as_uint( static_cast< float >( 1.0 / step ) ), bytes[0], bytes[1] );
    o[2] = make_uint4( bytes[2], bytes[3], bytes[4], bytes[5] );
    o[3] = make_uint4( bytes[6], bytes[7], bytes[8], bytes[9] );
}

} // namespace

void BuildReferenceNodes( hipStream_t stream, const LayoutInputs& in, const uint32_t* topSorted, const uint32_t* topRank, uint32_t topCount, float4* out )
{
    if ( in.numNodes == 0 ) return;
    hipLaunchKernelGGL( ReferenceNodesKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, topSorted, topRank, topCount, out );
}

uint32_t LayoutScanBlocks( uint32_t numNodes ) { return numNodes / kScanTile + 1u; } // covers index numNodes too

void BuildOrderedNodes( hipStream_t stream, const LayoutInputs& in, const uint32_t* topOrder, uint32_t topCount, const uint32_t* roots, const uint32_t* rootBase,
                        uint32_t numRoots, uint32_t numOrdered, uint32_t* at, uint32_t* parent, uint32_t* innerPrefix, uint32_t* blockSums, uint4* out )
{
    if ( in.numNodes == 0 ) return;
    const uint32_t scanBlocks = LayoutScanBlocks( in.numNodes );
    hipLaunchKernelGGL( InnerCountKernel, dim3( scanBlocks ), dim3( kBlock ), 0, stream, in, blockSums );
    hipLaunchKernelGGL( ScanBlockSumsKernel, dim3( 1 ), dim3( 1 ), 0, stream, blockSums, scanBlocks );
    hipLaunchKernelGGL( InnerPrefixKernel, dim3( scanBlocks ), dim3( kBlock ), 0, stream, in, blockSums, innerPrefix );
    (void)hipMemsetAsync( at, 0xFF, 4ull * in.numNodes, stream );
    hipLaunchKernelGGL( TopRecordsKernel, dim3( BlocksFor( topCount ) ), dim3( kBlock ), 0, stream, topOrder, topCount, at );
    hipLaunchKernelGGL( PairRecordsKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, roots, rootBase, numRoots, innerPrefix, at, parent );
    hipLaunchKernelGGL( OrderedNodesKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, at, parent, numOrdered, out );
}

void BuildPrimRecords( hipStream_t stream, const LayoutInputs& in, float4* prims, float4* shade )
{
    if ( in.numPrims == 0 ) return;
    hipLaunchKernelGGL( PrimRecordsKernel, dim3( BlocksFor( in.numPrims ) ), dim3( kBlock ), 0, stream, in, prims, shade );
    hipLaunchKernelGGL( LeafBoxKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, prims );
}

void GatherWidePrims( hipStream_t stream, const LayoutInputs& in, const float4* prims, const uint32_t* primOrder, float4* widePrims )
{
    if ( in.numPrims == 0 ) return;
    hipLaunchKernelGGL( WidePrimsKernel, dim3( BlocksFor( in.numPrims ) ), dim3( kBlock ), 0, stream, in, prims, primOrder, widePrims );
}

void BuildShadowFrames( hipStream_t stream, const LayoutInputs& in, const ShadowAxes& axes, bool fullFences, uint32_t* parent, uint32_t* arrivals, float* frame,
                        uint32_t* verdict )
{
    if ( in.numNodes == 0 ) return;
    (void)hipMemsetAsync( verdict, 0, 4, stream );
    hipLaunchKernelGGL( ParentsKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, parent, verdict );
    (void)hipMemsetAsync( arrivals, 0, 4ull * in.numNodes, stream );
    if ( fullFences )
        hipLaunchKernelGGL( ShadowFramesKernel< true >, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, axes, parent, arrivals, frame );
    else
        hipLaunchKernelGGL( ShadowFramesKernel< false >, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, axes, parent, arrivals, frame );
    hipLaunchKernelGGL( FramesCheckKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, reinterpret_cast< const float4* >( frame ), verdict );
}

void CountShadowRecords( hipStream_t stream, const LayoutInputs& in, const uint32_t* topSorted, const uint32_t* topRank, uint32_t topCount, uint32_t shadowTop,
                         uint32_t* before, uint32_t* blockSums )
{
    const uint32_t count = in.numNodes + 1, tiles = count / kScanTile + 1u;
    hipLaunchKernelGGL( ShadowSlotsKernel, dim3( BlocksFor( count ) ), dim3( kBlock ), 0, stream, in, topSorted, topRank, topCount, shadowTop, before );
    hipLaunchKernelGGL( SumTilesKernel, dim3( tiles ), dim3( kBlock ), 0, stream, before, count, blockSums );
    hipLaunchKernelGGL( ScanBlockSumsKernel, dim3( 1 ), dim3( 1 ), 0, stream, blockSums, tiles );
    hipLaunchKernelGGL( PrefixTilesKernel, dim3( tiles ), dim3( kBlock ), 0, stream, before, count, blockSums );
}

void BuildShadowRecords( hipStream_t stream, const LayoutInputs& in, const float* frame, const ShadowGrid& grid, const uint32_t* topSorted, const uint32_t* topRank,
                         uint32_t topCount, const uint32_t* before, uint4* rec, uint32_t* topWords )
{
    if ( in.numNodes == 0 ) return;
    hipLaunchKernelGGL( ShadowRecordsKernel, dim3( BlocksFor( in.numNodes ) ), dim3( kBlock ), 0, stream, in, frame, grid, topSorted, topRank, topCount, before, rec, topWords );
}


void BuildLightWideNodes( hipStream_t stream, const LayoutInputs& in, const ShadowAxes& axes, const ShadowGrid& grid, const void* wideNodes, uint32_t numWideNodes,
                          const float4* widePrims, uint2* primRect, float* primFar, uint32_t* parent, uint32_t* arrivals, uint32_t* nodeBounds, uint32_t* verdict, uint4* out )
{
    if ( numWideNodes == 0 || in.numPrims == 0 ) return;
    const uint32_t* nodes = static_cast< const uint32_t* >( wideNodes );
    const dim3 perNode( BlocksFor( numWideNodes ) ), perPrim( BlocksFor( static_cast< uint32_t >( in.numPrims ) ) ), block( kBlock );
    (void)hipMemsetAsync( verdict, 0, 4, stream );
    (void)hipMemsetAsync( arrivals, 0, 4ull * numWideNodes, stream );
    hipLaunchKernelGGL( LightPrimBoundsKernel, perPrim, block, 0, stream, in, axes, grid, widePrims, primRect, primFar );
    hipLaunchKernelGGL( WideParentsKernel, perNode, block, 0, stream, nodes, numWideNodes, parent );
    hipLaunchKernelGGL( LightClimbKernel, perNode, block, 0, stream, nodes, numWideNodes, parent, arrivals, primRect, primFar, nodeBounds );
    hipLaunchKernelGGL( LightBoundsCheckKernel, perNode, block, 0, stream, nodes, numWideNodes, primRect, primFar, nodeBounds, verdict );
    hipLaunchKernelGGL( LightWideNodesKernel, perNode, block, 0, stream, nodes, numWideNodes, primRect, primFar, nodeBounds, out );
}

} // namespace ptd
