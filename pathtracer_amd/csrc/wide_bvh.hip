// Host side of the wide node layout (wide_bvh.h): collapses the reference's binary tree into nodes of up to eight
// children and quantises the child boxes.  Nothing here is on the per-ray path; it runs once per pt_upload_scene.
#include "wide_bvh.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <queue>
#include <thread>

namespace ptd
{

namespace
{

// What a slot of a wide node stands for: the subtree of an inner node of the reference's tree (count = 0), or shapes
// [first, first + count) of one of its leaves.  A leaf of several shapes is halved like an inner node until single
// shapes are left, every part keeping the leaf's box - the box the reference tests before any of them (bvh.cpp:242).
struct Item
{
    uint32_t node, first, count;
};

// The node construction below runs on the host (the top of the tree, and whole trees under PT_SCENE_HOST_LAYOUTS) and in
// a kernel (everything below the top): one source for both, double precision throughout, so that the records agree bit for bit.
#define PT_HD __host__ __device__ inline

PT_HD Item ItemOf( const PtBvhNode* nodes, uint32_t node )
{
    const PtBvhNode& n = nodes[node];
    return n.num_shapes ? Item{ node, static_cast< uint32_t >( n.offset ), n.num_shapes } : Item{ node, 0u, 0u };
}

PT_HD bool IsInner( const Item& it ) { return it.count != 1; }

PT_HD void Halve( const PtBvhNode* nodes, const Item& it, Item* a, Item* b )
{
    if ( it.count == 0 )
    {
        *a = ItemOf( nodes, it.node + 1 );
        *b = ItemOf( nodes, static_cast< uint32_t >( nodes[it.node].offset ) );
    }
    else
    {
        const uint32_t h = it.count / 2;
        *a = Item{ it.node, it.first, h };
        *b = Item{ it.node, it.first + h, it.count - h };
    }
}

PT_HD double Area( const PtBvhNode* nodes, const Item& it )
{
    const PtBvhNode& n = nodes[it.node];
    const double x = static_cast< double >( n.aabb_max[0] ) - n.aabb_min[0], y = static_cast< double >( n.aabb_max[1] ) - n.aabb_min[1],
                 z = static_cast< double >( n.aabb_max[2] ) - n.aabb_min[2];
    return x * y + y * z + z * x;
}

struct MadeNode
{
    uint32_t rec[kWideNodeWords];
    Item inner[8];
    uint32_t leafPrim[8];
    int numInner = 0, numLeaf = 0;
};

// 2^k as a double, k within the range of normal numbers
PT_HD double PowerOfTwo( int k )
{
    const unsigned long long bits = static_cast< unsigned long long >( k + 1023 ) << 52;
    double x;
    __builtin_memcpy( &x, &bits, 8 );
    return x;
}

// k with x = m * 2^k, 0.5 <= m < 1 (frexp), for a normal x > 0
PT_HD int ExponentOf( double x )
{
    unsigned long long bits;
    __builtin_memcpy( &bits, &x, 8 );
    return static_cast< int >( ( bits >> 52 ) & 0x7FFu ) - 1022;
}

// The record of the wide node that stands for `it` (words 4 and 5, the indices of its children, are the caller's),
// its inner children and its leaf children, both in slot order.  False: a box that cannot be quantised.
PT_HD bool MakeNode( const PtBvhNode* nodes, const PtPrim* prims, const uint8_t* sphereBelow, const Item& it, MadeNode* out )
{
    Item child[8];
    int n = 0;
    if ( IsInner( it ) )
    {
        Halve( nodes, it, &child[0], &child[1] );
        n = 2;
        // open the child with the largest box until eight slots are taken: the large boxes are the ones most rays
        // enter, so they are the ones worth replacing by their parts
        while ( n < 8 )
        {
            int best = -1;
            double bestArea = -1.0;
            for ( int i = 0; i < n; ++i )
                if ( IsInner( child[i] ) )
                {
                    const double a = Area( nodes, child[i] );
                    if ( a > bestArea )
                    {
                        bestArea = a;
                        best     = i;
                    }
                }
            if ( best < 0 ) break;
            Item a, b;
            Halve( nodes, child[best], &a, &b );
            child[best] = a;
            child[n++]  = b;
        }
    }
    else
        child[n++] = it;

    const PtBvhNode& box = nodes[it.node];
    // slots: greedily the (child, slot) pair whose slot direction the child's centre lies furthest along
    int slotOf[8], childAt[8];
    for ( int i = 0; i < 8; ++i ) slotOf[i] = childAt[i] = -1;
    {
        double rel[8][3];
        for ( int i = 0; i < n; ++i )
        {
            const PtBvhNode& c = nodes[child[i].node];
            for ( int a = 0; a < 3; ++a )
                rel[i][a] = ( static_cast< double >( c.aabb_min[a] ) + c.aabb_max[a] ) - ( static_cast< double >( box.aabb_min[a] ) + box.aabb_max[a] );
        }
        for ( int round = 0; round < n; ++round )
        {
            int bi = -1, bs = -1;
            double bc = 0;
            for ( int i = 0; i < n; ++i )
            {
                if ( slotOf[i] >= 0 ) continue;
                for ( int s = 0; s < 8; ++s )
                {
                    if ( childAt[s] >= 0 ) continue;
                    const double c = ( ( s & 1 ) ? rel[i][0] : -rel[i][0] ) + ( ( s & 2 ) ? rel[i][1] : -rel[i][1] ) + ( ( s & 4 ) ? rel[i][2] : -rel[i][2] );
                    if ( bi < 0 || c > bc )
                    {
                        bi = i;
                        bs = s;
                        bc = c;
                    }
                }
            }
            slotOf[bi]  = bs;
            childAt[bs] = bi;
        }
    }

    uint32_t* r = out->rec;
    for ( uint32_t k = 0; k < kWideNodeWords; ++k ) r[k] = 0u;
    uint8_t planes[6][8];
    for ( int s = 0; s < 8; ++s )
        for ( int a = 0; a < 3; ++a )
        {
            planes[a][s]     = 255;
            planes[3 + a][s] = 0;
        }
    uint32_t exps[3];
    for ( int a = 0; a < 3; ++a )
    {
        const double p = box.aabb_min[a], extent = static_cast< double >( box.aabb_max[a] ) - p;
        if ( !( extent >= 0.0 ) || !__builtin_isfinite( extent ) || !__builtin_isfinite( p ) ) return false;
        int e = 1;
        if ( extent > 0.0 )
        {
            const int k = ExponentOf( extent / 255.0 ); // extent / 255 = m * 2^k with m < 1: 255 cells of 2^k cover the box
            e = k + 127 > 1 ? k + 127 : 1;
        }
        for ( ;; ++e )
        {
            if ( e > 254 ) return false;
            const double cell = PowerOfTwo( e - 127 );
            bool fits = true;
            for ( int i = 0; i < n && fits; ++i )
            {
                const PtBvhNode& c = nodes[child[i].node];
                const double lo = c.aabb_min[a], hi = c.aabb_max[a];
                double qlo = __builtin_floor( ( lo - p ) / cell ), qhi = __builtin_ceil( ( hi - p ) / cell );
                qlo = qlo < 0.0 ? 0.0 : qlo > 255.0 ? 255.0 : qlo;
                qhi = qhi < 0.0 ? 0.0 : qhi > 255.0 ? 255.0 : qhi;
                while ( qlo > 0.0 && p + qlo * cell > lo ) qlo -= 1.0;
                while ( qhi < 255.0 && p + qhi * cell < hi ) qhi += 1.0;
                // outer bounds in exact arithmetic (all three terms fit a double): p + qlo * cell <= lo, hi <= p + qhi * cell
                if ( p + qlo * cell > lo || p + qhi * cell < hi ) fits = false;
                planes[a][slotOf[i]]     = static_cast< uint8_t >( qlo );
                planes[3 + a][slotOf[i]] = static_cast< uint8_t >( qhi );
            }
            if ( fits ) break;
        }
        exps[a] = static_cast< uint32_t >( e );
    }
    uint32_t imask = 0, lmask = 0, smask = 0;
    out->numInner = out->numLeaf = 0;
    for ( int s = 0; s < 8; ++s )
    {
        if ( childAt[s] < 0 ) continue;
        const Item& c = child[childAt[s]];
        if ( sphereBelow )
        {
            bool sphere = c.count == 0 && sphereBelow[c.node];
            for ( uint32_t k = 0; k < c.count; ++k ) sphere = sphere || prims[c.first + k].type == PT_PRIM_SPHERE;
            if ( sphere ) smask |= 1u << s;
        }
        if ( IsInner( c ) )
        {
            imask |= 1u << s;
            out->inner[out->numInner++] = c;
        }
        else
        {
            lmask |= 1u << s;
            out->leafPrim[out->numLeaf++] = c.first;
        }
    }
    __builtin_memcpy( &r[0], box.aabb_min, 12 );
    r[3] = exps[0] | exps[1] << 8 | exps[2] << 16 | imask << 24;
    r[6] = lmask;
    r[7] = smask;
    for ( int k = 0; k < 6; ++k )
        for ( int s = 0; s < 8; ++s ) r[8 + 2 * k + ( s >> 2 )] |= static_cast< uint32_t >( planes[k][s] ) << ( 8 * ( s & 3 ) );
    return true;
}

// A subtree built on its own: the records below its root with child and primitive indices relative to the subtree's
// own arrays, and the root's record (which lives in its parent's block of children, outside the subtree's arrays).
struct Subtree
{
    Item root;
    uint32_t rootIndex = 0; // where the root's record goes in the whole array
    uint32_t rootDepth = 0;
    std::vector< uint32_t > nodes, prims;
    uint32_t rootRec[kWideNodeWords];
    uint32_t depth = 0; // deepest level reached, counted from the whole tree's root = 1
    bool ok = true;
};

void BuildSubtree( const PtSceneDesc* d, const uint8_t* sphereBelow, Subtree* st )
{
    struct Open
    {
        Item it;
        int64_t at; // record index in st->nodes, -1: the root
        uint32_t depth;
    };
    std::vector< Open > stack{ Open{ st->root, -1, st->rootDepth } };
    MadeNode made;
    while ( !stack.empty() )
    {
        const Open open = stack.back();
        stack.pop_back();
        if ( !MakeNode( d->nodes, d->prims, sphereBelow, open.it, &made ) )
        {
            st->ok = false;
            return;
        }
        st->depth = std::max( st->depth, open.depth );
        const uint32_t childBase = static_cast< uint32_t >( st->nodes.size() / kWideNodeWords ), primBase = static_cast< uint32_t >( st->prims.size() );
        made.rec[4] = childBase;
        made.rec[5] = primBase;
        st->nodes.resize( st->nodes.size() + static_cast< size_t >( made.numInner ) * kWideNodeWords );
        for ( int k = 0; k < made.numLeaf; ++k ) st->prims.push_back( made.leafPrim[k] );
        std::memcpy( open.at < 0 ? st->rootRec : &st->nodes[static_cast< size_t >( open.at ) * kWideNodeWords], made.rec, sizeof( made.rec ) );
        // depth first, first child next: a subtree's records stay close together
        for ( int k = made.numInner; k-- > 0; ) stack.push_back( Open{ made.inner[k], static_cast< int64_t >( childBase ) + k, open.depth + 1 } );
    }
}

// The part of the build that stays on the host whichever way the rest is built: the check that boxes nest, and the top of
// the tree down to enough subtrees to go round (`jobs`: their roots, and where each root's record goes).
struct WidePlan
{
    std::vector< uint32_t > top, topPrims; // records placed so far (the job roots' slots still empty), their leaf primitives
    std::vector< Subtree > jobs;
    uint32_t topDepth = 0;
    unsigned threads  = 1;
};

bool PlanWideLayout( const PtSceneDesc* d, const std::vector< uint32_t >& escape, const uint8_t* sphereBelow, WidePlan& plan )
{
    const uint32_t N = static_cast< uint32_t >( d->num_nodes );
    unsigned threads = std::thread::hardware_concurrency();
    if ( threads > 16 ) threads = 16;
    if ( threads < 1 ) threads = 1;

    // The leaf test of the traversal kernels checks a primitive's own leaf box and takes the boxes of its ancestors
    // for granted, which holds when every box lies inside its parent's: true of any tree built from the shapes' boxes
    // by unions (bvh.cpp:56-59, 187-189), checked here because the caller's array is only known to be a tree.
    {
        std::atomic< bool > nested{ true };
        auto check = [&]( uint32_t begin, uint32_t end ) {
            for ( uint32_t i = begin; i < end && nested.load( std::memory_order_relaxed ); ++i )
            {
                const PtBvhNode& n = d->nodes[i];
                bool good = true;
                for ( int a = 0; a < 3; ++a ) good = good && std::isfinite( n.aabb_min[a] ) && std::isfinite( n.aabb_max[a] ) && n.aabb_min[a] <= n.aabb_max[a];
                if ( good && n.num_shapes == 0 )
                {
                    const PtBvhNode* kids[2] = { &d->nodes[i + 1], &d->nodes[static_cast< uint32_t >( n.offset )] };
                    for ( const PtBvhNode* c : kids )
                        for ( int a = 0; a < 3; ++a ) good = good && c->aabb_min[a] >= n.aabb_min[a] && c->aabb_max[a] <= n.aabb_max[a];
                }
                if ( !good ) nested.store( false );
            }
        };
        if ( N < ( 1u << 18 ) || threads < 2 )
            check( 0, N );
        else
        {
            std::vector< std::thread > pool;
            const uint32_t per = ( N + threads - 1 ) / threads;
            for ( unsigned t = 0; t < threads; ++t ) pool.emplace_back( check, std::min( N, t * per ), std::min( N, ( t + 1 ) * per ) );
            for ( std::thread& t : pool ) t.join();
        }
        if ( !nested.load() ) return false;
    }

    // The top of the tree, breadth first and largest subtree first, until there are enough subtrees to go round the
    // host threads; then every subtree on its own.  A small tree is one subtree.
    plan.threads = threads;
    std::vector< uint32_t >& top = plan.top;
    top.assign( kWideNodeWords, 0u ); // records placed so far: the root's slot
    std::vector< uint32_t >& topPrims = plan.topPrims;
    std::vector< Subtree >& jobs      = plan.jobs;
    uint32_t& topDepth                = plan.topDepth;
    {
        struct Open
        {
            uint32_t size;
            Item it;
            uint32_t at, depth;
            bool operator<( const Open& o ) const { return size < o.size; }
        };
        auto sizeOf = [&]( const Item& it ) { return it.count ? it.count : escape[it.node] - it.node; };
        std::priority_queue< Open > open;
        open.push( Open{ N, ItemOf( d->nodes, 0 ), 0u, 1u } );
        const size_t wanted = ( N >= ( 1u << 18 ) && threads > 1 ) ? 64u * threads : 1u;
        MadeNode made;
        while ( !open.empty() && open.size() < wanted && open.top().size >= 4096 )
        {
            const Open o = open.top();
            open.pop();
            if ( !MakeNode( d->nodes, d->prims, sphereBelow, o.it, &made ) ) return false;
            topDepth = std::max( topDepth, o.depth );
            const uint32_t childBase = static_cast< uint32_t >( top.size() / kWideNodeWords );
            made.rec[4] = childBase;
            made.rec[5] = static_cast< uint32_t >( topPrims.size() );
            top.resize( top.size() + static_cast< size_t >( made.numInner ) * kWideNodeWords );
            for ( int k = 0; k < made.numLeaf; ++k ) topPrims.push_back( made.leafPrim[k] );
            std::memcpy( &top[static_cast< size_t >( o.at ) * kWideNodeWords], made.rec, sizeof( made.rec ) );
            for ( int k = 0; k < made.numInner; ++k ) open.push( Open{ sizeOf( made.inner[k] ), made.inner[k], childBase + k, o.depth + 1 } );
        }
        jobs.resize( open.size() );
        for ( size_t j = 0; !open.empty(); open.pop(), ++j )
        {
            jobs[j].root      = open.top().it;
            jobs[j].rootIndex = open.top().at;
            jobs[j].rootDepth = open.top().depth;
        }
    }
    return true;
}

// ---- the subtrees below the top, built by kernels
//
// All jobs at once, level by level.  A node under construction lives in a WideTmp; where its children's WideTmps lie is
// decided by an atomic counter and does not matter: the final index of every record follows from the tree alone, in
// three sweeps - down (MakeNode per node, children handed out), up (records and leaf primitives below every node), down
// again (positions: a node's children form one block, followed by the first child's whole subtree, then the second's -
// the order in which BuildSubtree's stack places them).
struct WideTmp
{
    Item it;
    uint32_t depth, firstChild, numInner, numLeaf;
    uint32_t below, leaves;             // records below this node; leaf primitives in its subtree, its own included
    uint32_t index, blockPos, primPos;  // where its record goes, where its children's block and its leaf primitives begin
    uint32_t rec[kWideNodeWords];
    Item inner[8];
    uint32_t leafPrim[8];
};

constexpr int kWideBuildBlock = 128;

__global__ __launch_bounds__( kWideBuildBlock ) void WideMakeKernel( const PtBvhNode* __restrict__ nodes, const PtPrim* __restrict__ prims, const uint8_t* __restrict__ sphereBelow,
                                                                     WideTmp* __restrict__ tmp, uint32_t begin, uint32_t end, uint32_t capacity, uint32_t* __restrict__ counter,
                                                                     uint32_t* __restrict__ failed )
{
    const uint32_t x = begin + blockIdx.x * kWideBuildBlock + threadIdx.x;
    if ( x >= end ) return;
    MadeNode made;
    WideTmp& t = tmp[x];
    if ( !MakeNode( nodes, prims, sphereBelow, t.it, &made ) )
    {
        atomicOr( failed, 1u );
        t.numInner = t.numLeaf = 0;
        return;
    }
    t.numInner = static_cast< uint32_t >( made.numInner );
    t.numLeaf  = static_cast< uint32_t >( made.numLeaf );
    for ( uint32_t k = 0; k < kWideNodeWords; ++k ) t.rec[k] = made.rec[k];
    for ( int k = 0; k < made.numLeaf; ++k ) t.leafPrim[k] = made.leafPrim[k];
    const uint32_t first = made.numInner ? atomicAdd( counter, t.numInner ) : 0u;
    if ( first + t.numInner > capacity )
    {
        atomicOr( failed, 2u );
        t.numInner = 0;
        return;
    }
    t.firstChild = first;
    for ( int k = 0; k < made.numInner; ++k )
    {
        tmp[first + k].it    = made.inner[k];
        tmp[first + k].depth = t.depth + 1;
    }
}

__global__ __launch_bounds__( kWideBuildBlock ) void WideSumKernel( WideTmp* __restrict__ tmp, uint32_t begin, uint32_t end, uint32_t* __restrict__ deepest )
{
    const uint32_t x = begin + blockIdx.x * kWideBuildBlock + threadIdx.x;
    if ( x >= end ) return;
    WideTmp& t = tmp[x];
    uint32_t below = 0, leaves = t.numLeaf;
    for ( uint32_t k = 0; k < t.numInner; ++k )
    {
        below += 1u + tmp[t.firstChild + k].below;
        leaves += tmp[t.firstChild + k].leaves;
    }
    t.below  = below;
    t.leaves = leaves;
    atomicMax( deepest, t.depth );
}

__global__ __launch_bounds__( kWideBuildBlock ) void WidePlaceKernel( WideTmp* __restrict__ tmp, uint32_t begin, uint32_t end, uint32_t* __restrict__ outNodes,
                                                                      uint32_t* __restrict__ primOrder )
{
    const uint32_t x = begin + blockIdx.x * kWideBuildBlock + threadIdx.x;
    if ( x >= end ) return;
    const WideTmp& t = tmp[x];
    uint32_t* r = outNodes + static_cast< size_t >( t.index ) * kWideNodeWords;
    for ( uint32_t k = 0; k < kWideNodeWords; ++k ) r[k] = k == 4 ? t.blockPos : k == 5 ? t.primPos : t.rec[k];
    for ( uint32_t k = 0; k < t.numLeaf; ++k ) primOrder[t.primPos + k] = t.leafPrim[k];
    uint32_t block = t.blockPos + t.numInner, prim = t.primPos + t.numLeaf;
    for ( uint32_t k = 0; k < t.numInner; ++k )
    {
        WideTmp& c = tmp[t.firstChild + k];
        c.index    = t.blockPos + k;
        c.blockPos = block;
        c.primPos  = prim;
        block += c.below;
        prim += c.leaves;
    }
}

struct DeviceMemory
{
    void* ptr = nullptr;
    ~DeviceMemory() { if ( ptr ) (void)hipFree( ptr ); }
    hipError_t Allocate( size_t bytes ) { return hipMalloc( &ptr, bytes ? bytes : 1 ); }
    void* Release()
    {
        void* p = ptr;
        ptr     = nullptr;
        return p;
    }
};

} // namespace

bool BuildWideLayout( const PtSceneDesc* d, const std::vector< uint32_t >& escape, const uint8_t* sphereBelow, WideLayout& out )
{
    WidePlan plan;
    if ( !PlanWideLayout( d, escape, sphereBelow, plan ) ) return false;
    const unsigned threads = plan.threads;
    std::vector< uint32_t >& top = plan.top;
    std::vector< uint32_t >& topPrims = plan.topPrims;
    std::vector< Subtree >& jobs = plan.jobs;
    const uint32_t topDepth = plan.topDepth;
    {
        std::atomic< size_t > next{ 0 };
        auto work = [&] {
            for ( size_t j = next.fetch_add( 1 ); j < jobs.size(); j = next.fetch_add( 1 ) ) BuildSubtree( d, sphereBelow, &jobs[j] );
        };
        std::vector< std::thread > pool;
        for ( unsigned t = 1; t < threads && t < jobs.size(); ++t ) pool.emplace_back( work );
        work();
        for ( std::thread& t : pool ) t.join();
    }
    uint64_t numNodes = top.size() / kWideNodeWords, numPrims = topPrims.size();
    uint32_t depth = topDepth;
    std::vector< uint64_t > nodeBase( jobs.size() ), primBase( jobs.size() );
    for ( size_t j = 0; j < jobs.size(); ++j )
    {
        if ( !jobs[j].ok ) return false;
        nodeBase[j] = numNodes;
        primBase[j] = numPrims;
        numNodes += jobs[j].nodes.size() / kWideNodeWords;
        numPrims += jobs[j].prims.size();
        depth = std::max( depth, jobs[j].depth );
    }
    // the kernels address the records through a 32-bit byte offset
    if ( depth > kWideMaxDepth || numNodes * kWideNodeWords * 4ull > 0xFFFFFFF0ull || numPrims != d->num_prims ) return false;
    out.depth = depth;
    out.nodes.swap( top );
    out.nodes.resize( numNodes * kWideNodeWords );
    out.primOrder.swap( topPrims );
    out.primOrder.resize( numPrims );
    {
        std::atomic< size_t > next{ 0 };
        auto work = [&] {
            for ( size_t j = next.fetch_add( 1 ); j < jobs.size(); j = next.fetch_add( 1 ) )
            {
                Subtree& st = jobs[j];
                const uint32_t nb = static_cast< uint32_t >( nodeBase[j] ), pb = static_cast< uint32_t >( primBase[j] );
                uint32_t* root = &out.nodes[static_cast< size_t >( st.rootIndex ) * kWideNodeWords];
                std::memcpy( root, st.rootRec, sizeof( st.rootRec ) );
                root[4] += nb;
                root[5] += pb;
                uint32_t* dst = &out.nodes[static_cast< size_t >( nb ) * kWideNodeWords];
                if ( !st.nodes.empty() ) std::memcpy( dst, st.nodes.data(), st.nodes.size() * 4 );
                for ( size_t k = 0; k < st.nodes.size(); k += kWideNodeWords )
                {
                    dst[k + 4] += nb;
                    dst[k + 5] += pb;
                }
                if ( !st.prims.empty() ) std::memcpy( &out.primOrder[pb], st.prims.data(), st.prims.size() * 4 );
                std::vector< uint32_t >().swap( st.nodes );
                std::vector< uint32_t >().swap( st.prims );
            }
        };
        std::vector< std::thread > pool;
        for ( unsigned t = 1; t < threads && t < jobs.size(); ++t ) pool.emplace_back( work );
        work();
        for ( std::thread& t : pool ) t.join();
    }
    return true;
}

bool BuildWideLayoutOnDevice( const PtSceneDesc* d, const std::vector< uint32_t >& escape, const uint8_t* sphereBelow, const PtBvhNode* deviceNodes,
                              const PtPrim* devicePrims, const uint8_t* deviceSphereBelow, WideDeviceLayout& out )
{
    WidePlan plan;
    if ( !PlanWideLayout( d, escape, sphereBelow, plan ) ) return false;
    const uint32_t J = static_cast< uint32_t >( plan.jobs.size() );
    const uint64_t P = d->num_prims;
    // every record but a job's root has at least two children, so there are fewer records than primitives
    if ( P + J + 16 > 0xFFFFFFF0ull ) return false;
    const uint32_t capacity = static_cast< uint32_t >( P + J + 16 );
    DeviceMemory tmpMem, counters;
    if ( tmpMem.Allocate( sizeof( WideTmp ) * static_cast< size_t >( capacity ) ) != hipSuccess || counters.Allocate( 16 ) != hipSuccess )
    {
        (void)hipGetLastError();
        return false;
    }
    WideTmp* tmp       = static_cast< WideTmp* >( tmpMem.ptr );
    uint32_t* counter  = static_cast< uint32_t* >( counters.ptr );
    uint32_t* failed   = counter + 1;
    uint32_t* deepest  = counter + 2;
    std::vector< WideTmp > roots( J );
    for ( uint32_t j = 0; j < J; ++j )
    {
        std::memset( &roots[j], 0, sizeof( WideTmp ) );
        roots[j].it    = plan.jobs[j].root;
        roots[j].depth = plan.jobs[j].rootDepth;
        roots[j].index = plan.jobs[j].rootIndex;
    }
    const uint32_t init[4] = { J, 0u, 0u, 0u };
    bool ok = hipMemcpy( tmp, roots.data(), sizeof( WideTmp ) * J, hipMemcpyHostToDevice ) == hipSuccess &&
              hipMemcpy( counter, init, sizeof( init ), hipMemcpyHostToDevice ) == hipSuccess;
    auto blocksFor = []( uint32_t n ) { return dim3( ( n + kWideBuildBlock - 1 ) / kWideBuildBlock ); };
    // down: a level's nodes are the WideTmps handed out while the level above was made
    std::vector< std::pair< uint32_t, uint32_t > > levels;
    for ( uint32_t begin = 0, end = J; ok && begin < end; )
    {
        levels.push_back( { begin, end } );
        hipLaunchKernelGGL( WideMakeKernel, blocksFor( end - begin ), dim3( kWideBuildBlock ), 0, nullptr, deviceNodes, devicePrims, deviceSphereBelow, tmp, begin, end, capacity,
                            counter, failed );
        uint32_t state[2] = { 0u, 1u };
        ok = hipMemcpy( state, counter, sizeof( state ), hipMemcpyDeviceToHost ) == hipSuccess && state[1] == 0 && levels.size() <= kWideMaxDepth + 1;
        begin = end;
        end   = state[0];
    }
    // up
    for ( size_t l = levels.size(); ok && l-- > 0; )
        hipLaunchKernelGGL( WideSumKernel, blocksFor( levels[l].second - levels[l].first ), dim3( kWideBuildBlock ), 0, nullptr, tmp, levels[l].first, levels[l].second, deepest );
    ok = ok && hipMemcpy( roots.data(), tmp, sizeof( WideTmp ) * J, hipMemcpyDeviceToHost ) == hipSuccess;
    uint32_t depth = plan.topDepth, deviceDepth = 0;
    ok = ok && hipMemcpy( &deviceDepth, deepest, 4, hipMemcpyDeviceToHost ) == hipSuccess;
    if ( !ok )
    {
        (void)hipGetLastError();
        return false;
    }
    depth = std::max( depth, deviceDepth );
    uint64_t numNodes = plan.top.size() / kWideNodeWords, numPrims = plan.topPrims.size();
    for ( uint32_t j = 0; j < J; ++j )
    {
        roots[j].blockPos = static_cast< uint32_t >( numNodes );
        roots[j].primPos  = static_cast< uint32_t >( numPrims );
        numNodes += roots[j].below;
        numPrims += roots[j].leaves;
    }
    // the kernels address the records through a 32-bit byte offset
    if ( depth > kWideMaxDepth || numNodes * kWideNodeWords * 4ull > 0xFFFFFFF0ull || numPrims != P ) return false;
    DeviceMemory nodesMem, orderMem;
    ok = nodesMem.Allocate( numNodes * kWideNodeWords * 4ull ) == hipSuccess && orderMem.Allocate( 4ull * P ) == hipSuccess &&
         hipMemcpy( nodesMem.ptr, plan.top.data(), plan.top.size() * 4, hipMemcpyHostToDevice ) == hipSuccess &&
         hipMemcpy( orderMem.ptr, plan.topPrims.data(), plan.topPrims.size() * 4, hipMemcpyHostToDevice ) == hipSuccess &&
         hipMemcpy( tmp, roots.data(), sizeof( WideTmp ) * J, hipMemcpyHostToDevice ) == hipSuccess;
    // down again: positions, and the records written where they belong
    for ( size_t l = 0; ok && l < levels.size(); ++l )
        hipLaunchKernelGGL( WidePlaceKernel, blocksFor( levels[l].second - levels[l].first ), dim3( kWideBuildBlock ), 0, nullptr, tmp, levels[l].first, levels[l].second,
                            static_cast< uint32_t* >( nodesMem.ptr ), static_cast< uint32_t* >( orderMem.ptr ) );
    ok = ok && hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if ( !ok )
    {
        (void)hipGetLastError();
        return false;
    }
    out.numNodes  = static_cast< uint32_t >( numNodes );
    out.depth     = depth;
    out.nodeBytes = numNodes * kWideNodeWords * 4ull;
    out.nodes     = nodesMem.Release();
    out.primOrder = static_cast< uint32_t* >( orderMem.Release() );
    return true;
}

} // namespace ptd
