// Host side of the wide node layout (wide_bvh.h): collapses the reference's binary tree into nodes of up to eight
// children and quantises the child boxes.  Nothing here is on the per-ray path; it runs once per pt_upload_scene.
#include "wide_bvh.h"

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

Item ItemOf( const PtSceneDesc* d, uint32_t node )
{
    const PtBvhNode& n = d->nodes[node];
    return n.num_shapes ? Item{ node, static_cast< uint32_t >( n.offset ), n.num_shapes } : Item{ node, 0u, 0u };
}

bool IsInner( const Item& it ) { return it.count != 1; }

void Halve( const PtSceneDesc* d, const Item& it, Item* a, Item* b )
{
    if ( it.count == 0 )
    {
        *a = ItemOf( d, it.node + 1 );
        *b = ItemOf( d, static_cast< uint32_t >( d->nodes[it.node].offset ) );
    }
    else
    {
        const uint32_t h = it.count / 2;
        *a = Item{ it.node, it.first, h };
        *b = Item{ it.node, it.first + h, it.count - h };
    }
}

double Area( const PtSceneDesc* d, const Item& it )
{
    const PtBvhNode& n = d->nodes[it.node];
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

// The record of the wide node that stands for `it` (words 4 and 5, the indices of its children, are the caller's),
// its inner children and its leaf children, both in slot order.  False: a box that cannot be quantised.
bool MakeNode( const PtSceneDesc* d, const uint8_t* sphereBelow, const Item& it, MadeNode* out )
{
    Item child[8];
    int n = 0;
    if ( IsInner( it ) )
    {
        Halve( d, it, &child[0], &child[1] );
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
                    const double a = Area( d, child[i] );
                    if ( a > bestArea )
                    {
                        bestArea = a;
                        best     = i;
                    }
                }
            if ( best < 0 ) break;
            Item a, b;
            Halve( d, child[best], &a, &b );
            child[best] = a;
            child[n++]  = b;
        }
    }
    else
        child[n++] = it;

    const PtBvhNode& box = d->nodes[it.node];
    // slots: greedily the (child, slot) pair whose slot direction the child's centre lies furthest along
    int slotOf[8], childAt[8];
    for ( int i = 0; i < 8; ++i ) slotOf[i] = childAt[i] = -1;
    {
        double rel[8][3];
        for ( int i = 0; i < n; ++i )
        {
            const PtBvhNode& c = d->nodes[child[i].node];
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
    std::memset( r, 0, sizeof( out->rec ) );
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
        if ( !( extent >= 0.0 ) || !std::isfinite( extent ) || !std::isfinite( p ) ) return false;
        int e = 1;
        if ( extent > 0.0 )
        {
            int k;
            (void)std::frexp( extent / 255.0, &k ); // extent / 255 = m * 2^k with m < 1: 255 cells of 2^k cover the box
            e = std::max( 1, k + 127 );
        }
        for ( ;; ++e )
        {
            if ( e > 254 ) return false;
            const double cell = std::ldexp( 1.0, e - 127 );
            bool fits = true;
            for ( int i = 0; i < n && fits; ++i )
            {
                const PtBvhNode& c = d->nodes[child[i].node];
                const double lo = c.aabb_min[a], hi = c.aabb_max[a];
                double qlo = std::floor( ( lo - p ) / cell ), qhi = std::ceil( ( hi - p ) / cell );
                qlo = std::min( 255.0, std::max( 0.0, qlo ) );
                qhi = std::min( 255.0, std::max( 0.0, qhi ) );
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
            for ( uint32_t k = 0; k < c.count; ++k ) sphere = sphere || d->prims[c.first + k].type == PT_PRIM_SPHERE;
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
    std::memcpy( &r[0], box.aabb_min, 12 );
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
        if ( !MakeNode( d, sphereBelow, open.it, &made ) )
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

} // namespace

bool BuildWideLayout( const PtSceneDesc* d, const std::vector< uint32_t >& escape, const uint8_t* sphereBelow, WideLayout& out )
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
    std::vector< uint32_t > top( kWideNodeWords ); // records placed so far: the root's slot
    std::vector< uint32_t > topPrims;
    std::vector< Subtree > jobs;
    uint32_t topDepth = 0;
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
        open.push( Open{ N, ItemOf( d, 0 ), 0u, 1u } );
        const size_t wanted = ( N >= ( 1u << 18 ) && threads > 1 ) ? 64u * threads : 1u;
        MadeNode made;
        while ( !open.empty() && open.size() < wanted && open.top().size >= 4096 )
        {
            const Open o = open.top();
            open.pop();
            if ( !MakeNode( d, sphereBelow, o.it, &made ) ) return false;
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

} // namespace ptd
