#include "bvh_builder.hpp"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>

namespace pth
{

int Aabb::LongestDimension() const
{
    vec3 d = max - min;
    if ( d[0] > d[1] && d[0] > d[2] ) return 0;
    return d[1] > d[2] ? 1 : 2;
}

float Aabb::SurfaceArea() const
{
    vec3 d = max - min;
    return 2 * ( d.x * d.y + d.x * d.z + d.y * d.z );
}

vec3 Aabb::Offset( vec3 p ) const
{
    vec3 d   = max - min;
    vec3 rel = p - min;
    if ( max.x > min.x ) rel.x /= d.x;
    if ( max.y > min.y ) rel.y /= d.y;
    if ( max.z > min.z ) rel.z /= d.z;
    return rel;
}

namespace
{

    struct Item
    {
        Aabb box;
        vec3 centroid;
        uint32_t shape;
    };

    constexpr int kBuckets        = 12; // src/bvh.cpp:102
    constexpr int kSmallRangeSize = 4;  // src/bvh.cpp:95,150

    // Ranges of at least this many shapes hand their first half to another thread while one is free.
    constexpr int kParallelRange = 1 << 15;

    // Emits the subtree over items[start, end) straight into the pre-order array - the layout
    // FlattenBVHBuild produces (src/bvh.cpp:180-202) without the intermediate pointer tree.  Every leaf holds
    // exactly one shape, so the subtree over n shapes has 2n - 1 nodes: a node's slot is known before its
    // subtree is built, its first child takes the next slot and its second child the slot 2 * (cutoff - start)
    // further on; leaf i of the tree is items[i].  That makes the two halves of a range independent, and they
    // are built on different threads with the same result as the sequential recursion.
    struct Builder
    {
        std::vector< Item >& items;
        SplitMethod method;
        BvhResult& out;
        std::atomic< int > freeThreads;

        int BucketOf( const Aabb& centroidBox, const Item& it, int dim ) const
        {
            return std::min( kBuckets - 1, static_cast< int >( kBuckets * centroidBox.Offset( it.centroid )[dim] ) );
        }

        Item* SplitEqualCounts( int start, int end, int dim )
        {
            Item* mid = &items[( start + end ) / 2];
            std::nth_element( &items[start], mid, &items[0] + end,
                              [dim]( const Item& a, const Item& b ) { return a.centroid[dim] < b.centroid[dim]; } );
            return mid;
        }

        // Bucketed surface-area split for ranges of more than four shapes (src/bvh.cpp:100-169).  Such a
        // range is always split: the reference's leaf branch needs numShapes <= 4, which never reaches it.
        Item* SplitSah( int start, int end, int dim, const Aabb& nodeBox, const Aabb& centroidBox )
        {
            Aabb bucketBox[kBuckets];
            int bucketCount[kBuckets] = {};
            for ( int i = start; i < end; ++i )
            {
                int b = BucketOf( centroidBox, items[i], dim );
                bucketCount[b]++;
                bucketBox[b].Union( items[i].box );
            }

            float cost[kBuckets - 1];
            for ( int i = 0; i < kBuckets - 1; ++i )
            {
                Aabb b0, b1;
                int count0 = 0, count1 = 0;
                for ( int j = 0; j <= i; ++j )
                {
                    b0.Union( bucketBox[j] );
                    count0 += bucketCount[j];
                }
                for ( int j = i + 1; j < kBuckets; ++j )
                {
                    b1.Union( bucketBox[j] );
                    count1 += bucketCount[j];
                }
                cost[i] = 0.5f + ( count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea() ) / nodeBox.SurfaceArea();
            }

            float minCost = cost[0];
            int minBucket = 0;
            for ( int i = 1; i < kBuckets - 1; ++i )
            {
                if ( cost[i] < minCost )
                {
                    minCost   = cost[i];
                    minBucket = i;
                }
            }

            return std::partition( &items[start], &items[0] + end,
                                   [&]( const Item& it ) { return BucketOf( centroidBox, it, dim ) <= minBucket; } );
        }

        void Emit( int start, int end, size_t slot )
        {
            Aabb box;
            for ( int i = start; i < end; ++i ) box.Union( items[i].box );

            PtBvhNode node{};
            for ( int k = 0; k < 3; ++k )
            {
                node.aabb_min[k] = box.min[k];
                node.aabb_max[k] = box.max[k];
            }

            const int count = end - start;
            if ( count == 1 ) // src/bvh.cpp:47-56
            {
                node.offset     = start;
                node.num_shapes = 1;
                out.order[start] = items[start].shape;
                out.nodes[slot]  = node;
                return;
            }

            Aabb centroidBox;
            for ( int i = start; i < end; ++i ) centroidBox.Union( items[i].centroid );
            const int dim = centroidBox.LongestDimension();
            node.axis     = static_cast< uint8_t >( dim );

            Item* mid = nullptr;
            switch ( method )
            {
                case SplitMethod::Middle: // src/bvh.cpp:74-84, falls through to EqualCounts if one side is empty
                {
                    float split = ( centroidBox.min[dim] + centroidBox.max[dim] ) / 2;
                    mid = std::partition( &items[start], &items[0] + end,
                                          [dim, split]( const Item& it ) { return it.centroid[dim] < split; } );
                    if ( mid != &items[0] + end && mid != &items[start] ) break;
                    mid = SplitEqualCounts( start, end, dim );
                    break;
                }
                case SplitMethod::EqualCounts: // src/bvh.cpp:85-90
                    mid = SplitEqualCounts( start, end, dim );
                    break;
                case SplitMethod::SAH: // src/bvh.cpp:91-170
                default:
                    mid = count <= kSmallRangeSize ? SplitEqualCounts( start, end, dim )
                                                   : SplitSah( start, end, dim, box, centroidBox );
                    break;
            }

            int cutoff = start + static_cast< int >( mid - &items[start] );
            if ( cutoff == start || cutoff == end )
            {
                // All centroids coincide along the split axis.  The reference recurses without end here
                // (src/bvh.cpp:173-175 with midShape == endShape); split the range in half instead.
                cutoff = ( start + end ) / 2;
            }

            const size_t second = slot + 2 * static_cast< size_t >( cutoff - start );
            node.offset         = static_cast< int32_t >( second );
            out.nodes[slot]     = node;
            if ( count >= kParallelRange )
            {
                std::thread firstHalf;
                if ( freeThreads.fetch_sub( 1 ) > 0 )
                {
                    try
                    {
                        firstHalf = std::thread( [=]() { Emit( start, cutoff, slot + 1 ); } );
                    }
                    catch ( const std::system_error& )
                    {
                        // no thread to be had: this one builds both halves
                    }
                }
                if ( firstHalf.joinable() )
                {
                    Emit( cutoff, end, second );
                    firstHalf.join();
                    freeThreads.fetch_add( 1 );
                    return;
                }
                freeThreads.fetch_add( 1 );
            }
            Emit( start, cutoff, slot + 1 );
            Emit( cutoff, end, second );
        }
    };

} // namespace

BvhResult BuildBvh( const std::vector< Aabb >& bounds, SplitMethod method )
{
    if ( bounds.empty() ) throw std::runtime_error( "BuildBvh: scene has no shapes" );
    if ( bounds.size() > 0x7fffffffu / 2 ) throw std::runtime_error( "BuildBvh: too many shapes for 32-bit node offsets" );

    std::vector< Item > items( bounds.size() );
    for ( size_t i = 0; i < bounds.size(); ++i ) // src/bvh.cpp:209-215
    {
        items[i].box      = bounds[i];
        items[i].centroid = bounds[i].Centroid();
        items[i].shape    = static_cast< uint32_t >( i );
    }

    BvhResult out;
    out.nodes.resize( 2 * bounds.size() - 1 );
    out.order.resize( bounds.size() );
    int threads = static_cast< int >( std::min( 16u, std::max( 1u, std::thread::hardware_concurrency() ) ) );
    if ( const char* env = std::getenv( "PT_BUILD_THREADS" ) ) threads = std::max( 1, std::atoi( env ) );
    Builder b{ items, method, out, { threads - 1 } };
    b.Emit( 0, static_cast< int >( bounds.size() ), 0 );
    return out;
}

namespace
{
    std::atomic< ExternalBvhBuilder > g_externalBuilder{ nullptr };
    std::atomic< int32_t > g_externalDevice{ 0 };
} // namespace

void SetExternalBvhBuilder( ExternalBvhBuilder builder, int32_t device )
{
    g_externalDevice  = device;
    g_externalBuilder = builder;
}

BvhResult BuildSceneBvh( const std::vector< Aabb >& bounds, SplitMethod method )
{
    const ExternalBvhBuilder external = g_externalBuilder;
    if ( !external || method != SplitMethod::SAH ) return BuildBvh( bounds, method );
    if ( bounds.empty() ) throw std::runtime_error( "BuildBvh: scene has no shapes" );
    static_assert( sizeof( Aabb ) == 24, "a box is six floats: min xyz, max xyz" );
    BvhResult out;
    out.nodes.resize( 2 * bounds.size() - 1 );
    out.order.resize( bounds.size() );
    uint64_t count = 0;
    const int rc   = external( g_externalDevice, &bounds[0].min.x, bounds.size(), out.nodes.data(), out.order.data(), &count, nullptr );
    if ( rc != 0 || count != out.nodes.size() ) throw std::runtime_error( "the external BVH builder failed (status " + std::to_string( rc ) + ")" );
    return out;
}

} // namespace pth
