// Reference-side binding of pt_build_bvh (include/pt_api.h): what a maintainer of the reference puts in place of the
// body of BVH::Build (src/bvh.cpp:204-228) to have the SAH tree built on the GPU.  Compiled against the reference's
// own headers (oracle/Makefile.ref) and run by the tests on PT::BVH objects; copy it to src/pt_bvh_binding.hpp.
//
//     void BVH::Build( std::vector< std::shared_ptr< Shape > >& listOfShapes )
//     {
//         if ( splitMethod == SplitMethod::SAH && PT::BuildOnGpu( *this, listOfShapes, pt_build_bvh, 0 ) ) return;
//         ... the reference's body, for the other split methods ...
//     }
#pragma once

#include "bvh.hpp"
#include "pt_api.h"
#include <cstring>
#include <memory>
#include <vector>

namespace PT
{

using PtBuildBvhFn = int ( * )( int32_t device, const float* boxes, uint64_t n, PtBvhNode* nodes_out, uint32_t* order_out,
                                uint64_t* num_nodes_out, float* device_ms_out );

static_assert( sizeof( LinearBVHNode ) == sizeof( PtBvhNode ), "PtBvhNode is LinearBVHNode byte for byte" );

// Fills bvh.shapes and bvh.nodes as BVH::Build does (src/bvh.cpp:206-227): the boxes Shape::WorldSpaceAABB gives,
// in list order, go to the builder; the node array it returns is the flattened tree, its order the new shape list.
// Returns false (bvh untouched but for the moved list, as in the reference) if the builder fails: pt_last_error says why.
inline bool BuildOnGpu( BVH& bvh, std::vector< std::shared_ptr< Shape > >& listOfShapes, PtBuildBvhFn build, int device )
{
    std::vector< std::shared_ptr< Shape > > list = std::move( listOfShapes ); // src/bvh.cpp:206
    const size_t n = list.size();
    std::vector< float > boxes( 6 * n );
    for ( size_t i = 0; i < n; ++i ) // src/bvh.cpp:210-215
    {
        const AABB box = list[i]->WorldSpaceAABB();
        std::memcpy( &boxes[6 * i], &box.min, 12 );
        std::memcpy( &boxes[6 * i + 3], &box.max, 12 );
    }
    std::vector< uint32_t > order( n );
    LinearBVHNode* nodes = new LinearBVHNode[2 * n - 1]; // src/bvh.cpp:224; every leaf of the reference's trees holds one shape
    uint64_t count = 0;
    if ( build( device, boxes.data(), n, reinterpret_cast< PtBvhNode* >( nodes ), order.data(), &count, nullptr ) != 0 || count != 2 * n - 1 )
    {
        delete[] nodes;
        bvh.shapes = std::move( list );
        return false;
    }
    bvh.shapes.resize( n );
    for ( size_t k = 0; k < n; ++k ) bvh.shapes[k] = list[order[k]]; // src/bvh.cpp:217-221
    delete[] bvh.nodes;
    bvh.nodes = nodes;
    return true;
}

} // namespace PT
