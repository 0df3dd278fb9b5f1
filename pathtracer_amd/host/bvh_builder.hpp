// Host BVH builder: produces the flattened 32-byte node array and the shape order the integrator consumes.
#pragma once

#include "hmath.hpp"
#include "pt_api.h"
#include <cstdint>
#include <vector>

namespace pth
{

enum class SplitMethod
{
    SAH,
    Middle,
    EqualCounts
}; // src/bvh.hpp:26

struct Aabb
{
    vec3 min = vec3( 3.402823466e+38f );  // src/aabb.cpp:6-10
    vec3 max = vec3( -3.402823466e+38f );
    void Union( const Aabb& b ) { min = vmin( min, b.min ); max = vmax( max, b.max ); } // src/aabb.cpp:18-22
    void Union( vec3 p ) { min = vmin( min, p ); max = vmax( max, p ); }               // src/aabb.cpp:24-28
    vec3 Centroid() const { return 0.5f * ( min + max ); }                            // src/aabb.cpp:30-33
    int LongestDimension() const;                                                     // src/aabb.cpp:35-50
    float SurfaceArea() const;                                                        // src/aabb.cpp:52-56
    vec3 Offset( vec3 p ) const;                                                      // src/aabb.cpp:58-66
};

struct BvhResult
{
    std::vector< PtBvhNode > nodes; // depth-first pre-order, first child = index + 1
    std::vector< uint32_t > order;  // order[k] = input index of the shape at ordered position k
};

// Builds the tree over `bounds` (one box per shape, in scene order) exactly as BVH::Build does
// (src/bvh.cpp:204-228): same split decisions, same std::partition / std::nth_element calls, same
// flattening.  A range whose centroids are all equal is halved where it lies (the reference's recursion never returns from one, src/bvh.cpp:173-175).
BvhResult BuildBvh( const std::vector< Aabb >& bounds, SplitMethod method );

// A builder with the interface of pt_build_bvh (pt_api.h) that takes over the SAH trees of HostScene::BuildAccel,
// or nullptr for BuildBvh above.
using ExternalBvhBuilder = int ( * )( int32_t device, const float* boxes, uint64_t n, PtBvhNode* nodesOut, uint32_t* orderOut,
                                      uint64_t* numNodesOut, float* deviceMsOut );
void SetExternalBvhBuilder( ExternalBvhBuilder builder, int32_t device );
// BuildBvh, or the external builder if one is set and the method is SAH; throws std::runtime_error if that one fails.
BvhResult BuildSceneBvh( const std::vector< Aabb >& bounds, SplitMethod method );

} // namespace pth
