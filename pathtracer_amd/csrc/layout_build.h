// The per-node and per-primitive records of an uploaded scene, written by kernels from the scene description as it is
// (the reference's LinearBVHNode array, src/bvh.hpp:10-21, its shape list and the meshes' vertex arrays) instead of on
// the host: the description crosses PCIe once, at about half the size of the records, and no host thread touches a
// record.  Every function produces, bit for bit, what the host code in api.hip (PT_SCENE_HOST_LAYOUTS) produces.
#pragma once

#include "pt_api.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd
{

// Device copies of the parts of a PtSceneDesc the records are made from
struct LayoutInputs
{
    const PtBvhNode* nodes = nullptr;
    uint32_t numNodes      = 0;
    const uint32_t* escape = nullptr; // first node after the subtree of node i in the reference's pre-order
    const PtPrim* prims    = nullptr;
    uint64_t numPrims      = 0;
    const PtTriangle* triangles = nullptr;
    const PtMesh* meshes   = nullptr;
    const float *vertices = nullptr, *normals = nullptr, *tangents = nullptr, *uvs = nullptr;
    const PtSphere* spheres    = nullptr;
    const uint8_t* sphereBelow = nullptr; // per node: its subtree holds a sphere; null: the scene has none
};

// First layout: 32-byte records in device order.  topSorted / topRank: the nodes of the top of the tree in ascending
// order of their reference index, and the position of each in device order.
void BuildReferenceNodes( hipStream_t stream, const LayoutInputs& in, const uint32_t* topSorted, const uint32_t* topRank, uint32_t topCount, float4* out );

// Second layout: 64-byte records with a link per octant.  topOrder: the top of the tree in device order (the root, then
// sibling pairs); roots / rootBase: the inner nodes of the top whose children lie below it, ascending, and the record
// index of each one's first pair.  at, parent ( numNodes words each ), innerPrefix ( numNodes + 1 words ) and blockSums
// ( LayoutScanBlocks( numNodes ) + 1 words ) are scratch.
uint32_t LayoutScanBlocks( uint32_t numNodes );
void BuildOrderedNodes( hipStream_t stream, const LayoutInputs& in, const uint32_t* topOrder, uint32_t topCount, const uint32_t* roots, const uint32_t* rootBase,
                        uint32_t numRoots, uint32_t numOrdered, uint32_t* at, uint32_t* parent, uint32_t* innerPrefix, uint32_t* blockSums, uint4* out );

// The 64-byte intersection records and the 128-byte attribute records of the primitives, reference order
void BuildPrimRecords( hipStream_t stream, const LayoutInputs& in, float4* prims, float4* shade );

// The intersection records once more in the order of the wide layout, the primitive's reference index in word 3
void GatherWidePrims( hipStream_t stream, const LayoutInputs& in, const float4* prims, const uint32_t* primOrder, float4* widePrims );

// Third layout (one directional light): bounds of every node in the light's frame, then 16-byte records on a 16-bit grid.
struct ShadowAxes
{
    float u[3], v[3]; // the two axes of the light's frame across the rays
};
struct ShadowGrid
{
    double origin[2], cell[2], slack[2]; // the grid the rectangles are rounded to, and what they are widened by first (in cells)
    uint32_t depthAxis, towardsPositive; // the world axis the rays mostly travel along, and in which sense
    uint32_t shadowTop, numRecords;      // records of the top of the tree, all records
};
// parent, arrivals: numNodes words of scratch each; frame: 4 floats per node {u0, u1, v0, v1}, 16-byte aligned; verdict: one
// word, 0 afterwards or a sum of the bits below.  fullFences: agent-scope fences in the climb (slow; see ShadowFramesKernel)
constexpr uint32_t kShadowLeafTooLarge = 1u; // a leaf holds more than 8 shapes: the records have no room for that count
constexpr uint32_t kShadowFramesWrong  = 2u; // an inner node's bounds are not its children's joined: the bounds must not be used
void BuildShadowFrames( hipStream_t stream, const LayoutInputs& in, const ShadowAxes& axes, bool fullFences, uint32_t* parent, uint32_t* arrivals, float* frame,
                        uint32_t* verdict );
// before[i] (numNodes + 1 words) <- records that nodes with a smaller reference index take behind the top; before[numNodes] is
// their total.  blockSums: LayoutScanBlocks( numNodes ) + 1 words of scratch
void CountShadowRecords( hipStream_t stream, const LayoutInputs& in, const uint32_t* topSorted, const uint32_t* topRank, uint32_t topCount, uint32_t shadowTop,
                         uint32_t* before, uint32_t* blockSums );
void BuildShadowRecords( hipStream_t stream, const LayoutInputs& in, const float* frame, const ShadowGrid& grid, const uint32_t* topSorted, const uint32_t* topRank,
                         uint32_t topCount, const uint32_t* before, uint4* rec, uint32_t* topWords );

// The wide layout once more for the shadow rays of the one directional light (ConnectLightWideKernel in kernels.hip):
// 64-byte records, one per wide node and with the same indices, of the children's rectangles on the light's grid and
// their reach along the rays' main axis (see LightWideNodesKernel).  primRect ( 8 bytes ), primFar ( 4 bytes ) per
// primitive, parent, arrivals ( 4 bytes ) and nodeBounds ( 16 bytes ) per wide node are scratch; verdict: one word, 0
// afterwards, or kShadowFramesWrong: the records must not be used.
void BuildLightWideNodes( hipStream_t stream, const LayoutInputs& in, const ShadowAxes& axes, const ShadowGrid& grid, const void* wideNodes, uint32_t numWideNodes,
                          const float4* widePrims, uint2* primRect, float* primFar, uint32_t* parent, uint32_t* arrivals, uint32_t* nodeBounds, uint32_t* verdict, uint4* out );

} // namespace ptd
