// The fourth node layout: the reference's binary tree (src/bvh.cpp:180-202) collapsed into nodes of up to eight
// children whose boxes are stored as 8-bit offsets on a grid over the node's own box, rounded outwards.
//
//   node (80 bytes = 20 words, read with five 16-byte loads):
//     0..2   p: the low corner of the node's box (float)
//     3      ex | ey << 8 | ez << 16 | imask << 24   e*: biased binary exponent of the grid cell along an axis
//                                                   (cell = 2^(e - 127)), imask: slots that hold an inner node
//     4      index of the node's first inner child; the inner children of a node are consecutive, in slot order
//     5      index (into the wide primitive array) of the node's first leaf child; consecutive, in slot order
//     6      lmask: slots that hold a primitive
//     7      smask: slots whose subtree holds a sphere
//     8..19  child planes, one byte per slot: lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8];
//            plane = p + byte * cell.  An empty slot has lo = 255, hi = 0.
//   A child's slot says where it lies in its parent: bit a of the slot is set if the child's centre lies on the
//   high side of the parent's centre along axis a (as far as the slots go round), so that a ray whose direction
//   signs form the octant `oct` meets the children roughly front to back in ascending order of slot ^ oct.
//   Every leaf child is ONE primitive: a reference leaf of several shapes becomes inner nodes over its shapes, each
//   with the leaf's box.
//   primitives: the 64-byte records of the reference-order array in the order the nodes refer to them, word 3
//   replaced by the primitive's index in the reference's order (bit 31: sphere).
//
// The boxes are outer bounds, the records keep nothing of the reference's visit order: what a traversal over them
// finds is a candidate, and the kernels decide with the reference's own arithmetic (the primitive's leaf box from its
// record, then the primitive test) what counts - see kernels.hip.
#pragma once

#include "pt_api.h"

#include <stdint.h>
#include <vector>

namespace ptd
{

constexpr uint32_t kWideNodeWords = 20;
constexpr uint32_t kWideMaxDepth  = 56; // levels of wide nodes a traversal stack is sized for

struct WideLayout
{
    std::vector< uint32_t > nodes;     // kWideNodeWords per node, node 0 is the root
    std::vector< uint32_t > primOrder; // wide primitive index -> index in the reference's order
    uint32_t depth = 0;                // levels of wide nodes
};

// False if the tree cannot be expressed (a bound that is not finite, deeper than kWideMaxDepth): the scene then has
// no wide layout.  `escape[i]` = first node after the subtree of node i in the reference's pre-order.
// sphereBelow: per node of the reference's tree, whether its subtree holds a sphere; null: the scene has none.
bool BuildWideLayout( const PtSceneDesc* d, const std::vector< uint32_t >& escape, const uint8_t* sphereBelow, WideLayout& out );

// The same layout with everything below the top of the tree built by kernels, from device copies of the node array, the
// shape list and sphereBelow: the same records and the same primitive order, left in device memory (hipMalloc; the
// caller frees both).  False as above, or if the device runs out of memory: the caller then builds on the host.
struct WideDeviceLayout
{
    void* nodes         = nullptr; // numNodes * kWideNodeWords words
    uint32_t* primOrder = nullptr; // num_prims words
    uint64_t nodeBytes  = 0;
    uint32_t numNodes = 0, depth = 0;
};
bool BuildWideLayoutOnDevice( const PtSceneDesc* d, const std::vector< uint32_t >& escape, const uint8_t* sphereBelow, const PtBvhNode* deviceNodes,
                              const PtPrim* devicePrims, const uint8_t* deviceSphereBelow, WideDeviceLayout& out );

} // namespace ptd
