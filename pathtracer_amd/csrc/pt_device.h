// Device-side data layout and launch interface of the path-tracing core (gfx950).
//
// All scene data is rewritten at upload into layouts chosen for 16-byte coalescable loads:
//   nodes      2 x float4 per node (32 B): {lo.x, lo.y, hi.x, hi.y} {lo.z, hi.z, link, word}; link = node to go to after
//              this node's subtree (top bit: leaf), word = leaf descriptor or first child.  The first numTopNodes
//              nodes are the top of the tree (sibling pairs, largest parent box first), the rest follows in the reference's pre-order
//   prims      4 x float4 per primitive (64 B = one cache line): triangle {v0, tag 0} {v1 - v0, lo.x} {v2 - v0, lo.y}
//              {lo.z, hi.xyz}; sphere {centre, tag 1} {radius, sphere index, -, lo.x} {-, -, -, lo.y} {lo.z, hi.xyz};
//              lo / hi = the box of the BVH leaf that holds the primitive
//   shadowNodes  1 x float4 per record, for the shadow rays of a scene lit by one directional light (see
//              ConnectParallelKernel in kernels.hip), or absent
//   wideNodes  5 x float4 per node of the tree collapsed to eight children with quantised boxes, widePrims the
//              primitive records in that tree's order (wide_bvh.h)
//   triShade   8 x float4 per triangle primitive (112 B used, 128 B apart = one L2 line): n0 n1 n2 t0 t1 t2 uv0 uv1 uv2 material
//   spheres    6 x float4: world-to-local columns 0..3, {centre, radius}, {material}
//   materials  2 x float4: {albedo, texture index} {Ke}
//   lights     9 x float4, see DevLight
// Path state lives in structure-of-float4 arrays indexed by path id (see PathBuffers).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd
{

constexpr uint32_t kInnerNode    = 0xFFFFFFFFu; // "no leaf pending" in the traversal state; never a valid leaf word
constexpr uint32_t kLeafFlag     = 0x80000000u; // top bit of a node's link word
constexpr uint32_t kSphereBelowFlag = 0x10u;    // front-to-back layout, second word of a record: the subtree holds a sphere
constexpr uint32_t kTopNodes     = 4095;          // default size of the top of the tree that is stored first
constexpr uint32_t kLeafFirstMask = 0x0FFFFFFFu; // leaf word: low 28 bits first primitive, high 4 bits count - 1
constexpr int kMaxLeafShapes     = 16;
constexpr uint32_t kShadowTopNodes = 3840;       // records at the front of shadowNodes that the kernel keeps in LDS (20 bytes each)
// nodes and primitive records are read through 32-bit-offset buffer descriptors: each array stays under 4 GiB
constexpr uint32_t kMaxNodes = 0x7FFFFFFFu / 32u - 1u;
constexpr uint32_t kMaxPrims = 0xFFFFFFFFu / 64u - 1u;
constexpr uint32_t kTagTriangle = 0, kTagSphere = 1;

struct DevLight // 144 bytes
{
    uint32_t type;      // PT_LIGHT_*
    int32_t nSamples;
    uint32_t drawOffset; // area lights: index of this light's first sample among all area-light samples
    uint32_t shapeType;  // area lights: PT_PRIM_TRIANGLE / PT_PRIM_SPHERE
    float Lemit[3];
    float invArea;       // area lights: 1 / Area(), the area pdf (src/shapes.cpp:48,113)
    float a[3];          // point: position; directional: direction; sphere light: centre
    float radius;        // sphere light
    float v0[3], pad0;   // triangle light: vertices and vertex normals
    float v1[3], pad1;
    float v2[3], pad2;
    float n0[3], pad3;
    float n1[3], pad4;
    float n2[3], pad5;
};
static_assert( sizeof( DevLight ) == 144, "DevLight is nine float4" );

struct DevTexture
{
    int32_t width, height;
    uint32_t offsetLo, offsetHi; // texel offset into the RGBA8 pool
};

struct SceneView
{
    const float4* nodes;
    const float4* orderedNodes; // second layout for front-to-back traversal: 4 x float4 per node, or nullptr
    const float4* shadowNodes;  // third layout, in the space of the one directional light: 1 x float4 per record, or nullptr
    const uint32_t* shadowTopWords; // second word of the first numShadowTop records
    uint32_t numShadowNodes, numShadowTop;
    float lightU[3], lightV[3]; // shadowNodes: two axes across the light's direction
    float shadowOrigin[2], shadowInvCell[2]; // shadowNodes: grid coordinate along u = ( dot( p, lightU ) - shadowOrigin[0] ) * shadowInvCell[0]
    uint32_t shadowDepthAxis;   // shadowNodes: the world axis the rays mostly travel along; a record's third word is the box's far plane
    float shadowDepthSign;      // there in the rays' sense: max if they travel up the axis (+1), -min if down (-1)
    const float4* wideNodes;    // fourth layout (wide_bvh.h): 5 x float4 per node of up to eight children, or nullptr
    const float4* widePrims;    // the primitive records in the order the wide nodes refer to them
    uint32_t numWideNodes;
    const float4* lightWideNodes; // the wide nodes in the frame of the one directional light: 4 x float4 per node, same indices, or nullptr
    uint32_t wideDepth;         // levels of wide nodes: the deepest a traversal stack can get
    const float4* prims;
    const float4* triShade;
    const float4* spheres;
    const float4* materials;
    const DevTexture* textures;
    const uint32_t* texels;
    const DevLight* lights;
    const float4* skybox; // 6 * w * h texels or nullptr
    uint32_t numNodes;
    uint32_t numTopNodes;
    uint32_t numOrderedNodes; // records in orderedNodes (nodes + 1: record 1 is padding, so that sibling pairs share a 128-byte line)
    uint32_t numTopOrdered;   // records at the front of orderedNodes that hold the top of the tree
    uint32_t numPrims;
    uint32_t numSpheres;
    uint32_t numLights;
    int32_t shadowSlots;     // sum of nSamples over all lights
    int32_t drawsPerBounce;  // 2 * (area light samples) + 2
    int32_t skyboxWidth, skyboxHeight;
    float background[3];
};

struct CameraFrame // src/path_tracer.cpp:168-173, computed on the host with the host's tanf
{
    float position[3];
    float UL[3];
    float dU[3];
    float dV[3];
};

struct RngParams
{
    uint32_t mode;
    uint32_t seedLo, seedHi;
    const float* table;
    uint64_t tableLen;
    int32_t sppTotal;
    int32_t tableDraws;
};

// A vector of the path state as it is stored: three floats, no padding
struct Packed3
{
    float x, y, z;
};

// Path state of one pass.  What a bounce reads - ray, throughput, closest hit - is indexed by queue position
// (`entry` = unit * kUnitPaths + position in the unit's segment of the bounce's queue), so the arrays stay dense
// as paths die; the bounce writes the state of the next one at the positions it appends to.  At bounce 0 the
// queue is the identity (entry = path id).  Shadow records belong to the bounce that casts them and are indexed by
// its entries too; only the radiance, which outlives the bounces, is indexed by path id.
struct PathBuffers
{
    Packed3* rayO[2];  // [entry] origin, double buffered per bounce (12 bytes an entry: the streams of path state are what
    Packed3* rayD[2];  // [entry] direction                          ShadeKernel's time goes by, and a fourth word would be padding)
    Packed3* thr[2];   // [entry] throughput
    float4* hit;       // [entry] t, u, v, prim (int bits; -1 = miss)
    float4* rad;       // [path] accumulated radiance of the path
    Packed3* shadowO;  // [entry] origin of the path's shadow rays of this bounce
    float4* shDir;     // [slot * entries + entry]: wi.xyz, tMax (-1 = no ray, -2 = occluded); entries = units * kUnitPaths
    float4* shEst;     // [slot * entries + entry]: unoccluded contribution; with directSum: throughput * contribution, path id
    uint32_t* queue[2];     // [entry] path id, double buffered per bounce
    uint32_t* shadowPaths;  // entries (of this bounce's queue) whose path cast at least one shadow ray
    uint32_t* shadowRays;   // shadow record indices (slot * entries + entry) to test for occlusion, kUnitPaths * slots per unit
    uint32_t* sphereQueue;  // entries whose closest hit is a sphere, shaded by their own kernel instance
    uint32_t* counts;       // [bounce][kCountsPerBounce][unit] queue segment lengths
    uint32_t* cursors;      // [bounce][2] next unit to hand out in ExtendKernel / ConnectKernel
    uint32_t* redo;         // entries whose front-to-back closest hit was ambiguous, traced again in the reference's order
    uint32_t* redoCount;    // [bounce] entries in redo
    uint64_t capacity;      // paths in flight; units * kUnitPaths * shadow slots < 2^32
    uint32_t units;         // units of kUnitPaths consecutive path ids in this pass
    uint32_t shadowSlots;   // shadow records per path, at least 1
    uint32_t directSum;     // 1: one light with one sample - ConnectKernel adds the direct light itself, no ResolveKernel
    float origin0[3];       // where every ray of bounce 0 starts (the camera): not stored per path, like the throughput 1 they start with
    // 1: every shadow ray of this render has the direction shadowDir and no end (one directional light, one sample, traced by
    // ConnectWideKernel): ShadeKernel does not write shDir and the any-hit kernel does not read it
    uint32_t constantShadowDir;
    float shadowDir[3];
};

// Paths per unit of work; every queue has one segment of this many entries (times shadowSlots for shadowRays) per unit
constexpr uint32_t kUnitPaths = 256;

// Queue length slots of one bounce in PathBuffers::counts
enum
{
    kCountNextPaths, kCountShadowPaths, kCountShadowRays, kCountSphereHits, kCountsPerBounce
};

// Counter slots in the device statistics block
enum
{
    ST_RAYS_CLOSEST, ST_RAYS_SHADOW, ST_NODES_CLOSEST, ST_NODES_SHADOW, ST_TRI_CLOSEST, ST_TRI_SHADOW,
    ST_SPH_CLOSEST, ST_SPH_SHADOW, ST_SHADED, ST_STEPS_CLOSEST, ST_STEPS_SHADOW, ST_RAYS_REDONE, ST_COUNT
};
// The shading kernel runs one wave per unit of 256 paths - a million waves per launch of a large pass - and atomics on
// one address complete one after the other (about 13 ns each: 13 ms per launch, measured).  Its count of shaded hits
// therefore goes to one of kShadedSlots counters behind the block, 128 bytes apart; the read-back sums them.
constexpr uint32_t kShadedSlots = 4096, kShadedStride = 16; // counters, and the distance between two in 64-bit words
constexpr size_t kStatsWords = ST_COUNT + static_cast< size_t >( kShadedSlots ) * kShadedStride;

struct LaunchCfg
{
    hipStream_t stream;
    int blocks;        // grid size of the grid-stride kernels
    int extendBlocks;  // resident blocks of ExtendKernel
    int connectBlocks; // resident blocks of ConnectKernel
    bool ordered;      // front-to-back traversal over SceneView::orderedNodes
    bool wide = false; // closest hits over SceneView::wideNodes (implies the rest of `ordered`)
    bool wideShadows = false;    // shadow rays over SceneView::wideNodes (ConnectWideKernel)
    int wideShadowBlocks = 0;    // resident blocks of ConnectWideKernel
    int wideBlocks = 0;          // resident blocks of ExtendWideKernel
    uint2* wideSpill = nullptr;  // stack entries beyond the ones a lane keeps in LDS: [level][thread of the grid]
};

// --- launch wrappers (kernels.hip) ---------------------------------------------------------------------
void LaunchRaygen( const LaunchCfg& cfg, const CameraFrame& cam, const RngParams& rng, const PathBuffers& pb,
                   uint32_t width, uint32_t height, uint32_t sampleBegin, uint32_t numPaths );
// resident blocks per compute unit of the two traversal kernels (occupancy query)
void TraversalOccupancy( bool spheres, int* extendBlocksPerCu, int* connectBlocksPerCu );
// resident blocks per compute unit and threads per block of the wide traversal kernel
void WideOccupancy( bool spheres, int* blocksPerCu, int* threadsPerBlock, int* ldsStackEntries );
void WideShadowOccupancy( int* blocksPerCu, int* threadsPerBlock );
void LaunchExtend( const LaunchCfg& cfg, const SceneView& sv, const PathBuffers& pb, int bounce, uint32_t numPaths, uint64_t* stats );
void LaunchShade( const LaunchCfg& cfg, const SceneView& sv, const RngParams& rng, const PathBuffers& pb, int bounce,
                  int maxDepth, uint32_t width, uint32_t height, uint32_t sampleBegin, uint32_t numPaths, uint64_t* stats );
void LaunchConnect( const LaunchCfg& cfg, const SceneView& sv, const PathBuffers& pb, int bounce, uint64_t* stats );
void LaunchResolve( const LaunchCfg& cfg, const SceneView& sv, const PathBuffers& pb, int bounce );
void LaunchFilm( const LaunchCfg& cfg, const PathBuffers& pb, float4* film, uint32_t width, uint32_t height, uint32_t samplesInPass );
void LaunchAddFilm( const LaunchCfg& cfg, float4* film, const float4* other, uint32_t numPixels );
void LaunchTonemap( const LaunchCfg& cfg, const float4* film, float* linearOut, float* tonemappedOut, uint32_t numPixels,
                    float exposure, float gamma, uint32_t op );
void LaunchTraceClosest( const LaunchCfg& cfg, const SceneView& sv, uint64_t n, const float* origins, const float* dirs,
                         int32_t* prim, float* t, float* u, float* v, uint32_t* nodes, uint32_t* tests );
void LaunchTraceAny( const LaunchCfg& cfg, const SceneView& sv, uint64_t n, const float* origins, const float* dirs,
                     const float* tMax, uint8_t* occluded, uint32_t* nodes, uint32_t* tests );

} // namespace ptd
