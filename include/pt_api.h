/*
 * pt_api.h - C-ABI of the MI355X path-tracing core (libptamd.so).
 *
 * The reference renderer (LiamTyler/PathTracer) has no plugin or FFI interface.  Its narrowest seam is
 *     void PathTracer::Render( Scene* scene, int samplesPerPixelIteration )      src/path_tracer.hpp:14
 * which reads the immutable Scene (src/scene.hpp:27-37) and fills PathTracer::renderedImage
 * (src/path_tracer.hpp:18).  This header is that seam as plain C: every entry point takes plain pointers
 * and sizes, no C++ or torch types.  A host (the reference's own C++ host, or the host in
 * pathtracer_amd/host/) flattens its Scene into a PtSceneDesc, and the three calls
 *     pt_upload_scene -> pt_render -> pt_readback
 * replace the body of PathTracer::Render (src/path_tracer.cpp:158-222).
 *
 * Ownership: the caller owns every host array and may free it as soon as the call that received it
 * returns.  The library owns device memory from pt_upload_scene until pt_destroy_scene.
 * Errors: every call returns 0 on success, a PT_ERR_* code otherwise, and never throws or aborts;
 * pt_last_error() gives a thread-local message.  (The reference's convention is bool + LOG_ERR,
 * src/main.cpp:16,27.)
 * Threading: calls on one PtScene are not re-entrant; one process drives one GPU.
 * All floating point data is IEEE binary32; matrices are column-major like glm.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API_VERSION 2u

enum
{
    PT_OK              = 0,
    PT_ERR_INVALID_ARG = 1, /* null pointer, bad size, struct_size/version mismatch           */
    PT_ERR_INVALID_SCENE = 2, /* index out of range, malformed BVH, unsupported value        */
    PT_ERR_DEVICE      = 3, /* a HIP call failed (no device, out of memory, launch failure)  */
    PT_ERR_STATE       = 4  /* call order: readback before render, ...                       */
};

/* Byte-identical to the reference's LinearBVHNode (src/bvh.hpp:10-21): 32 bytes, alignment 4.
 * Nodes are in depth-first pre-order: the first child of inner node i is i+1, `offset` is the index of
 * its second child (src/bvh.cpp:180-202).  For a leaf (num_shapes > 0) `offset` is the first index of
 * its shapes in PtSceneDesc::prims and the leaf covers [offset, offset + num_shapes). */
typedef struct PtBvhNode
{
    float    aabb_min[3];
    float    aabb_max[3];
    int32_t  offset;
    uint16_t num_shapes;
    uint8_t  axis;
    uint8_t  pad;
} PtBvhNode;

enum
{
    PT_PRIM_TRIANGLE = 0,
    PT_PRIM_SPHERE   = 1
};

/* One entry of the reference's reordered shape list BVH::shapes (src/bvh.cpp:217-221).  The position of
 * an entry in PtSceneDesc::prims is the "primitive id" every hit query reports. */
typedef struct PtPrim
{
    uint32_t type;  /* PT_PRIM_*                                   */
    uint32_t index; /* into PtSceneDesc::triangles or ::spheres    */
} PtPrim;

/* Triangle (src/shapes.hpp:57-60): three mesh-local vertex indices and the mesh they index. */
typedef struct PtTriangle
{
    uint32_t i0, i1, i2;
    uint32_t mesh;
} PtTriangle;

/* MeshInstance::data (src/resource/model.hpp:26-32, baked to world space by model.cpp:271-289).  The
 * vertex attributes of all meshes are concatenated in PtSceneDesc::vertices/normals/tangents/uvs;
 * vertex k of this mesh is element vertex_offset + k of each. */
typedef struct PtMesh
{
    uint32_t vertex_offset;
    uint32_t num_vertices;
    int32_t  material;
    uint32_t reserved;
} PtMesh;

/* Sphere (src/shapes.hpp:40-47).  world_to_local is Transform(position, rotation, radius).Inverse()
 * (src/scene.cpp:273), column-major. */
typedef struct PtSphere
{
    float   position[3];
    float   radius;
    float   world_to_local[16];
    int32_t material;
} PtSphere;

/* The fields of Material the integrator reads (src/resource/material.hpp:28-34). */
typedef struct PtMaterial
{
    float   albedo[3];
    float   Ke[3];
    int32_t albedo_texture; /* index into textures, -1 = none */
} PtMaterial;

/* Texture (src/resource/texture.hpp:16-32): RGBA8 rows exactly as stbi_load returned them (already
 * flipped if the scene asked for it).  Texels live in PtSceneDesc::texels at texel_offset (in texels). */
typedef struct PtTexture
{
    int32_t  width, height;
    uint64_t texel_offset;
} PtTexture;

enum
{
    PT_LIGHT_POINT       = 0, /* src/lights.hpp:29-34 */
    PT_LIGHT_DIRECTIONAL = 1, /* src/lights.hpp:36-41 */
    PT_LIGHT_AREA        = 2  /* src/lights.hpp:47-52 */
};

/* One entry of Scene::lights, in that order (src/scene.hpp:29). */
typedef struct PtLight
{
    uint32_t type;
    int32_t  n_samples;    /* Light::nSamples (src/lights.hpp:24, src/scene.cpp:333-339) */
    float    Lemit[3];
    float    position[3];  /* point light                                               */
    float    direction[3]; /* directional light, normalised by the host (scene.cpp:97)  */
    int32_t  prim;         /* area light: index into prims of the emitting shape        */
} PtLight;

/* Camera (src/camera.hpp:16-32); the three direction vectors are the results of
 * Camera::UpdateOrientationVectors (src/camera.cpp:6-14).  Angles in radians. */
typedef struct PtCamera
{
    float position[3];
    float view_dir[3];
    float up_dir[3];
    float right_dir[3];
    float vfov;
    float aspect_ratio;
    float exposure;
    float gamma;
} PtCamera;

/* PtSceneDesc::flags.  None of them changes a result; they select between equivalent device layouts and kernel
 * sequences, for measurements and tests. */
enum
{
    PT_SCENE_NO_DIRECT_SUM    = 1u << 0, /* always run the separate direct-light summation pass (PT_K_RESOLVE), also for
                                            a scene with one light and one sample per hit                             */
    PT_SCENE_TOP_BREADTH_FIRST = 1u << 1, /* top of the tree in plain breadth-first order instead of largest box first */
    PT_SCENE_NO_PARALLEL_SHADOWS = 1u << 3, /* do not build the light-space node layout that PT_RENDER_ORDERED uses
                                             for the shadow rays of a scene lit by one directional light (same
                                             picture either way; for measurements)                                     */
    PT_SCENE_NO_ORDERED_LAYOUT = 1u << 2, /* do not build the second node layout that PT_RENDER_ORDERED and
                                            PT_TRAVERSAL_ORDERED need (64 bytes of device memory per node); those
                                            then fail with PT_ERR_STATE                                               */
    PT_SCENE_HOST_LAYOUTS = 1u << 5,      /* write the node and primitive records on the host and copy them over,
                                            instead of copying the description and writing them with kernels: the
                                            same bytes, slower for large scenes; kept as the checker of the kernels    */
    PT_SCENE_NO_WIDE_LAYOUT = 1u << 4     /* do not build the eight-wide node layout that PT_RENDER_WIDE and
                                            PT_TRAVERSAL_WIDE need (about 25 bytes per node of the tree and a second
                                            copy of the 64-byte primitive records); those then fail with PT_ERR_STATE   */
};

/* Everything PathTracer::Render reads from Scene (src/scene.hpp:27-37). */
typedef struct PtSceneDesc
{
    uint32_t struct_size; /* sizeof(PtSceneDesc) */
    uint32_t version;     /* PT_API_VERSION      */

    const PtBvhNode*  nodes;      uint64_t num_nodes;
    const PtPrim*     prims;      uint64_t num_prims;
    const PtTriangle* triangles;  uint64_t num_triangles;
    const PtSphere*   spheres;    uint64_t num_spheres;
    const PtMesh*     meshes;     uint64_t num_meshes;
    const float*      vertices;   /* 3 floats per vertex */
    const float*      normals;    /* 3 floats per vertex */
    const float*      tangents;   /* 3 floats per vertex */
    const float*      uvs;        /* 2 floats per vertex */
    uint64_t          num_vertices;
    const PtMaterial* materials;  uint64_t num_materials;
    const PtTexture*  textures;   uint64_t num_textures;
    const uint8_t*    texels;     uint64_t num_texels; /* RGBA8, 4 bytes per texel */
    const PtLight*    lights;     uint64_t num_lights;

    /* Skybox (src/resource/skybox.hpp:21-38): 6 faces in the order right, left, top, bottom, back, front
     * (src/resource/skybox.cpp:23-28), each skybox_width * skybox_height RGBA float texels.  NULL = none,
     * then background_radiance is the miss radiance (src/scene.cpp:386-393). */
    const float* skybox;
    int32_t      skybox_width, skybox_height;
    float        background_radiance[3];

    PtCamera camera;
    int32_t  image_width, image_height; /* Scene::imageResolution */
    int32_t  max_depth;                 /* Scene::maxDepth        */
    int32_t  device;                    /* HIP device ordinal     */
    uint32_t flags;                     /* PT_SCENE_*, 0 = defaults                                             */
    uint32_t top_nodes;                 /* nodes of the top of the tree stored first in device memory (the part
                                           the traversal kernels keep in LDS is its beginning); 0 = default     */
} PtSceneDesc;

enum
{
    PT_RNG_PHILOX = 0, /* Philox4x32-10, key = seed, counter = (pixel, sample, draw / 4); word = draw % 4 */
    PT_RNG_TABLE  = 1  /* draw d of sample s of pixel p = table[((p * spp_total + s) * draws + d) % len]  */
};

enum
{
    PT_TONEMAP_UNCHARTED2 = 0, /* what Render hard-codes (src/path_tracer.cpp:211-219)   */
    PT_TONEMAP_REINHARD   = 1, /* src/tonemap.cpp:29-33                                  */
    PT_TONEMAP_NONE       = 2  /* the reference's TONEMAP_AND_GAMMA NOT_IN_USE build      */
};

enum
{
    PT_RENDER_ACCUMULATE = 1u << 0, /* keep the film of earlier pt_render calls and add to it          */
    PT_RENDER_PROFILE    = 1u << 1, /* bracket every kernel class with events (PtStats::kernel_ms)     */
    PT_RENDER_ORDERED    = 1u << 2, /* front-to-back traversal, see PT_TRAVERSAL_ORDERED; in a scene lit by one
                                       directional light the shadow rays, all parallel, are traced in the light's
                                       space (same occlusion answers, other node counts)                          */
    PT_RENDER_WIDE       = 1u << 3, /* PT_RENDER_ORDERED with the closest hits found over the eight-wide layout, see
                                       PT_TRAVERSAL_WIDE: the same film, a fifth of the node visits                  */
    PT_RENDER_WIDE_SHADOWS = 1u << 4 /* the shadow rays over the eight-wide layout as well, whatever the lights (in place
                                       of the light-space search of PT_RENDER_ORDERED and of the reference-order tree):
                                       same occlusion answers, same film                                             */
};

/* Order in which a ray visits the children of an inner BVH node.
 * PT_TRAVERSAL_REFERENCE: first child first, always - the order of BVH::Intersect / BVH::Occluded
 *   (src/bvh.cpp:254-258, 294-298; the `axis` the reference stores in every node, src/bvh.hpp:19, is never read
 *   there).  Node and primitive-test counters equal the reference's ray by ray.
 * PT_TRAVERSAL_ORDERED: front to back - the child on the ray's side of the split first (the first child if the ray
 *   direction is not negative along the node's `axis`, else the second), so that the closest hit is found early and
 *   culls the rest: 0.8 x the node visits on the Sponza-class scene, 0.3 x on the triangle soup.  Any-hit queries give
 *   the reference's answer in any order (its box test is against the fixed t_max).  For closest-hit queries the
 *   reference's answer depends on its order only among hits that lie within rounding of one another (coplanar or
 *   coincident surfaces, shared edges: whether a box is culled by a hit found earlier can hinge on one ulp).  The
 *   ordered pass keeps everything within a factor 1 + 2^-12 of the closest distance alive, and a ray that saw a
 *   second hit inside that band is traced again in the reference's order (PtStats::rays_redone), as is a ray whose hit
 *   lies in front of its own primitive's box.  Same primitive, t, u, v as PT_TRAVERSAL_REFERENCE; the node and test
 *   counters differ.  One class of rays is outside that statement: a ray for which the REFERENCE's answer is a
 *   primitive whose computed t lies in front of the primitive's bounding box by more than the band - an ill-conditioned
 *   primitive test (a ray that grazes a triangle's plane from far away, a sliver triangle), whose hit point is not on
 *   the primitive - while a sound hit lies between: a traversal that culls boxes by distance does not test that
 *   primitive and returns the sound hit.  None among the 2.7 G closest-hit rays of three 128-sample frames of the
 *   Sponza-class scene; 0.005 % of rays built to provoke it (profiles/soak_traversal.py).  Hosts that need the
 *   reference's answer for those too use PT_TRAVERSAL_REFERENCE.
 * PT_TRAVERSAL_WIDE: over the tree collapsed into nodes of up to eight children whose boxes are stored as 8-bit offsets
 *   on a grid over the node's box, rounded outwards; children are visited front to back by the octant of the ray's
 *   direction, with a short stack per ray.  The quantised boxes only select candidates: a primitive counts if the
 *   reference's own test of its leaf's box (src/bvh.cpp:242) passes, and the rules of PT_TRAVERSAL_ORDERED about hits
 *   within rounding of one another apply, and so does its one exception.  Same primitive, t, u, v and occlusion
 *   answers as PT_TRAVERSAL_REFERENCE; nodes_visited counts wide nodes. */
enum
{
    PT_TRAVERSAL_REFERENCE = 0,
    PT_TRAVERSAL_ORDERED   = 1,
    PT_TRAVERSAL_WIDE      = 2
};

typedef struct PtRenderParams
{
    uint32_t struct_size;      /* sizeof(PtRenderParams)                                             */
    uint32_t flags;            /* PT_RENDER_*                                                         */
    int32_t  spp_total;        /* Scene::numSamplesPerPixel[i]: samples per pixel of the whole image  */
    int32_t  sample_begin;     /* this call renders samples [sample_begin, sample_end) of every pixel */
    int32_t  sample_end;
    uint32_t rng_mode;         /* PT_RNG_*                                                            */
    uint64_t seed;
    const float* sample_table; /* host pointer, PT_RNG_TABLE only                                     */
    uint64_t sample_table_len;
    int32_t  table_draws_per_sample;
    int32_t  samples_per_pass; /* samples of every pixel in flight at once; 0 = choose                */
    void*    stream;           /* hipStream_t to launch on, NULL = the library's own stream           */
} PtRenderParams;

enum
{
    PT_K_RAYGEN  = 0,
    PT_K_EXTEND  = 1, /* closest-hit traversal: BVH::Intersect, src/bvh.cpp:230-268 */
    PT_K_SHADE   = 2,
    PT_K_CONNECT = 3, /* any-hit traversal:     BVH::Occluded,  src/bvh.cpp:270-310 */
    PT_K_FILM    = 4,
    PT_K_TONEMAP = 5,
    PT_K_RESOLVE = 6, /* direct-light sum over the unoccluded shadow records, src/path_tracer.cpp:90-104; not launched
                         for a scene with one light and one sample per hit, where PT_K_CONNECT adds it */
    PT_K_COUNT   = 7
};

/* Counters of everything rendered into the film since it was last cleared.  The byte model of one
 * traced ray is 32 * nodes + 48 * triangle tests + 84 * sphere tests (closest) or 16 * sphere tests
 * (any-hit) + 32 in + 16 out (closest) or 4 out (any-hit). */
typedef struct PtStats
{
    uint64_t samples;            /* pixel samples started                       */
    uint64_t rays_closest;       /* closest-hit rays traced                     */
    uint64_t rays_shadow;        /* any-hit rays traced                         */
    uint64_t nodes_closest;      /* BVH nodes fetched by closest-hit rays       */
    uint64_t nodes_shadow;
    uint64_t tri_tests_closest;
    uint64_t tri_tests_shadow;
    uint64_t sphere_tests_closest;
    uint64_t sphere_tests_shadow;
    uint64_t shaded_hits;
    uint64_t node_steps_closest; /* wave-level node visits (one step serves up to 64 rays): SIMD lane  */
    uint64_t node_steps_shadow;  /* utilisation of the traversal = nodes / (64 * node_steps)          */
    uint64_t rays_redone;        /* PT_RENDER_ORDERED: closest-hit rays traced a second time, in the reference's
                                    order, because two hits lay within rounding of one another; their node visits
                                    and tests of both passes are in the counters above, rays_closest counts them once */
    uint64_t launches[PT_K_COUNT];
    double   kernel_ms[PT_K_COUNT]; /* PT_RENDER_PROFILE only, summed over launches */
    double   render_ms;             /* host wall time of the pt_render calls, synchronised */
} PtStats;

typedef struct PtScene PtScene;

/* Replaces: the Scene the reference hands to PathTracer::Render (src/main.cpp:24,35; src/scene.hpp:15-38).
 * Validates every index and the BVH topology, copies the scene to the GPU named by desc->device in the
 * traversal layout, and returns a handle in *out. */
int pt_upload_scene( const PtSceneDesc* desc, PtScene** out );

/* Replaces: the sample loop of PathTracer::Render (src/path_tracer.cpp:160-206): ray generation with
 * jitter, Li (src/path_tracer.cpp:106-156), and accumulation of samples [sample_begin, sample_end) of
 * every pixel into the film.  Returns after the work has finished on the GPU. */
int pt_render( PtScene* scene, const PtRenderParams* params );

/* Replaces: reading PathTracer::renderedImage after Render (src/path_tracer.hpp:18, src/image.cpp:102-106).
 * rgb_linear    (nullable): W*H*3 floats, row-major r*W+c, = film sum / sample count, i.e. the image before
 *                           the ForAllPixels pass of src/path_tracer.cpp:210-221;
 * rgb_tonemapped(nullable): W*H*3 floats after tonemap(exposure) -> pow(1/gamma) -> clamp[0,1]
 *                           (src/path_tracer.cpp:211-219) with the operator `tonemap` (PT_TONEMAP_*);
 * stats         (nullable): counters since the film was last cleared. */
int pt_readback( PtScene* scene, uint32_t tonemap, float* rgb_linear, float* rgb_tonemapped, PtStats* stats );

/* Replaces: Scene::Intersect (src/scene.cpp:370-373) applied to a batch.  origins/directions: n*3 host
 * floats.  Outputs (host, each n elements, any may be NULL): prim = index into prims of the closest hit
 * or -1, t = hit distance (FLT_MAX on a miss), u/v = triangle barycentrics of src/intersection_tests.cpp:34-66
 * (0 for spheres).  nodes_visited/prim_tests (nullable): per-ray counters. */
int pt_trace_closest( PtScene* scene, uint64_t n, const float* origins, const float* directions,
                      int32_t* prim, float* t, float* u, float* v,
                      uint32_t* nodes_visited, uint32_t* prim_tests );

/* Replaces: Scene::Occluded (src/scene.cpp:381-384) applied to a batch.  t_max: n host floats.
 * occluded: n bytes, 1 = some shape is hit with 0 < t < t_max. */
int pt_trace_any( PtScene* scene, uint64_t n, const float* origins, const float* directions,
                  const float* t_max, uint8_t* occluded,
                  uint32_t* nodes_visited, uint32_t* prim_tests );

/* pt_trace_closest / pt_trace_any with the traversal order chosen by the caller (PT_TRAVERSAL_*). */
int pt_trace_closest_mode( PtScene* scene, uint32_t traversal, uint64_t n, const float* origins, const float* directions,
                           int32_t* prim, float* t, float* u, float* v,
                           uint32_t* nodes_visited, uint32_t* prim_tests );
int pt_trace_any_mode( PtScene* scene, uint32_t traversal, uint64_t n, const float* origins, const float* directions,
                       const float* t_max, uint8_t* occluded,
                       uint32_t* nodes_visited, uint32_t* prim_tests );

/* pt_trace_closest for callers that must not allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, a process whose memory is spoken for).  No counterpart in the reference.  Same arguments and results as
 * pt_trace_closest - it always traverses in the reference's order, PT_TRAVERSAL_REFERENCE - but the rays and results are
 * staged in an array that is part of the library's code object (12 MiB per GPU, there until the process ends) instead of
 * freshly allocated buffers: the call makes no hipMalloc or hipFree.  It copies from and to the caller's host arrays
 * and waits for the GPU before it returns, so it is not for use under stream capture.  At most
 * PT_TRACE_STATIC_MAX_RAYS rays per call, more is PT_ERR_INVALID_ARG.  All scenes on one GPU share that GPU's array, so
 * calls that overlap in time on one GPU race; calls on different GPUs do not. */
#define PT_TRACE_STATIC_MAX_RAYS 262144u
int pt_trace_closest_static( PtScene* scene, uint64_t n, const float* origins, const float* directions,
                             int32_t* prim, float* t, float* u, float* v,
                             uint32_t* nodes_visited, uint32_t* prim_tests );

/* Copies one of the scene's record arrays from the GPU to `dst` (at most `capacity` bytes) and reports its size: for
 * tests that compare the records the upload's kernels wrote with the ones its host loops write (PT_SCENE_HOST_LAYOUTS).
 * dst may be NULL to ask for the size alone.  No counterpart in the reference. */
enum
{
    PT_LAYOUT_REFERENCE_NODES = 0, /* 32-byte records, reference order of visits                    */
    PT_LAYOUT_ORDERED_NODES   = 1, /* 64-byte records with a link per octant                        */
    PT_LAYOUT_PRIMS           = 2, /* 64-byte intersection records                                  */
    PT_LAYOUT_ATTRIBUTES      = 3, /* 128-byte attribute records                                    */
    PT_LAYOUT_WIDE_NODES      = 4, /* 80-byte nodes of up to eight children                         */
    PT_LAYOUT_WIDE_PRIMS      = 5, /* intersection records in the wide layout's order               */
    PT_LAYOUT_SHADOW_NODES    = 6, /* 16-byte light-space records (one directional light)           */
    PT_LAYOUT_SHADOW_TOP      = 7, /* the second word of the first of those records (kept in LDS)   */
    PT_LAYOUT_LIGHT_WIDE_NODES = 8 /* 64-byte records: the wide nodes in the directional light's frame */
};
int pt_read_layout( PtScene* scene, uint32_t which, void* dst, uint64_t capacity, uint64_t* bytes );

/* The film of this scene on its GPU: image_width * image_height float4 (sum r, sum g, sum b, sample
 * count), for a host that merges the films of several GPUs with one RCCL reduce.  No counterpart in the
 * reference (it is single-process). */
int pt_film_device_ptr( PtScene* scene, void** device_ptr, uint64_t* bytes );

/* Adds the films of `sources`, in the order given, to the film of `dst`, on dst's GPU: one host process that drives
 * several GPUs (one scene each) sums their films without staging them in host memory.  A source on another GPU is
 * read in place over xGMI (peer access; copied GPU to GPU first where the GPUs cannot map each other's memory).
 * The sources are left as they are; each source's GPU is synchronised first, so a film rendered on a caller's stream is
 * complete.  No counterpart in the reference (it is single-process, single-device). */
int pt_merge_films( PtScene* dst, PtScene* const* sources, uint32_t num_sources );

/* pt_merge_films with PT_MERGE_* flags. */
enum
{
    PT_MERGE_STAGE_ALWAYS = 1u << 0 /* copy every source film into a buffer on dst's GPU before it is added, whether or not
                                       dst's GPU can map it: the path taken between GPUs without peer access, selectable so
                                       that it can be run (and timed) on any machine, one GPU included */
};
int pt_merge_films_ex( PtScene* dst, PtScene* const* sources, uint32_t num_sources, uint32_t flags );

/* Clears the film and the counters. */
int pt_clear_film( PtScene* scene );

/* Checkpoint and resume of a long render: copies the film (image_width * image_height * 4 floats: sum r, sum g,
 * sum b, sample count per pixel, row-major) to or from host memory.  After pt_upload_film, pt_render with
 * PT_RENDER_ACCUMULATE and the remaining sample range continues the picture; because the film holds plain sums
 * in sample order, the result is bit-identical to an uninterrupted render.  No counterpart in the reference. */
int pt_download_film( PtScene* scene, float* film_rgba );
int pt_upload_film( PtScene* scene, const float* film_rgba );

/* Replaces: BVH::Build with SplitMethod::SAH (src/bvh.cpp:204-228, the recursion of src/bvh.cpp:33-178 and the
 * flattening of src/bvh.cpp:180-202), on GPU `device`.  `boxes` are n * 6 floats (min xyz, max xyz), one box per
 * shape in scene order - what Shape::WorldSpaceAABB gives BVH::Build (src/bvh.cpp:209-215).  nodes_out receives the
 * 2n - 1 nodes of the reference's pre-order array (every leaf holds one shape), order_out the n input indices in the
 * order of BVH::shapes (order_out[k] = input index of the shape at position k; src/bvh.cpp:217-221), *num_nodes_out
 * the node count, *device_ms_out the time the GPU spent building (without the copies); either of the last two may be
 * null.  The tree equals the reference's node for node: same split axis, same bucket costs, same order of the shapes
 * after every std::partition and std::nth_element (libstdc++ as in GCC 11) - a zero bound of a box may differ in its
 * sign, which no consumer can tell.  Bounds must be finite.  The same interface as pth_build_bvh of pt_host.h, which
 * is the host builder for the other split methods. */
int pt_build_bvh( int32_t device, const float* boxes, uint64_t n, PtBvhNode* nodes_out, uint32_t* order_out,
                  uint64_t* num_nodes_out, float* device_ms_out );

/* Replaces: Scene::~Scene / BVH::~BVH (src/scene.cpp:13-19, src/bvh.cpp:7-13). */
int pt_destroy_scene( PtScene* scene );

/* Message of the last failing call on this thread ("" if none). */
const char* pt_last_error( void );

#ifdef __cplusplus
}
#endif

#endif /* PT_API_H */
