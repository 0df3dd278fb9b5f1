/*
 * pt_oracle.h - CPU restatement of the reference's hot path (PathTracer::Render / Li and everything they
 * call) over the flattened scene of include/pt_api.h.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: the product path is the
 * HIP library and fails loudly without it.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py load libptoracle.so, and only as the checker / the timed CPU baseline.
 *
 * Pinning: tests/test_oracle_vs_reference.py compares every function here bit for bit with the reference's
 * own code (oracle/_ref, built from /root/reference by oracle/Makefile.ref) wherever that library exists,
 * and tests/golden holds vectors produced by it for machines without the reference.  The cubemap lookup
 * (Skybox::GetPixel, src/resource/skybox.cpp:84-118) cannot be built from the reference here and is
 * restated from its source text only: parity unpinned for that one function.
 *
 * Arithmetic: IEEE binary32 in the reference's operation order, glm 0.9.9.2 semantics, float M_PI, built
 * with -ffp-contract=off.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtoScene PtoScene;

typedef struct PtoStats
{
    uint64_t samples;
    uint64_t rays_closest, rays_shadow;
    uint64_t nodes_closest, nodes_shadow;
    uint64_t tri_tests_closest, tri_tests_shadow;
    uint64_t sphere_tests_closest, sphere_tests_shadow;
    uint64_t shaded_hits;
    uint64_t draws; /* Random::Rand() calls */
    uint64_t rays_redone; /* PT_TRAVERSAL_ORDERED: closest-hit rays traced a second time in the reference's order */
} PtoStats;

/* Copies the scene arrays. */
int  pto_scene_create( const PtSceneDesc* desc, PtoScene** out );
void pto_scene_destroy( PtoScene* scene );

/* Order in which pto_intersect / pto_occluded / pto_li_image visit the children of an inner node: PT_TRAVERSAL_REFERENCE
 * (default; BVH::Intersect's first-child-first order, src/bvh.cpp:254-258) or PT_TRAVERSAL_ORDERED, the front-to-back
 * order of the GPU library's second mode - not a restatement of anything in the reference; it checks that mode's
 * per-ray counters, while its hits are checked against the reference order's. */
int pto_scene_set_traversal( PtoScene* scene, uint32_t traversal );
const char* pto_last_error( void );

/* Scene::Intersect over a batch (src/scene.cpp:370-373, src/bvh.cpp:230-268, src/shapes.cpp:52-76,117-134).
 * Every output pointer may be NULL.  prim = index into prims or -1. */
int pto_intersect( const PtoScene* scene, uint64_t n, const float* origins, const float* dirs, int32_t* prim, float* t,
                   float* u, float* v, float* position, float* normal, float* tangent, float* bitangent, float* uv,
                   int32_t* material, uint32_t* nodes_visited, uint32_t* prim_tests );

/* Scene::Occluded over a batch (src/scene.cpp:381-384, src/bvh.cpp:270-310). */
int pto_occluded( const PtoScene* scene, uint64_t n, const float* origins, const float* dirs, const float* t_max,
                  uint8_t* occluded, uint32_t* nodes_visited, uint32_t* prim_tests );

/* Per pixel, the sum over samples [s0, s1) of Li( jittered primary ray ), added in sample order: the inner
 * loops of PathTracer::Render (src/path_tracer.cpp:168-191).  rgb_sum: H*W*3 floats.  Rows run in parallel
 * on `threads` OpenMP threads (0 = all). */
int pto_li_image( const PtoScene* scene, int32_t spp_total, int32_t s0, int32_t s1, uint32_t rng_mode, uint64_t seed,
                  const float* table, uint64_t table_len, int32_t draws_per_sample, int32_t threads, float* rgb_sum,
                  PtoStats* stats );

/* Number of OpenMP threads pto_li_image uses when `threads` is 0. */
int pto_max_threads( void );

/* Tonemap (op = PT_TONEMAP_*), gamma, clamp of src/path_tracer.cpp:211-219 on n RGB triples. */
int pto_tonemap( uint64_t n, const float* rgb, float exposure, float gamma, uint32_t op, float* out );

/* Leaf functions, element-wise (same argument conventions as the reference functions they restate). */
int pto_ray_aabb( uint64_t n, const float* o, const float* inv_dir, const float* bmin, const float* bmax,
                  const float* max_t, uint8_t* hit );                                     /* intersection_tests.cpp:91-110 */
int pto_ray_triangle( uint64_t n, const float* o, const float* d, const float* v0, const float* v1, const float* v2,
                      const float* max_t, uint8_t* hit, float* t, float* u, float* v );   /* intersection_tests.cpp:34-66  */
int pto_ray_sphere( uint64_t n, const float* o, const float* d, const float* centre, const float* radius,
                    const float* max_t, uint8_t* hit, float* t );                         /* intersection_tests.cpp:10-31  */
int pto_sampling( uint64_t n, const float* u1, const float* u2, float* cosine_hemisphere, float* concentric_disk,
                  float* uniform_sphere, float* uniform_triangle );                       /* sampling.cpp:15-67            */
int pto_texture_fetch( const PtoScene* scene, int32_t texture, uint64_t n, const float* u, const float* v,
                       float* rgba );                                                     /* texture.cpp:49-66             */
int pto_skybox_fetch( const PtoScene* scene, uint64_t n, const float* dirs, float* rgba ); /* skybox.cpp:84-118 (unpinned) */

#ifdef __cplusplus
}
#endif

#endif /* PT_ORACLE_H */
