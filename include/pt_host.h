/*
 * pt_host.h - C entry points of the host-side library (libpthost.so, CPU only).
 *
 * The reference keeps scene loading, the BVH build, camera set-up and image output in its C++ host
 * (src/scene.cpp, src/bvh.cpp:33-228, src/camera.cpp, src/image.cpp).  libpthost.so is this repository's
 * host for the same jobs; it produces the PtSceneDesc that pt_upload_scene (pt_api.h) consumes.  These
 * entry points exist so that tests, the benchmark and bindings in other languages can drive the host
 * without C++ linkage.  All return 0 on success and a non-zero code with pth_last_error() set otherwise.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include "pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PthScene PthScene;

enum
{
    PTH_SPLIT_SAH          = 0,
    PTH_SPLIT_MIDDLE       = 1,
    PTH_SPLIT_EQUAL_COUNTS = 2
}; /* BVH::SplitMethod, src/bvh.hpp:26 */

typedef struct PthSceneInfo
{
    int32_t num_spp;               /* entries of Scene::numSamplesPerPixel (src/scene.hpp:36)         */
    int32_t spp[32];
    int32_t samples_per_area_light;
    int32_t split_method;          /* PTH_SPLIT_*                                                      */
    char    output_filename[512];  /* Scene::outputImageFilename                                       */
} PthSceneInfo;

PthScene* pth_scene_new( void );
void      pth_scene_free( PthScene* scene );

/* Scene::Load (src/scene.cpp:294-368): parse the JSON scene file, load its models and textures from
 * resource_dir (the reference's compiled-in RESOURCE_DIR, with trailing '/'), build the BVH.
 * Warnings go to stdout like the reference's. */
int pth_load_scene_file( PthScene* scene, const char* filename, const char* resource_dir );

/* Procedural stand-ins for assets that are not available (see pathtracer_amd/host/generators.hpp). */
int pth_generate_atrium( PthScene* scene, uint32_t seed, int32_t triangles, int32_t texture_size );
int pth_generate_knot( PthScene* scene, uint32_t seed, int32_t triangles );
int pth_generate_soup( PthScene* scene, uint32_t seed, uint64_t triangles, float edge, int32_t skybox_size );

/* Overrides for benchmark configurations that render a scene file at another size
 * (aspect_ratio <= 0 keeps the scene's). */
int pth_set_image( PthScene* scene, int32_t width, int32_t height, float aspect_ratio );
int pth_set_max_depth( PthScene* scene, int32_t max_depth );

/* Fills *out with pointers into the scene's arrays; they stay valid until the scene is changed or freed. */
int pth_scene_desc( PthScene* scene, int32_t device, PtSceneDesc* out );
int pth_scene_info( PthScene* scene, PthSceneInfo* out );

/* BVH::Build (src/bvh.cpp:204-228) on n boxes given as n * 6 floats (min xyz, max xyz).  nodes_out must
 * hold 2n - 1 nodes, order_out n indices (order_out[k] = input index of the shape at BVH position k). */
int pth_build_bvh( const float* boxes, uint64_t n, int32_t split_method, PtBvhNode* nodes_out,
                   uint32_t* order_out, uint64_t* num_nodes_out );

/* Lets another builder make the trees of the scenes loaded or generated from now on whose split method is SAH:
 * pt_build_bvh of pt_api.h has this signature and builds the same tree on GPU `device`.  A null builder restores the
 * host builder.  A failing builder fails the scene (there is no silent second try).  Process-wide. */
typedef int ( *PthBvhBuilder )( int32_t device, const float* boxes, uint64_t n, PtBvhNode* nodes_out, uint32_t* order_out,
                                uint64_t* num_nodes_out, float* device_ms_out );
int pth_set_bvh_builder( PthBvhBuilder builder, int32_t device );

/* Camera::UpdateOrientationVectors (src/camera.cpp:6-14): rotation in radians -> view, up, right. */
int pth_camera_vectors( const float rotation[3], float view[3], float up[3], float right[3] );

/* Transform( position, rotation, vec3( radius ) ).Inverse() (src/scene.cpp:273), column-major. */
int pth_sphere_world_to_local( const float position[3], const float rotation[3], float radius, float out16[16] );

/* Image::Save (src/image.cpp:43-100): quantise W*H*3 floats with 255 * clamp01 truncated and write
 * .png / .tga / .bmp / .jpg / .jpeg as the reference's writer does, byte for byte, or .ppm. */
int pth_write_image( const char* filename, const float* rgb, int32_t width, int32_t height );

/* The tonemap pass PathTracer::Render ends with (src/path_tracer.cpp:211-219), in place on num_pixels * 3 floats of
 * linear radiance: op 0 = Uncharted2, 1 = Reinhard; then pow( x, 1 / gamma ) and clamp.  Uses the C library's powf
 * as the reference does, so the result is the reference's bit for bit (pt_readback's tonemap on the device is
 * within one unit in the last place of it). */
int pth_tonemap( float* rgb, uint64_t num_pixels, float exposure, float gamma, int32_t op );

/* Texture::Load's decode (src/resource/texture.cpp:19-33: stbi_load with four channels, rows flipped when asked):
 * RGBA8, row 0 first.  Call with rgba = NULL to get the size.  TGA, PNG, baseline and progressive JPEG, BMP, PPM / PGM. */
int pth_read_image( const char* filename, int32_t flip_vertically, int32_t* width, int32_t* height, uint8_t* rgba, uint64_t capacity );

/* Skybox::Load's decode of one face (src/resource/skybox.cpp:33-37: stbi_loadf, four channels, hdr gamma and scale
 * 1): RGBA float, row 0 first; a Radiance .hdr file's radiance, any other format's bytes / 255.  `capacity` in
 * floats; rgba = NULL to get the size. */
int pth_read_image_float( const char* filename, int32_t flip_vertically, int32_t* width, int32_t* height, float* rgba, uint64_t capacity );

const char* pth_last_error( void );

#ifdef __cplusplus
}
#endif

#endif /* PT_HOST_H */
