#include "pt_host.h"
#include "generators.hpp"
#include "host_scene.hpp"
#include "image_io.hpp"
#include "scene_loader.hpp"
#include <cstring>
#include <iostream>
#include <string>

struct PthScene
{
    pth::HostScene scene;
};

namespace
{

thread_local std::string g_error;

int Fail( const std::string& msg )
{
    g_error = msg;
    return 1;
}

template < typename F >
int Guard( F&& f )
{
    try
    {
        return f();
    }
    catch ( const std::exception& e )
    {
        return Fail( e.what() );
    }
    catch ( ... )
    {
        return Fail( "unknown error" );
    }
}

pth::SplitMethod ToSplit( int32_t m )
{
    switch ( m )
    {
        case PTH_SPLIT_MIDDLE: return pth::SplitMethod::Middle;
        case PTH_SPLIT_EQUAL_COUNTS: return pth::SplitMethod::EqualCounts;
        default: return pth::SplitMethod::SAH;
    }
}

} // namespace

extern "C" {

PthScene* pth_scene_new( void )
{
    try
    {
        return new PthScene();
    }
    catch ( ... )
    {
        g_error = "out of memory";
        return nullptr;
    }
}

void pth_scene_free( PthScene* scene ) { delete scene; }

int pth_load_scene_file( PthScene* scene, const char* filename, const char* resource_dir )
{
    if ( !scene || !filename ) return Fail( "pth_load_scene_file: null argument" );
    return Guard( [&] {
        pth::SceneLoadOptions opts;
        opts.resourceDir = resource_dir ? resource_dir : "";
        opts.log         = []( const std::string& line ) { std::cout << line << std::endl; };
        std::string err;
        if ( !pth::LoadSceneFile( scene->scene, filename, opts, err ) ) return Fail( err );
        return 0;
    } );
}

int pth_generate_atrium( PthScene* scene, uint32_t seed, int32_t triangles, int32_t texture_size )
{
    if ( !scene ) return Fail( "pth_generate_atrium: null scene" );
    if ( texture_size < 1 || texture_size > 8192 ) return Fail( "pth_generate_atrium: texture_size out of range" );
    return Guard( [&] {
        pth::GenerateAtrium( scene->scene, seed, triangles, texture_size );
        return 0;
    } );
}

int pth_generate_knot( PthScene* scene, uint32_t seed, int32_t triangles )
{
    if ( !scene ) return Fail( "pth_generate_knot: null scene" );
    return Guard( [&] {
        pth::GenerateKnot( scene->scene, seed, triangles );
        return 0;
    } );
}

int pth_generate_soup( PthScene* scene, uint32_t seed, uint64_t triangles, float edge, int32_t skybox_size )
{
    if ( !scene ) return Fail( "pth_generate_soup: null scene" );
    if ( skybox_size < 1 || skybox_size > 4096 ) return Fail( "pth_generate_soup: skybox_size out of range" );
    return Guard( [&] {
        pth::GenerateSoup( scene->scene, seed, triangles, edge, skybox_size );
        return 0;
    } );
}

int pth_set_image( PthScene* scene, int32_t width, int32_t height, float aspect_ratio )
{
    if ( !scene || width <= 0 || height <= 0 ) return Fail( "pth_set_image: bad argument" );
    scene->scene.imageWidth  = width;
    scene->scene.imageHeight = height;
    if ( aspect_ratio > 0 ) scene->scene.camera.aspect_ratio = aspect_ratio;
    return 0;
}

int pth_set_max_depth( PthScene* scene, int32_t max_depth )
{
    if ( !scene || max_depth < 0 ) return Fail( "pth_set_max_depth: bad argument" );
    scene->scene.maxDepth = max_depth;
    return 0;
}

int pth_scene_desc( PthScene* scene, int32_t device, PtSceneDesc* out )
{
    if ( !scene || !out ) return Fail( "pth_scene_desc: null argument" );
    return Guard( [&] {
        *out = scene->scene.Desc( device );
        return 0;
    } );
}

int pth_scene_info( PthScene* scene, PthSceneInfo* out )
{
    if ( !scene || !out ) return Fail( "pth_scene_info: null argument" );
    std::memset( out, 0, sizeof( *out ) );
    const auto& s = scene->scene;
    // the reference renders every entry of the list (src/main.cpp:32): dropping some silently would drop pictures
    if ( s.numSamplesPerPixel.size() > 32 ) return Fail( "pth_scene_info: \"SamplesPerPixel\" has " + std::to_string( s.numSamplesPerPixel.size() ) + " entries, PthSceneInfo holds 32" );
    if ( s.outputImageFilename.size() >= sizeof( out->output_filename ) ) return Fail( "pth_scene_info: the output file name is longer than 511 bytes" );
    out->num_spp  = static_cast< int32_t >( s.numSamplesPerPixel.size() );
    for ( int i = 0; i < out->num_spp; ++i ) out->spp[i] = s.numSamplesPerPixel[i];
    out->samples_per_area_light = s.numSamplesPerAreaLight;
    out->split_method = s.splitMethod == pth::SplitMethod::SAH ? PTH_SPLIT_SAH
                        : s.splitMethod == pth::SplitMethod::Middle ? PTH_SPLIT_MIDDLE : PTH_SPLIT_EQUAL_COUNTS;
    std::strncpy( out->output_filename, s.outputImageFilename.c_str(), sizeof( out->output_filename ) - 1 );
    return 0;
}

int pth_build_bvh( const float* boxes, uint64_t n, int32_t split_method, PtBvhNode* nodes_out, uint32_t* order_out,
                   uint64_t* num_nodes_out )
{
    if ( !boxes || !nodes_out || !order_out || !num_nodes_out || n == 0 ) return Fail( "pth_build_bvh: bad argument" );
    return Guard( [&] {
        std::vector< pth::Aabb > bounds( n );
        for ( uint64_t i = 0; i < n; ++i )
        {
            bounds[i].min = pth::vec3( boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2] );
            bounds[i].max = pth::vec3( boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5] );
        }
        pth::BvhResult r = pth::BuildBvh( bounds, ToSplit( split_method ) );
        std::memcpy( nodes_out, r.nodes.data(), r.nodes.size() * sizeof( PtBvhNode ) );
        std::memcpy( order_out, r.order.data(), r.order.size() * sizeof( uint32_t ) );
        *num_nodes_out = r.nodes.size();
        return 0;
    } );
}

int pth_set_bvh_builder( PthBvhBuilder builder, int32_t device )
{
    pth::SetExternalBvhBuilder( builder, device );
    return 0;
}

int pth_camera_vectors( const float rotation[3], float view[3], float up[3], float right[3] )
{
    if ( !rotation || !view || !up || !right ) return Fail( "pth_camera_vectors: null argument" );
    pth::HostScene s;
    s.cameraRotation = pth::vec3( rotation[0], rotation[1], rotation[2] );
    s.UpdateCameraOrientation();
    std::memcpy( view, s.camera.view_dir, 12 );
    std::memcpy( up, s.camera.up_dir, 12 );
    std::memcpy( right, s.camera.right_dir, 12 );
    return 0;
}

int pth_sphere_world_to_local( const float position[3], const float rotation[3], float radius, float out16[16] )
{
    if ( !position || !rotation || !out16 ) return Fail( "pth_sphere_world_to_local: null argument" );
    pth::mat4 m = pth::inverse( pth::makeTransform( pth::vec3( position[0], position[1], position[2] ),
                                                    pth::vec3( rotation[0], rotation[1], rotation[2] ), pth::vec3( radius ) ) );
    for ( int c = 0; c < 4; ++c )
        for ( int r = 0; r < 4; ++r ) out16[4 * c + r] = m[c][r];
    return 0;
}

int pth_write_image( const char* filename, const float* rgb, int32_t width, int32_t height )
{
    if ( !filename || !rgb || width <= 0 || height <= 0 ) return Fail( "pth_write_image: bad argument" );
    return Guard( [&] {
        std::vector< uint8_t > q = pth::QuantizeRGB8( rgb, width, height );
        std::string err;
        if ( !pth::WriteImageRGB8( filename, q.data(), width, height, err ) ) return Fail( err );
        return 0;
    } );
}

int pth_tonemap( float* rgb, uint64_t num_pixels, float exposure, float gamma, int32_t op )
{
    if ( !rgb || ( op != 0 && op != 1 ) ) return Fail( "pth_tonemap: bad argument" );
    return Guard( [&] {
        pth::TonemapRGB( rgb, static_cast< size_t >( num_pixels ) * 3, exposure, gamma, op );
        return 0;
    } );
}

int pth_read_image( const char* filename, int32_t flip_vertically, int32_t* width, int32_t* height, uint8_t* rgba, uint64_t capacity )
{
    if ( !filename || !width || !height ) return Fail( "pth_read_image: bad argument" );
    return Guard( [&] {
        pth::ImageRGBA8 img;
        std::string err;
        if ( !pth::ReadImageRGBA8( filename, flip_vertically != 0, img, err ) ) return Fail( err );
        *width  = img.width;
        *height = img.height;
        if ( rgba )
        {
            if ( capacity < img.pixels.size() ) return Fail( "pth_read_image: buffer too small" );
            std::memcpy( rgba, img.pixels.data(), img.pixels.size() );
        }
        return 0;
    } );
}

int pth_read_image_float( const char* filename, int32_t flip_vertically, int32_t* width, int32_t* height, float* rgba, uint64_t capacity )
{
    if ( !filename || !width || !height ) return Fail( "pth_read_image_float: bad argument" );
    return Guard( [&] {
        std::vector< float > pixels;
        std::string err;
        int w = 0, h = 0;
        if ( !pth::ReadImageRGBA32F( filename, flip_vertically != 0, w, h, pixels, err ) ) return Fail( err );
        *width  = w;
        *height = h;
        if ( rgba )
        {
            if ( capacity < pixels.size() ) return Fail( "pth_read_image_float: buffer too small" );
            std::memcpy( rgba, pixels.data(), pixels.size() * sizeof( float ) );
        }
        return 0;
    } );
}

const char* pth_last_error( void ) { return g_error.c_str(); }

} // extern "C"
