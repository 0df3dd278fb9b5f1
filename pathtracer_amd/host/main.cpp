// pathTracer - command line host with the reference's usage (src/main.cpp:11-53): `pathTracer SCENE_FILE`
// loads the scene, renders it once per entry of "SamplesPerPixel" and saves the image, with the suffix
// "_<spp>" when there are several entries.  The scene comes from libpthost.so (scene JSON, OBJ/MTL, BVH build),
// the rendering from libptamd.so (pt_upload_scene / pt_render / pt_readback): this file is the body of
// PathTracer::Render / SaveImage (src/path_tracer.cpp:158-227) written against the C-ABI.
//
// Everything beyond the positional argument is optional and absent from the reference, whose settings all come
// from the scene file: a resource directory (the reference compiles RESOURCE_DIR in), image size / depth / spp
// overrides for the benchmark configurations, the seed (the reference is unseeded), the GPU to use, and the
// procedural scenes that stand in for assets the reference does not ship.
#include "pt_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <random>
#include <string>
#include <vector>

namespace fs = std::filesystem;

namespace
{

struct Options
{
    std::string sceneFile, resourceDir, output, generate;
    int width = 0, height = 0, depth = -1, spp = 0, device = 0, triangles = 0;
    uint64_t seed = 0;
    bool haveSeed = false, stats = false, hostTonemap = false;
};

void Usage() { std::cout << "Usage: pathTracer SCENE_FILE" << std::endl; }

void Help()
{
    Usage();
    std::cout << "Optional (not in the reference):\n"
                 "  --resource-dir DIR   directory the scene's Filename entries are relative to (default: $PT_RESOURCE_DIR,\n"
                 "                       else the parent of the scene file's directory)\n"
                 "  --width W --height H render at another resolution (aspect ratio follows)\n"
                 "  --depth D            override MaxDepth\n"
                 "  --spp N              render once with N samples per pixel instead of the scene's list\n"
                 "  --seed S             seed of the counter-based generator (default: std::random_device)\n"
                 "  --device N           GPU index\n"
                 "  --output FILE        override OutputImageFilename\n"
                 "  --generate atrium|knot|soup [--triangles N]   procedural scene instead of SCENE_FILE\n"
                 "  --stats              print ray and traversal counters after each render\n"
                 "  --host-tonemap       run the tonemap pass on the host with the C library's powf, as the reference\n"
                 "                       does (the saved picture is then the CPU path's to the last bit for equal samples)\n";
}

bool ParseArgs( int argc, char** argv, Options& o )
{
    for ( int i = 1; i < argc; ++i )
    {
        const std::string a = argv[i];
        auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if ( a == "--help" || a == "-h" )
        {
            Help();
            std::exit( 0 );
        }
        else if ( a == "--stats" )
            o.stats = true;
        else if ( a == "--resource-dir" && ( v = value() ) )
            o.resourceDir = v;
        else if ( a == "--width" && ( v = value() ) )
            o.width = std::atoi( v );
        else if ( a == "--height" && ( v = value() ) )
            o.height = std::atoi( v );
        else if ( a == "--depth" && ( v = value() ) )
            o.depth = std::atoi( v );
        else if ( a == "--spp" && ( v = value() ) )
            o.spp = std::atoi( v );
        else if ( a == "--device" && ( v = value() ) )
            o.device = std::atoi( v );
        else if ( a == "--triangles" && ( v = value() ) )
            o.triangles = std::atoi( v );
        else if ( a == "--output" && ( v = value() ) )
            o.output = v;
        else if ( a == "--generate" && ( v = value() ) )
            o.generate = v;
        else if ( a == "--host-tonemap" )
            o.hostTonemap = true;
        else if ( a == "--seed" && ( v = value() ) )
        {
            o.seed     = std::strtoull( v, nullptr, 0 );
            o.haveSeed = true;
        }
        else if ( !a.empty() && a[0] != '-' && o.sceneFile.empty() )
            o.sceneFile = a;
        else
            return false;
    }
    return !o.sceneFile.empty() || !o.generate.empty();
}

// RESOURCE_DIR of the reference is "<checkout>/resources/" and its scenes live in "<checkout>/resources/scenes/"
std::string DefaultResourceDir( const std::string& sceneFile )
{
    if ( const char* env = std::getenv( "PT_RESOURCE_DIR" ) ) return env;
    fs::path dir = fs::absolute( fs::path( sceneFile ) ).parent_path().parent_path();
    return dir.string() + "/";
}

// PathTracer of src/path_tracer.hpp:9-19 on the C-ABI: Render fills renderedImage, SaveImage writes it.
class PathTracer
{
public:
    PathTracer( PtScene* gpuScene, int width, int height, const PtCamera& camera, bool hostTonemap )
        : m_scene( gpuScene ), m_width( width ), m_height( height ), m_camera( camera ), m_hostTonemap( hostTonemap )
    {
    }

    bool Render( int samplesPerPixel, uint64_t seed, PtStats* stats )
    {
        std::cout << "\nRendering scene with SPP = " << samplesPerPixel << "..." << std::endl;
        const auto timeStart = std::chrono::steady_clock::now();
        renderedImage.assign( static_cast< size_t >( m_width ) * m_height * 3, 0.0f );
        PtRenderParams params{};
        params.struct_size  = sizeof( params );
        params.spp_total    = samplesPerPixel;
        params.sample_begin = 0;
        params.sample_end   = samplesPerPixel;
        params.rng_mode     = PT_RNG_PHILOX;
        params.seed         = seed;
        if ( pt_render( m_scene, &params ) != PT_OK ) return false;
        // the image holds the tonemapped picture, like Image::m_pixels after src/path_tracer.cpp:211-219
        if ( m_hostTonemap )
        {
            if ( pt_readback( m_scene, PT_TONEMAP_UNCHARTED2, renderedImage.data(), nullptr, stats ) != PT_OK ) return false;
            if ( pth_tonemap( renderedImage.data(), static_cast< uint64_t >( m_width ) * m_height, m_camera.exposure, m_camera.gamma, 0 ) != 0 ) return false;
        }
        else if ( pt_readback( m_scene, PT_TONEMAP_UNCHARTED2, nullptr, renderedImage.data(), stats ) != PT_OK )
            return false;
        const double seconds = std::chrono::duration< double >( std::chrono::steady_clock::now() - timeStart ).count();
        std::cout << "\nRendered scene in " << seconds << " seconds" << std::endl;
        return true;
    }

    bool SaveImage( const std::string& filename ) const { return pth_write_image( filename.c_str(), renderedImage.data(), m_width, m_height ) == 0; }

    std::vector< float > renderedImage;

private:
    PtScene* m_scene;
    int m_width, m_height;
    PtCamera m_camera;
    bool m_hostTonemap;
};

} // namespace

int main( int argc, char** argv )
{
    Options opt;
    if ( !ParseArgs( argc, argv, opt ) )
    {
        Usage();
        return 0;
    }

    PthScene* scene = pth_scene_new();
    int rc;
    const auto loadStart = std::chrono::steady_clock::now();
    std::cout << "Loading scene '" << ( opt.generate.empty() ? opt.sceneFile : opt.generate ) << "'..." << std::endl;
    if ( opt.generate.empty() )
    {
        const std::string resourceDir = opt.resourceDir.empty() ? DefaultResourceDir( opt.sceneFile ) : opt.resourceDir;
        rc = pth_load_scene_file( scene, opt.sceneFile.c_str(), resourceDir.c_str() );
    }
    else if ( opt.generate == "atrium" )
        rc = pth_generate_atrium( scene, 1, opt.triangles > 0 ? opt.triangles : 262267, 1024 );
    else if ( opt.generate == "knot" )
        rc = pth_generate_knot( scene, 2, opt.triangles > 0 ? opt.triangles : 871414 );
    else if ( opt.generate == "soup" )
        rc = pth_generate_soup( scene, 3, opt.triangles > 0 ? static_cast< uint64_t >( opt.triangles ) : 20000000ull, 0.01f, 1024 );
    else
    {
        std::cout << "unknown --generate '" << opt.generate << "'" << std::endl;
        return 0;
    }
    if ( rc != 0 )
    {
        // the reference logs the failure and exits with status 0 (src/main.cpp:25-29)
        std::cout << "Could not load scene file '" << ( opt.generate.empty() ? opt.sceneFile : opt.generate ) << "': " << pth_last_error() << std::endl;
        return 0;
    }
    if ( opt.width > 0 && opt.height > 0 ) pth_set_image( scene, opt.width, opt.height, static_cast< float >( opt.width ) / opt.height );
    if ( opt.depth >= 0 ) pth_set_max_depth( scene, opt.depth );

    PtSceneDesc desc;
    PthSceneInfo info;
    if ( pth_scene_desc( scene, opt.device, &desc ) != 0 || pth_scene_info( scene, &info ) != 0 )
    {
        std::cout << pth_last_error() << std::endl;
        return 0;
    }
    // the statistics block of Scene::Load (src/scene.cpp:355-365); loading and the BVH build are one call here
    {
        size_t numSpheres = 0, numTris = 0, numArea = 0, numPoint = 0, numDirectional = 0;
        for ( uint64_t i = 0; i < desc.num_prims; ++i ) ( desc.prims[i].type == PT_PRIM_SPHERE ? numSpheres : numTris ) += 1;
        for ( uint64_t i = 0; i < desc.num_lights; ++i )
            ( desc.lights[i].type == PT_LIGHT_AREA ? numArea : desc.lights[i].type == PT_LIGHT_POINT ? numPoint : numDirectional ) += 1;
        std::cout << "Scene stats:\n------------------------------------------------------\n"
                  << "Load and BVH build time: " << std::chrono::duration< float >( std::chrono::steady_clock::now() - loadStart ).count() << " seconds\n"
                  << "Number of shapes: " << desc.num_prims << "\n\tSpheres: " << numSpheres << "\n\tTriangles: " << numTris << "\n"
                  << "Number of lights: " << desc.num_lights << "\n\tAreaLight: " << numArea << "\n\tPointLight: " << numPoint
                  << "\n\tDirectionalLights: " << numDirectional << std::endl;
    }
    PtScene* gpuScene = nullptr;
    if ( pt_upload_scene( &desc, &gpuScene ) != PT_OK )
    {
        std::cout << "Could not upload the scene: " << pt_last_error() << std::endl;
        return 0;
    }

    std::vector< int > sppList( info.spp, info.spp + info.num_spp );
    if ( opt.spp > 0 ) sppList.assign( 1, opt.spp );
    std::string outputName = opt.output.empty() ? info.output_filename : opt.output;
    std::random_device entropy;

    // Perform all the renderings ("SamplesPerPixel": [ 8, 32, ... ]), src/main.cpp:30-49
    for ( size_t it = 0; it < sppList.size(); ++it )
    {
        PathTracer pathTracer( gpuScene, desc.image_width, desc.image_height, desc.camera, opt.hostTonemap );
        PtStats stats{};
        const uint64_t seed = opt.haveSeed ? opt.seed : ( static_cast< uint64_t >( entropy() ) << 32 | entropy() );
        if ( !pathTracer.Render( sppList[it], seed, &stats ) )
        {
            std::cout << "Render failed: " << pt_last_error() << std::endl;
            break;
        }
        if ( opt.stats )
        {
            const double rays = static_cast< double >( stats.rays_closest + stats.rays_shadow );
            std::printf( "samples %llu, rays %.0f (%.3f per sample), nodes per ray %.2f, %.1f Mrays/s, %.2f Msamples/s\n",
                         (unsigned long long)stats.samples, rays, rays / static_cast< double >( stats.samples ),
                         static_cast< double >( stats.nodes_closest + stats.nodes_shadow ) / rays, rays / stats.render_ms * 1e-3,
                         static_cast< double >( stats.samples ) / stats.render_ms * 1e-3 );
        }
        // if there are multiple renderings, tack on the suffix "_[spp]" to the filename
        std::string filename = outputName;
        if ( sppList.size() > 1 )
        {
            fs::path path( filename );
            filename = path.stem().string() + "_" + std::to_string( sppList[it] ) + path.extension().string();
        }
        if ( !pathTracer.SaveImage( filename ) ) std::cout << "Could not save image '" << filename << "'" << std::endl;
    }

    pt_destroy_scene( gpuScene );
    pth_scene_free( scene );
    return 0;
}
