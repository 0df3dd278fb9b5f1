// pathTracer - command line host with the reference's usage (src/main.cpp:11-53): `pathTracer SCENE_FILE`
// loads the scene, renders it once per entry of "SamplesPerPixel" and saves the image, with the suffix
// "_<spp>" when there are several entries.  The scene comes from libpthost.so (scene JSON, OBJ/MTL, BVH build),
// the rendering from libptamd.so (pt_upload_scene / pt_render / pt_readback): this file is the body of
// PathTracer::Render / SaveImage (src/path_tracer.cpp:158-227) written against the C-ABI.
//
// Everything beyond the positional argument is optional and absent from the reference, whose settings all come
// from the scene file: a resource directory (the reference compiles RESOURCE_DIR in), image size / depth / spp
// overrides for the benchmark configurations, the seed (the reference is unseeded), the GPU or GPUs to use, and the
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
#include <algorithm>
#include <thread>
#include <vector>

namespace fs = std::filesystem;

namespace
{

struct Options
{
    std::string sceneFile, resourceDir, output, generate;
    int width = 0, height = 0, depth = -1, spp = 0, device = 0, triangles = 0;
    uint64_t seed = 0;
    bool haveSeed = false, stats = false, hostTonemap = false, referenceOrder = false, binaryTree = false, hostBvh = false;
    std::vector< int > devices; // --devices: the GPUs that share a render; empty = the one of --device
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
                 "  --devices A,B,...    several GPUs share every render: the scene is uploaded to each, the samples of\n"
                 "                       every pixel are split between them and the films are summed on the first\n"
                 "  --output FILE        override OutputImageFilename\n"
                 "  --generate atrium|knot|soup [--triangles N]   procedural scene instead of SCENE_FILE\n"
                 "  --stats              print ray and traversal counters after each render\n"
                 "  --reference-order    visit the BVH in the reference's first-child-first order instead of front to back over the\n"
                 "                       eight-wide layout: same picture, traversal counters equal to the CPU path's, slower on\n"
                 "                       large scenes\n"
                 "  --binary-tree        front to back over the binary tree itself instead of the eight-wide layout (same picture)\n"
                 "  --host-bvh           build SAH trees on the host instead of the GPU (the same tree either way, node for\n"
                 "                       node; the other split methods are always built on the host)\n"
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
        else if ( a == "--devices" && ( v = value() ) )
        {
            o.devices.clear();
            for ( const char* p = v; *p; )
            {
                char* end = nullptr;
                const long d = std::strtol( p, &end, 10 );
                if ( end == p || d < 0 ) return false;
                o.devices.push_back( static_cast< int >( d ) );
                p = *end == ',' ? end + 1 : end;
                if ( *end != ',' && *end != 0 ) return false;
            }
            if ( o.devices.empty() ) return false;
        }
        else if ( a == "--triangles" && ( v = value() ) )
            o.triangles = std::atoi( v );
        else if ( a == "--output" && ( v = value() ) )
            o.output = v;
        else if ( a == "--generate" && ( v = value() ) )
            o.generate = v;
        else if ( a == "--host-tonemap" )
            o.hostTonemap = true;
        else if ( a == "--reference-order" )
            o.referenceOrder = true;
        else if ( a == "--binary-tree" )
            o.binaryTree = true;
        else if ( a == "--host-bvh" )
            o.hostBvh = true;
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

// PathTracer of src/path_tracer.hpp:9-19 on the C-ABI: Render fills renderedImage, SaveImage writes it.  With
// several GPUs (one uploaded copy of the scene each) every GPU renders a contiguous share of the samples of every
// pixel from its own host thread - the shares differ by at most one sample, earlier GPUs take the larger ones -
// and the films (sums per pixel) are added in GPU order on the first GPU (pt_merge_films: the other GPUs' films are
// read in place over xGMI), which tonemaps.  The Philox streams are keyed by the absolute sample index, so the picture does not depend on the
// number of GPUs beyond the order of the float additions.
class PathTracer
{
public:
    // renderFlags: PT_RENDER_* that choose the traversal (the same picture from each)
    PathTracer( const std::vector< PtScene* >& gpuScenes, int width, int height, const PtCamera& camera, bool hostTonemap, uint32_t renderFlags )
        : m_scenes( gpuScenes ), m_width( width ), m_height( height ), m_camera( camera ), m_hostTonemap( hostTonemap ), m_renderFlags( renderFlags )
    {
    }

    bool Render( int samplesPerPixel, uint64_t seed, PtStats* stats )
    {
        std::cout << "\nRendering scene with SPP = " << samplesPerPixel << "..." << std::endl;
        const auto timeStart = std::chrono::steady_clock::now();
        renderedImage.assign( static_cast< size_t >( m_width ) * m_height * 3, 0.0f );
        const int gpus = static_cast< int >( m_scenes.size() );
        std::vector< std::string > errors( gpus );
        std::vector< std::thread > workers;
        for ( int g = 0; g < gpus; ++g )
        {
            const int base = samplesPerPixel / gpus, extra = samplesPerPixel % gpus;
            PtRenderParams params{};
            params.struct_size  = sizeof( params );
            params.flags        = m_renderFlags;
            params.spp_total    = samplesPerPixel;
            params.sample_begin = g * base + std::min( g, extra );
            params.sample_end   = params.sample_begin + base + ( g < extra ? 1 : 0 );
            params.rng_mode     = PT_RNG_PHILOX;
            params.seed         = seed;
            auto work = [this, g, params, &errors]() mutable {
                int rc = pt_render( m_scenes[g], &params );
                if ( rc == PT_ERR_STATE && ( params.flags & PT_RENDER_WIDE ) )
                {
                    // a tree the eight-wide layout cannot express: front to back over the tree itself (a small one: in the reference's order)
                    const bool small = !( params.flags & PT_RENDER_WIDE_SHADOWS );
                    params.flags = ( params.flags & ~static_cast< uint32_t >( PT_RENDER_WIDE | PT_RENDER_WIDE_SHADOWS ) ) | ( small ? 0u : PT_RENDER_ORDERED );
                    rc           = pt_render( m_scenes[g], &params );
                }
                if ( rc != PT_OK ) errors[g] = pt_last_error(); // the message is per thread
            };
            if ( gpus == 1 ) work();
            else workers.emplace_back( work );
        }
        for ( std::thread& t : workers ) t.join();
        for ( int g = 0; g < gpus; ++g )
            if ( !errors[g].empty() )
            {
                error = "GPU " + std::to_string( g ) + ": " + errors[g];
                return false;
            }
        PtStats total{};
        if ( gpus > 1 )
        {
            // the films are summed on the first GPU, the others' read in place over xGMI, in GPU order
            if ( pt_merge_films( m_scenes[0], m_scenes.data() + 1, static_cast< uint32_t >( gpus - 1 ) ) != PT_OK )
            {
                error = std::string( "merging the films: " ) + pt_last_error();
                return false;
            }
            for ( int g = 1; g < gpus; ++g )
            {
                PtStats one{};
                if ( pt_readback( m_scenes[g], PT_TONEMAP_NONE, nullptr, nullptr, &one ) != PT_OK ) return Failed( pt_last_error() );
                Add( total, one );
            }
        }
        // the image holds the tonemapped picture, like Image::m_pixels after src/path_tracer.cpp:211-219
        PtStats first{};
        if ( m_hostTonemap )
        {
            if ( pt_readback( m_scenes[0], PT_TONEMAP_UNCHARTED2, renderedImage.data(), nullptr, &first ) != PT_OK ) return Failed( pt_last_error() );
            if ( pth_tonemap( renderedImage.data(), static_cast< uint64_t >( m_width ) * m_height, m_camera.exposure, m_camera.gamma, 0 ) != 0 ) return Failed( pth_last_error() );
        }
        else if ( pt_readback( m_scenes[0], PT_TONEMAP_UNCHARTED2, nullptr, renderedImage.data(), &first ) != PT_OK )
            return Failed( pt_last_error() );
        Add( total, first );
        if ( stats ) *stats = total;
        const double seconds = std::chrono::duration< double >( std::chrono::steady_clock::now() - timeStart ).count();
        std::cout << "\nRendered scene in " << seconds << " seconds" << std::endl;
        return true;
    }

    bool SaveImage( const std::string& filename ) const { return pth_write_image( filename.c_str(), renderedImage.data(), m_width, m_height ) == 0; }

    std::vector< float > renderedImage;
    std::string error; // what made Render return false (the library keeps its messages per thread, and the GPUs render from threads of their own)

private:
    bool Failed( const char* message )
    {
        error = message;
        return false;
    }

    // counters add up over the GPUs; the GPUs run side by side, so the time is the longest of theirs
    static void Add( PtStats& sum, const PtStats& one )
    {
        sum.samples += one.samples;
        sum.rays_closest += one.rays_closest;
        sum.rays_shadow += one.rays_shadow;
        sum.nodes_closest += one.nodes_closest;
        sum.nodes_shadow += one.nodes_shadow;
        sum.render_ms = std::max( sum.render_ms, one.render_ms );
    }

    std::vector< PtScene* > m_scenes;
    int m_width, m_height;
    PtCamera m_camera;
    bool m_hostTonemap;
    uint32_t m_renderFlags;
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
    // BVH::Build with the SAH split runs on the GPU that renders (pt_build_bvh: the reference's tree, node for node)
    if ( !opt.hostBvh ) pth_set_bvh_builder( pt_build_bvh, opt.devices.empty() ? opt.device : opt.devices[0] );
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
        std::cout << "Could not load scene file '" << ( opt.generate.empty() ? opt.sceneFile : opt.generate ) << "': " << pth_last_error();
        if ( *pt_last_error() ) std::cout << " (" << pt_last_error() << "; --host-bvh builds the tree on the host instead)"; // the GPU tree builder's own words, if it was the one that failed
        std::cout << std::endl;
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
    if ( opt.devices.empty() ) opt.devices.assign( 1, opt.device );
    std::vector< PtScene* > gpuScenes;
    for ( int device : opt.devices )
    {
        PtSceneDesc copy = desc;
        copy.device      = device;
        PtScene* gpuScene = nullptr;
        if ( pt_upload_scene( &copy, &gpuScene ) != PT_OK )
        {
            std::cout << "Could not upload the scene: " << pt_last_error() << std::endl;
            for ( PtScene* uploaded : gpuScenes ) pt_destroy_scene( uploaded );
            return 0;
        }
        gpuScenes.push_back( gpuScene );
    }

    std::vector< int > sppList( info.spp, info.spp + info.num_spp );
    if ( opt.spp > 0 ) sppList.assign( 1, opt.spp );
    std::string outputName = opt.output.empty() ? info.output_filename : opt.output;
    std::random_device entropy;

    // Perform all the renderings ("SamplesPerPixel": [ 8, 32, ... ]), src/main.cpp:30-49
    for ( size_t it = 0; it < sppList.size(); ++it )
    {
        // over the eight-wide layout unless asked otherwise: closest hits always, shadow rays unless the tree is so small that
        // the binary-tree kernels' LDS copy holds it whole, where they are faster over that (same picture either way)
        uint32_t renderFlags = 0;
        if ( opt.binaryTree && !opt.referenceOrder ) renderFlags = PT_RENDER_ORDERED;
        else if ( !opt.referenceOrder ) renderFlags = desc.num_nodes > 2400 ? PT_RENDER_WIDE | PT_RENDER_WIDE_SHADOWS : PT_RENDER_WIDE;
        PathTracer pathTracer( gpuScenes, desc.image_width, desc.image_height, desc.camera, opt.hostTonemap, renderFlags );
        PtStats stats{};
        const uint64_t seed = opt.haveSeed ? opt.seed : ( static_cast< uint64_t >( entropy() ) << 32 | entropy() );
        if ( !pathTracer.Render( sppList[it], seed, &stats ) )
        {
            std::cout << "Render failed: " << pathTracer.error << std::endl;
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

    for ( PtScene* gpuScene : gpuScenes ) pt_destroy_scene( gpuScene );
    pth_scene_free( scene );
    return 0;
}
