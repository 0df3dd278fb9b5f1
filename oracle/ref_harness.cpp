// ref_harness.cpp - drives the reference's own hot-path code as a checker.  TEST INFRASTRUCTURE ONLY.
//
// oracle/Makefile.ref compiles the reference's translation units that build from their own sources with
// g++ (no cmake, no generated header, no Assimp) directly from /root/reference/src:
//     intersection_tests, bvh, shapes, lights, resource/material, resource/texture, sampling, math,
//     transform, aabb, tonemap, camera, anti_aliasing, image, path_tracer, utils/time
// and links them with this file into oracle/_ref/libptref.so.  Nothing is copied from the reference.
//
// The reference's remaining translation units include the cmake-generated configuration.hpp or Assimp
// (scene.cpp, utils/logger.cpp, resource/model.cpp, resource/skybox.cpp, resource/resource_manager.cpp,
// main.cpp) and are NOT built.  What the linked code needs from them is defined below, each a few lines:
//   - PT::Random::Rand / RandFloat   replaces utils/random.cpp on purpose: the reference seeds mt19937 from
//                                    random_device (random.cpp:11-12) and cannot be replayed
//   - PT::Scene::Intersect / Occluded / LEnvironment / ~Scene   (scene.cpp:13-19, 370-393; no skybox here)
//   - PT::MeshInstance::MeshInstance (model.cpp:271-289; the harness receives world-space data, so it only
//                                    stores it)
//   - g_Logger and PrintModifier     (utils/logger.cpp:4-15)
// Consequently the skybox lookup (skybox.cpp:84-118), the scene parser and the Assimp import are not
// covered by this harness.
#include "bvh.hpp"
#include "image.hpp"
#include "stb_image/stb_image.h"
#include "anti_aliasing.hpp"
#include "intersection_tests.hpp"
#include "lights.hpp"
#include "path_tracer.hpp"
#include "resource/material.hpp"
#include "resource/model.hpp"
#include "resource/texture.hpp"
#include "sampling.hpp"
#include "scene.hpp"
#include "shapes.hpp"
#include "tonemap.hpp"
#include "utils/logger.hpp"
#include "utils/random.hpp"

#include "pt_api.h"
#include "pt_rng.h"
#define PT_FLATTEN_NO_SKYBOX // skybox.cpp is not part of this build
#include "pt_flatten.hpp"
#define PT_BINDING_NO_LINK // the GPU library is handed over at run time, see ref_render_gpu
#include "pt_render_binding.hpp"
#include "pt_bvh_binding.hpp"

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unistd.h>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// ------------------------------------------------------------------ pieces of unbuildable translation units
Logger g_Logger;
PrintModifier::PrintModifier( Color c, Emphasis e ) : color( c ), emphasis( e ) {}
std::ostream& operator<<( std::ostream& out, const PrintModifier& mod ) { return out << "\033[" << mod.emphasis << ";" << mod.color << "m"; }

namespace
{

struct ThreadContext
{
    PtRngState rng{};
    bool sequentialRender = false; // true while the reference's own Render drives the sample loop
    uint32_t jitterCalls  = 0;
    uint32_t pixelCount = 0, sppTotal = 0;
    uint64_t raysClosest = 0, raysShadow = 0;
};
thread_local ThreadContext t_ctx;

} // namespace

namespace PT
{

namespace Random
{

    float Rand() { return pt_rng_next( &t_ctx.rng ); }

    float RandFloat( float l, float h )
    {
        // Only AntiAlias::Jitter calls RandFloat, twice per sample (anti_aliasing.cpp:91): while Render itself
        // runs the loops, the first of each pair marks the start of the next sample of its row-major,
        // sample-minor order (path_tracer.cpp:179-191).
        ThreadContext& c = t_ctx;
        if ( c.sequentialRender && ( c.jitterCalls++ & 1u ) == 0 )
        {
            uint32_t index = ( c.jitterCalls - 1 ) / 2;
            c.rng.pixel    = index / c.sppTotal;
            c.rng.sample   = index % c.sppTotal;
            c.rng.draw     = 0;
        }
        return Rand() * ( h - l ) + l; // utils/random.cpp:18-21
    }

} // namespace Random

Scene::~Scene()
{
    for ( auto light : lights ) delete light;
}

bool Scene::Intersect( const Ray& ray, IntersectionData& hitData )
{
    t_ctx.raysClosest++;
    hitData.t = FLT_MAX;
    return bvh.Intersect( ray, &hitData );
}

bool Scene::Occluded( const Ray& ray, float tMax )
{
    t_ctx.raysShadow++;
    return bvh.Occluded( ray, tMax );
}

glm::vec3 Scene::LEnvironment( const Ray& ) { return backgroundRadiance; }

MeshInstance::MeshInstance( const Mesh& worldSpaceMesh, const Transform& identity, std::shared_ptr< Material > material )
    : localToWorld( identity ), worldToLocal( identity ), data( worldSpaceMesh )
{
    if ( material ) data.material = material;
}

glm::vec3 Li( const Ray& ray, Scene* scene ); // path_tracer.cpp:106, external linkage, not in a header

} // namespace PT

// ------------------------------------------------------------------ scene construction from the C-ABI arrays
struct RefScene
{
    PT::Scene scene;
    std::vector< std::shared_ptr< PT::Material > > materials;
    std::vector< std::shared_ptr< PT::Texture > > textures;
    std::vector< std::shared_ptr< PT::MeshInstance > > meshes;
    std::vector< std::shared_ptr< PT::Shape > > ordered; // desc->prims order
    std::map< const PT::Shape*, int > primIndex;
    std::map< const PT::Material*, int > materialIndex;
    PtCamera camera{};
    int width = 0, height = 0;
    std::unique_ptr< PT::FlatScene > flat; // what ref_flatten last produced
};

static thread_local std::string g_error;
static int Fail( const std::string& m )
{
    g_error = m;
    return 1;
}

static glm::vec3 V3( const float* p ) { return glm::vec3( p[0], p[1], p[2] ); }

static bool WriteTga32( const std::string& path, int w, int h, const uint8_t* rgba )
{
    FILE* f = std::fopen( path.c_str(), "wb" );
    if ( !f ) return false;
    uint8_t header[18] = {};
    header[2]  = 2;
    header[12] = w & 0xff;
    header[13] = w >> 8;
    header[14] = h & 0xff;
    header[15] = h >> 8;
    header[16] = 32;
    header[17] = 0x28; // top-left origin, 8 alpha bits
    std::fwrite( header, 1, 18, f );
    std::vector< uint8_t > row( static_cast< size_t >( w ) * 4 );
    for ( int y = 0; y < h; ++y )
    {
        for ( int x = 0; x < w; ++x )
        {
            const uint8_t* p = rgba + 4 * ( static_cast< size_t >( y ) * w + x );
            row[4 * x + 0] = p[2];
            row[4 * x + 1] = p[1];
            row[4 * x + 2] = p[0];
            row[4 * x + 3] = p[3];
        }
        std::fwrite( row.data(), 1, row.size(), f );
    }
    std::fclose( f );
    return true;
}

static bool BuildShapes( const PtSceneDesc* d, RefScene& r, std::string& err )
{
    if ( d->skybox ) { err = "the reference harness has no skybox (skybox.cpp is not buildable here)"; return false; }
    // Texture keeps its pixels private and only Texture::Load fills them (texture.hpp:28-31): hand each
    // texture over as a temporary 32-bit TGA and let the reference's stb_image decode it.
    for ( uint64_t i = 0; i < d->num_textures; ++i )
    {
        const PtTexture& t = d->textures[i];
        std::string path = "/tmp/ptref_tex_" + std::to_string( getpid() ) + "_" + std::to_string( i ) + ".tga";
        if ( !WriteTga32( path, t.width, t.height, d->texels + 4 * t.texel_offset ) ) { err = "cannot write " + path; return false; }
        auto tex = std::make_shared< PT::Texture >();
        PT::TextureCreateInfo info;
        info.name           = "texture" + std::to_string( i );
        info.filename       = path;
        info.flipVertically = false; // the C-ABI carries rows as Texture::Load left them
        bool ok = tex->Load( info );
        std::remove( path.c_str() );
        if ( !ok || tex->GetWidth() != t.width || tex->GetHeight() != t.height ) { err = "Texture::Load failed on " + path; return false; }
        if ( std::memcmp( tex->GetPixels(), d->texels + 4 * t.texel_offset, static_cast< size_t >( t.width ) * t.height * 4 ) != 0 )
        {
            err = "texture round trip through stb_image changed pixels";
            return false;
        }
        r.textures.push_back( tex );
    }
    for ( uint64_t i = 0; i < d->num_materials; ++i )
    {
        auto m    = std::make_shared< PT::Material >();
        m->albedo = V3( d->materials[i].albedo );
        m->Ke     = V3( d->materials[i].Ke );
        if ( d->materials[i].albedo_texture >= 0 ) m->albedoTexture = r.textures.at( d->materials[i].albedo_texture );
        r.materialIndex[m.get()] = static_cast< int >( i );
        r.materials.push_back( m );
    }
    for ( uint64_t i = 0; i < d->num_meshes; ++i )
    {
        const PtMesh& pm = d->meshes[i];
        PT::Mesh mesh;
        mesh.material = r.materials.at( pm.material );
        for ( uint32_t k = 0; k < pm.num_vertices; ++k )
        {
            size_t v = pm.vertex_offset + k;
            mesh.vertices.push_back( V3( d->vertices + 3 * v ) );
            mesh.normals.push_back( V3( d->normals + 3 * v ) );
            mesh.tangents.push_back( V3( d->tangents + 3 * v ) );
            mesh.uvs.push_back( glm::vec2( d->uvs[2 * v], d->uvs[2 * v + 1] ) );
        }
        r.meshes.push_back( std::make_shared< PT::MeshInstance >( mesh, PT::Transform(), nullptr ) );
    }
    for ( uint64_t i = 0; i < d->num_prims; ++i )
    {
        std::shared_ptr< PT::Shape > shape;
        if ( d->prims[i].type == PT_PRIM_TRIANGLE )
        {
            const PtTriangle& t = d->triangles[d->prims[i].index];
            auto tri  = std::make_shared< PT::Triangle >();
            tri->mesh = r.meshes.at( t.mesh );
            tri->i0 = t.i0; tri->i1 = t.i1; tri->i2 = t.i2;
            shape = tri;
        }
        else
        {
            const PtSphere& s = d->spheres[d->prims[i].index];
            auto sp      = std::make_shared< PT::Sphere >();
            sp->position = V3( s.position );
            sp->radius   = s.radius;
            sp->material = r.materials.at( s.material );
            glm::mat4 m;
            std::memcpy( &m[0][0], s.world_to_local, 64 );
            sp->worldToLocal = PT::Transform( m );
            shape = sp;
        }
        r.primIndex[shape.get()] = static_cast< int >( i );
        r.ordered.push_back( shape );
    }
    for ( uint64_t i = 0; i < d->num_lights; ++i )
    {
        const PtLight& l = d->lights[i];
        PT::Light* light = nullptr;
        if ( l.type == PT_LIGHT_POINT )
        {
            auto p = new PT::PointLight;
            p->position = V3( l.position );
            light = p;
        }
        else if ( l.type == PT_LIGHT_DIRECTIONAL )
        {
            auto p = new PT::DirectionalLight;
            p->direction = V3( l.direction );
            light = p;
        }
        else
        {
            auto p = new PT::AreaLight;
            p->shape = r.ordered.at( l.prim );
            light = p;
        }
        light->Lemit    = V3( l.Lemit );
        light->nSamples = l.n_samples;
        r.scene.lights.push_back( light );
    }
    r.scene.backgroundRadiance = V3( d->background_radiance );
    r.scene.maxDepth           = d->max_depth;
    r.scene.imageResolution    = glm::ivec2( d->image_width, d->image_height );
    r.camera = d->camera;
    r.width  = d->image_width;
    r.height = d->image_height;
    return true;
}

static void SetRng( uint32_t mode, uint64_t seed, const float* table, uint64_t tableLen, int sppTotal, int drawsPerSample )
{
    PtRngState& s     = t_ctx.rng;
    s.mode            = mode;
    s.seed            = seed;
    s.table           = table;
    s.table_len       = tableLen;
    s.spp_total       = sppTotal;
    s.draws_per_sample = drawsPerSample;
    s.pixel = s.sample = s.draw = 0;
}

extern "C" {

const char* ref_last_error( void ) { return g_error.c_str(); }

// ---- leaf functions, element-wise over n inputs -------------------------------------------------------
int ref_ray_aabb( uint64_t n, const float* o, const float* invDir, const float* bmin, const float* bmax, const float* maxT, uint8_t* hit )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        glm::vec3 inv = V3( invDir + 3 * i );
        int neg[3]    = { inv.x < 0, inv.y < 0, inv.z < 0 }; // bvh.cpp:236
        hit[i] = PT::intersect::RayAABBFastest( V3( o + 3 * i ), inv, neg, V3( bmin + 3 * i ), V3( bmax + 3 * i ), maxT[i] );
    }
    return 0;
}

int ref_ray_triangle( uint64_t n, const float* o, const float* d, const float* v0, const float* v1, const float* v2,
                      const float* maxT, uint8_t* hit, float* t, float* u, float* v )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        float tt = 0, uu = 0, vv = 0;
        hit[i] = PT::intersect::RayTriangle( V3( o + 3 * i ), V3( d + 3 * i ), V3( v0 + 3 * i ), V3( v1 + 3 * i ), V3( v2 + 3 * i ), tt, uu, vv, maxT[i] );
        t[i] = tt; u[i] = uu; v[i] = vv;
    }
    return 0;
}

int ref_ray_sphere( uint64_t n, const float* o, const float* d, const float* centre, const float* radius, const float* maxT,
                    uint8_t* hit, float* t )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        float tt = 0;
        hit[i] = PT::intersect::RaySphere( V3( o + 3 * i ), V3( d + 3 * i ), V3( centre + 3 * i ), radius[i], tt, maxT[i] );
        t[i] = tt;
    }
    return 0;
}

int ref_sampling( uint64_t n, const float* u1, const float* u2, float* cosineHemisphere, float* concentricDisk,
                  float* uniformSphere, float* uniformTriangle )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        glm::vec3 c = PT::CosineSampleHemisphere( u1[i], u2[i] );
        glm::vec2 k = PT::ConcentricSampleDisk( u1[i], u2[i] );
        glm::vec3 s = PT::UniformSampleSphere( u1[i], u2[i] );
        glm::vec2 t = PT::UniformSampleTriangle( u1[i], u2[i] );
        std::memcpy( cosineHemisphere + 3 * i, &c, 12 );
        std::memcpy( concentricDisk + 2 * i, &k, 8 );
        std::memcpy( uniformSphere + 3 * i, &s, 12 );
        std::memcpy( uniformTriangle + 2 * i, &t, 8 );
    }
    return 0;
}

// Uncharted2 / Reinhard followed by gamma and clamp as in Render (path_tracer.cpp:211-219)
int ref_tonemap( uint64_t n, const float* rgb, float exposure, float gamma, float* uncharted2, float* reinhard )
{
    for ( uint64_t i = 0; i < n; ++i )
    {
        glm::vec3 p = V3( rgb + 3 * i );
        glm::vec3 a = glm::clamp( PT::GammaCorrect( PT::Uncharted2Tonemap( p, exposure ), gamma ), glm::vec3( 0 ), glm::vec3( 1 ) );
        glm::vec3 b = glm::clamp( PT::GammaCorrect( PT::ReinhardTonemap( p, exposure ), gamma ), glm::vec3( 0 ), glm::vec3( 1 ) );
        std::memcpy( uncharted2 + 3 * i, &a, 12 );
        std::memcpy( reinhard + 3 * i, &b, 12 );
    }
    return 0;
}

// Decodes an image file with the reference's Texture::Load (stb_image) and returns its RGBA8 pixels.
int ref_load_texture( const char* filename, int flipVertically, int32_t* width, int32_t* height, uint8_t* rgba, uint64_t capacity )
{
    PT::Texture tex;
    PT::TextureCreateInfo info;
    info.name           = "t";
    info.filename       = filename;
    info.flipVertically = flipVertically != 0;
    if ( !tex.Load( info ) ) return Fail( std::string( "Texture::Load failed for " ) + filename );
    *width  = tex.GetWidth();
    *height = tex.GetHeight();
    uint64_t bytes = static_cast< uint64_t >( tex.GetWidth() ) * tex.GetHeight() * 4;
    if ( rgba )
    {
        if ( bytes > capacity ) return Fail( "ref_load_texture: buffer too small" );
        std::memcpy( rgba, tex.GetPixels(), bytes );
    }
    return 0;
}

// Decodes one cube map face with the calls Skybox::Load makes (src/resource/skybox.cpp:33-37; that file itself
// needs the cmake-generated configuration header and is not part of this build): the reference's stb_image,
// stbi_loadf with four channels, hdr scale and gamma set to 1.
int ref_load_skybox_face( const char* filename, int flipVertically, int32_t* width, int32_t* height, float* rgba, uint64_t capacity )
{
    stbi_set_flip_vertically_on_load( flipVertically != 0 );
    stbi_ldr_to_hdr_scale( 1.0f );
    stbi_ldr_to_hdr_gamma( 1.0f );
    int w, h, nc;
    float* pixels = stbi_loadf( filename, &w, &h, &nc, 4 );
    if ( !pixels ) return Fail( std::string( "stbi_loadf failed for " ) + filename );
    *width  = w;
    *height = h;
    int rc  = 0;
    if ( rgba )
    {
        if ( static_cast< uint64_t >( w ) * h * 4 > capacity ) rc = Fail( "ref_load_skybox_face: buffer too small" );
        else std::memcpy( rgba, pixels, static_cast< size_t >( w ) * h * 16 );
    }
    stbi_image_free( pixels );
    return rc;
}

// ---- scenes ------------------------------------------------------------------------------------------
// Builds reference objects from the flattened scene and adopts its BVH arrays verbatim
// (BVH::nodes / BVH::shapes are public, bvh.hpp:36-38).
int ref_scene_create( const PtSceneDesc* d, RefScene** out )
{
    if ( !d || !out ) return Fail( "ref_scene_create: null argument" );
    static_assert( sizeof( PT::LinearBVHNode ) == sizeof( PtBvhNode ), "node layout" );
    auto r = new RefScene;
    std::string err;
    if ( !BuildShapes( d, *r, err ) )
    {
        delete r;
        return Fail( err );
    }
    r->scene.bvh.shapes = r->ordered;
    r->scene.bvh.nodes  = new PT::LinearBVHNode[d->num_nodes];
    std::memcpy( static_cast< void* >( r->scene.bvh.nodes ), d->nodes, d->num_nodes * sizeof( PtBvhNode ) );
    *out = r;
    return 0;
}

void ref_scene_destroy( RefScene* r ) { delete r; }

// Runs the reference's BVH::Build (bvh.cpp:204-228) over the shapes of `d` taken in d->prims order as the
// scene's shape list.  nodes_out: 2 * num_prims - 1 entries; order_out[k] = index in d->prims of the shape
// the reference placed at position k.
int ref_bvh_build( const PtSceneDesc* d, int splitMethod, PtBvhNode* nodes_out, uint32_t* order_out, uint64_t* num_nodes_out )
{
    RefScene r;
    std::string err;
    if ( !BuildShapes( d, r, err ) ) return Fail( err );
    PT::BVH bvh;
    bvh.splitMethod = splitMethod == 1 ? PT::BVH::SplitMethod::Middle : splitMethod == 2 ? PT::BVH::SplitMethod::EqualCounts : PT::BVH::SplitMethod::SAH;
    std::vector< std::shared_ptr< PT::Shape > > list = r.ordered;
    bvh.Build( list );
    // count nodes by walking the pre-order array: every leaf has exactly one shape per build rules
    uint64_t nodes = 0, leaves = 0;
    while ( leaves < bvh.shapes.size() )
    {
        if ( bvh.nodes[nodes].numShapes > 0 ) leaves += bvh.nodes[nodes].numShapes;
        ++nodes;
    }
    std::memcpy( static_cast< void* >( nodes_out ), bvh.nodes, nodes * sizeof( PtBvhNode ) );
    for ( uint64_t k = 0; k < nodes; ++k ) nodes_out[k].pad = 0; // the reference leaves this byte uninitialised
    for ( size_t k = 0; k < bvh.shapes.size(); ++k ) order_out[k] = r.primIndex.at( bvh.shapes[k].get() );
    *num_nodes_out = nodes;
    return 0;
}

// The same through the reference-side binding of the GPU builder (oracle/pt_bvh_binding.hpp): a PT::BVH filled by
// `build` (pt_build_bvh, passed as a pointer because this library also loads where there is no GPU) instead of by
// BVH::Build, read back the same way.
int ref_bvh_build_with( const PtSceneDesc* d, void* build, int device, PtBvhNode* nodes_out, uint32_t* order_out, uint64_t* num_nodes_out )
{
    RefScene r;
    std::string err;
    if ( !BuildShapes( d, r, err ) ) return Fail( err );
    PT::BVH bvh;
    bvh.splitMethod = PT::BVH::SplitMethod::SAH;
    std::vector< std::shared_ptr< PT::Shape > > list = r.ordered;
    if ( !PT::BuildOnGpu( bvh, list, reinterpret_cast< PT::PtBuildBvhFn >( build ), device ) ) return Fail( "BuildOnGpu: the builder failed" );
    const uint64_t nodes = 2 * bvh.shapes.size() - 1;
    std::memcpy( static_cast< void* >( nodes_out ), bvh.nodes, nodes * sizeof( PtBvhNode ) );
    for ( size_t k = 0; k < bvh.shapes.size(); ++k ) order_out[k] = r.primIndex.at( bvh.shapes[k].get() );
    *num_nodes_out = nodes;
    return 0;
}

// Closest hit per ray.  The reference reports no primitive id, so the traversal of BVH::Intersect
// (bvh.cpp:230-268) is replayed here with the reference's own RayAABBFastest and Shape::Intersect, noting
// the shape that last lowered t; the result is cross-checked against Scene::Intersect.
int ref_intersect( RefScene* r, uint64_t n, const float* origins, const float* dirs, int32_t* prim, float* t, float* u, float* v,
                   float* position, float* normal, float* tangent, float* bitangent, float* uv, int32_t* material,
                   uint32_t* nodesVisited, uint32_t* primTests )
{
    if ( !r ) return Fail( "ref_intersect: null scene" );
    const PT::BVH& bvh = r->scene.bvh;
    int bad = 0;
    #pragma omp parallel for schedule( dynamic, 256 ) reduction( + : bad )
    for ( int64_t i = 0; i < static_cast< int64_t >( n ); ++i )
    {
        PT::Ray ray( V3( origins + 3 * i ), V3( dirs + 3 * i ) );
        PT::IntersectionData hit;
        hit.t = FLT_MAX;
        int best = -1;
        uint32_t nv = 0, pt = 0;
        {
            int stack[64];
            int node = 0, top = 0;
            glm::vec3 inv = glm::vec3( 1.0 ) / ray.direction;
            int neg[3]    = { inv.x < 0, inv.y < 0, inv.z < 0 };
            while ( true )
            {
                const PT::LinearBVHNode& nd = bvh.nodes[node];
                ++nv;
                if ( PT::intersect::RayAABBFastest( ray.position, inv, neg, nd.aabb.min, nd.aabb.max, hit.t ) )
                {
                    if ( nd.numShapes > 0 )
                    {
                        for ( int s = nd.firstIndexOffset; s < nd.firstIndexOffset + nd.numShapes; ++s )
                        {
                            ++pt;
                            if ( bvh.shapes[s]->Intersect( ray, &hit ) ) best = s;
                        }
                        if ( top == 0 ) break;
                        node = stack[--top];
                    }
                    else
                    {
                        stack[top++] = nd.secondChildOffset;
                        node = node + 1;
                    }
                }
                else
                {
                    if ( top == 0 ) break;
                    node = stack[--top];
                }
            }
        }
        PT::IntersectionData check;
        bool found = r->scene.Intersect( ray, check );
        if ( found != ( best >= 0 ) || ( found && check.t != hit.t ) ) bad++;

        if ( prim ) prim[i] = best;
        if ( t ) t[i] = hit.t;
        float uu = 0, vv = 0;
        if ( best >= 0 )
        {
            if ( auto tri = dynamic_cast< const PT::Triangle* >( bvh.shapes[best].get() ) )
            {
                const auto& m = tri->mesh->data;
                float tt;
                PT::intersect::RayTriangle( ray.position, ray.direction, m.vertices[tri->i0], m.vertices[tri->i1], m.vertices[tri->i2], tt, uu, vv, FLT_MAX );
            }
        }
        if ( u ) u[i] = uu;
        if ( v ) v[i] = vv;
        if ( best >= 0 )
        {
            if ( position ) std::memcpy( position + 3 * i, &hit.position, 12 );
            if ( normal ) std::memcpy( normal + 3 * i, &hit.normal, 12 );
            if ( tangent ) std::memcpy( tangent + 3 * i, &hit.tangent, 12 );
            if ( bitangent ) std::memcpy( bitangent + 3 * i, &hit.bitangent, 12 );
            if ( uv ) std::memcpy( uv + 2 * i, &hit.texCoords, 8 );
            if ( material ) material[i] = r->materialIndex.at( hit.material );
        }
        else if ( material )
            material[i] = -1;
        if ( nodesVisited ) nodesVisited[i] = nv;
        if ( primTests ) primTests[i] = pt;
    }
    if ( bad ) return Fail( "replayed traversal disagrees with BVH::Intersect on " + std::to_string( bad ) + " rays" );
    return 0;
}

int ref_occluded( RefScene* r, uint64_t n, const float* origins, const float* dirs, const float* tMax, uint8_t* occluded )
{
    if ( !r ) return Fail( "ref_occluded: null scene" );
    #pragma omp parallel for schedule( dynamic, 256 )
    for ( int64_t i = 0; i < static_cast< int64_t >( n ); ++i )
        occluded[i] = r->scene.Occluded( PT::Ray( V3( origins + 3 * i ), V3( dirs + 3 * i ) ), tMax[i] );
    return 0;
}

// Texture::GetPixel (texture.cpp:49-66) on texture `tex` of the scene.
int ref_texture_fetch( RefScene* r, int tex, uint64_t n, const float* u, const float* v, float* rgba )
{
    if ( !r || tex < 0 || tex >= static_cast< int >( r->textures.size() ) ) return Fail( "ref_texture_fetch: bad texture" );
    for ( uint64_t i = 0; i < n; ++i )
    {
        glm::vec4 c = r->textures[tex]->GetPixel( u[i], v[i] );
        std::memcpy( rgba + 4 * i, &c, 16 );
    }
    return 0;
}

// Per pixel, the sum over samples [s0, s1) (added in sample order, starting from zero) of
// Li( primary ray ) - the inner loop of Render (path_tracer.cpp:181-191) - with the camera-frame arithmetic
// of path_tracer.cpp:168-173 and the reference's AntiAlias::Jitter.  The primary ray starts at the camera
// position (path_tracer.cpp:189 reads the not yet constructed `ray.position`; the intent is cam.position).
//
// emulateStaleOrigin != 0 reproduces, for cross-checking against ref_render only, what the expression at
// path_tracer.cpp:189 reads in THIS build of the reference: g++ 11 -O3 gives `ray` the stack slot of the
// temporary `totalColor / spp` passed to SetPixel (path_tracer.cpp:193), so the first sample of a pixel
// subtracts the previous pixel's colour instead of the camera position, later samples the camera position.
// Rows then run sequentially.
int ref_li_image( RefScene* r, int sppTotal, int s0, int s1, uint32_t rngMode, uint64_t seed, const float* table, uint64_t tableLen,
                  int drawsPerSample, float* rgbSum, uint64_t* raysClosest, uint64_t* raysShadow, int emulateStaleOrigin )
{
    if ( !r || !rgbSum ) return Fail( "ref_li_image: null argument" );
    const PtCamera& cam = r->camera;
    const int W = r->width, H = r->height;
    glm::vec3 position = V3( cam.position ), view = V3( cam.view_dir ), up = V3( cam.up_dir ), right = V3( cam.right_dir );
    float halfHeight = std::tan( cam.vfov / 2 );
    float halfWidth  = halfHeight * cam.aspect_ratio;
    glm::vec3 UL     = position + view + halfHeight * up - halfWidth * right;
    glm::vec3 dU     = right * ( 2 * halfWidth / W );
    glm::vec3 dV     = -up * ( 2 * halfHeight / H );
    UL += 0.5f * ( dU + dV );

    uint64_t closest = 0, shadow = 0;
    glm::vec3 stale = position;
    #pragma omp parallel for schedule( dynamic ) reduction( + : closest, shadow ) if ( !emulateStaleOrigin )
    for ( int row = 0; row < H; ++row )
    {
        SetRng( rngMode, seed, table, tableLen, sppTotal, drawsPerSample );
        t_ctx.sequentialRender = false;
        t_ctx.raysClosest = t_ctx.raysShadow = 0;
        for ( int col = 0; col < W; ++col )
        {
            glm::vec3 imagePlanePos = UL + dV * (float)row + dU * (float)col;
            glm::vec3 total( 0 );
            for ( int s = s0; s < s1; ++s )
            {
                t_ctx.rng.pixel  = static_cast< uint32_t >( row * W + col );
                t_ctx.rng.sample = static_cast< uint32_t >( s );
                t_ctx.rng.draw   = 0;
                glm::vec3 aa = PT::AntiAlias::Jitter( s, imagePlanePos, dU, dV );
                glm::vec3 from = ( emulateStaleOrigin && s == s0 ) ? stale : position;
                PT::Ray ray( position, glm::normalize( aa - from ) );
                total += PT::Li( ray, &r->scene );
            }
            if ( emulateStaleOrigin ) stale = total / (float)sppTotal;
            std::memcpy( rgbSum + 3 * ( static_cast< size_t >( row ) * W + col ), &total, 12 );
        }
        closest += t_ctx.raysClosest;
        shadow += t_ctx.raysShadow;
    }
    if ( raysClosest ) *raysClosest = closest;
    if ( raysShadow ) *raysShadow = shadow;
    return 0;
}

// Gives the reference's Camera the flattened scene's position and lens and `cameraRotation` (radians), lets
// Camera::UpdateOrientationVectors derive the direction vectors and fails if they are not the stored ones.
static int SetCamera( RefScene* r, const float cameraRotation[3] )
{
    PT::Camera& cam = r->scene.camera;
    cam.position    = V3( r->camera.position );
    cam.rotation    = V3( cameraRotation );
    cam.vfov        = r->camera.vfov;
    cam.aspectRatio = r->camera.aspect_ratio;
    cam.exposure    = r->camera.exposure;
    cam.gamma       = r->camera.gamma;
    cam.UpdateOrientationVectors();
    if ( cam.GetViewDir() != V3( r->camera.view_dir ) || cam.GetUpDir() != V3( r->camera.up_dir ) || cam.GetRightDir() != V3( r->camera.right_dir ) )
        return Fail( "Camera::UpdateOrientationVectors disagrees with the camera vectors of the flattened scene" );
    return 0;
}

// Runs the reference's PathTracer::Render itself (path_tracer.cpp:158-222), single threaded so that its
// loops visit samples in row-major, sample-minor order, and returns renderedImage: the tonemapped, gamma
// corrected, clamped picture.  cameraRotation (radians) goes through Camera::UpdateOrientationVectors;
// the call fails if that does not reproduce the direction vectors stored with the scene.
int ref_render( RefScene* r, const float cameraRotation[3], int spp, uint64_t seed, float* rgbTonemapped )
{
    if ( !r || !rgbTonemapped ) return Fail( "ref_render: null argument" );
#ifdef _OPENMP
    omp_set_num_threads( 1 );
#endif
    if ( int rc = SetCamera( r, cameraRotation ) ) return rc;
    r->scene.numSamplesPerPixel = { spp };

    SetRng( 0, seed, nullptr, 0, spp, 0 );
    t_ctx.sequentialRender = true;
    t_ctx.jitterCalls      = 0;
    t_ctx.sppTotal         = spp;
    PT::PathTracer tracer;
    tracer.Render( &r->scene, 0 );
    t_ctx.sequentialRender = false;
#ifdef _OPENMP
    omp_set_num_threads( omp_get_num_procs() );
#endif
    std::memcpy( rgbTonemapped, tracer.renderedImage.GetPixels(), static_cast< size_t >( r->width ) * r->height * 12 );
    return 0;
}

// The reference-side binding (pt_flatten.hpp) on the harness's PT::Scene: PT::Scene -> PtSceneDesc.  The arrays
// stay valid until the next ref_flatten on this scene or ref_scene_destroy.
int ref_flatten( RefScene* r, const float cameraRotation[3], PtSceneDesc* out )
{
    if ( !r || !cameraRotation || !out ) return Fail( "ref_flatten: null argument" );
    if ( int rc = SetCamera( r, cameraRotation ) ) return rc;
    try
    {
        r->flat.reset( new PT::FlatScene );
        PT::Flatten( r->scene, *r->flat );
    }
    catch ( const std::exception& e )
    {
        return Fail( e.what() );
    }
    *out = r->flat->desc;
    return 0;
}

// The reference's host objects over the GPU library: PT::Scene and PT::PathTracer as the reference defines
// them, Render's sample loop replaced by pt_render_binding.hpp, the tonemap pass either the reference's own
// (tonemapOnGpu = 0) or the library's, then the reference's own SaveImage -> Image::Save when `saveAs` is given.  entryPoints: pt_upload_scene, pt_render, pt_readback, pt_destroy_scene, pt_last_error of
// an already loaded libptamd.so, in that order.  Returns renderedImage (tonemapped, W * H * 3 floats).
int ref_render_gpu( RefScene* r, const float cameraRotation[3], int spp, uint64_t seed, void* const entryPoints[5], int tonemapOnGpu, float* rgbTonemapped, const char* saveAs )
{
    if ( !r || !cameraRotation || !entryPoints || !rgbTonemapped ) return Fail( "ref_render_gpu: null argument" );
    if ( int rc = SetCamera( r, cameraRotation ) ) return rc;
    PT::PtApi api;
    api.upload_scene  = reinterpret_cast< decltype( api.upload_scene ) >( entryPoints[0] );
    api.render        = reinterpret_cast< decltype( api.render ) >( entryPoints[1] );
    api.readback      = reinterpret_cast< decltype( api.readback ) >( entryPoints[2] );
    api.destroy_scene = reinterpret_cast< decltype( api.destroy_scene ) >( entryPoints[3] );
    api.last_error    = reinterpret_cast< decltype( api.last_error ) >( entryPoints[4] );
    r->scene.numSamplesPerPixel = { spp };
    PT::PathTracer tracer;
    try
    {
        if ( !PT::RenderOnGpu( tracer, &r->scene, 0, seed, api, tonemapOnGpu != 0 ) ) return Fail( std::string( "RenderOnGpu failed: " ) + api.last_error() );
    }
    catch ( const std::exception& e )
    {
        return Fail( e.what() );
    }
    if ( !tonemapOnGpu ) // the pass Render ends with, src/path_tracer.cpp:211-219
    {
        const PT::Camera& cam = r->scene.camera;
        tracer.renderedImage.ForAllPixels( [&]( const glm::vec3& pixel ) {
            glm::vec3 newColor = pixel;
            newColor           = PT::Uncharted2Tonemap( newColor, cam.exposure );
            newColor           = PT::GammaCorrect( newColor, cam.gamma );
            newColor           = glm::clamp( newColor, glm::vec3( 0 ), glm::vec3( 1 ) );
            return newColor;
        } );
    }
    std::memcpy( rgbTonemapped, tracer.renderedImage.GetPixels(), static_cast< size_t >( r->width ) * r->height * 12 );
    if ( saveAs && !tracer.SaveImage( saveAs ) ) return Fail( "PathTracer::SaveImage failed" );
    return 0;
}

// The reference's Image::Save (src/image.cpp:43-100) on a caller's picture: W * H * 3 floats, row-major.
int ref_save_image( const char* filename, const float* rgb, int width, int height )
{
    if ( !filename || !rgb || width <= 0 || height <= 0 ) return Fail( "ref_save_image: bad argument" );
    PT::Image image( width, height );
    for ( int r = 0; r < height; ++r )
        for ( int c = 0; c < width; ++c )
        {
            const float* p = rgb + 3 * ( static_cast< size_t >( r ) * width + c );
            image.SetPixel( r, c, glm::vec3( p[0], p[1], p[2] ) );
        }
    return image.Save( filename ) ? 0 : Fail( "Image::Save failed" );
}

} // extern "C"
