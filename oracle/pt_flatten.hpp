// pt_flatten.hpp - the reference-side binding: PT::Scene -> PtSceneDesc (include/pt_api.h).
//
// This is the one new file a maintainer of the reference adds to put libptamd.so under PathTracer::Render
// (src/path_tracer.cpp:158-222): it walks the loaded scene's public members and lays them out as the arrays
// pt_upload_scene takes.  It lives here, beside the checking harness, because it needs the reference's headers:
// oracle/Makefile.ref compiles it into oracle/_ref/libptref.so (ref_flatten in ref_harness.cpp) and
// tests/test_oracle_vs_reference.py checks that a scene built from a descriptor flattens back to arrays that
// render identically.  TEST INFRASTRUCTURE in this repository; nothing under pathtracer_amd/ includes it.
//
// Define PT_FLATTEN_NO_SKYBOX where src/resource/skybox.cpp is not linked (the harness: that file needs the
// cmake-generated configuration header).
#pragma once

#include "pt_api.h"

#include "lights.hpp"
#include "resource/material.hpp"
#include "resource/model.hpp"
#include "resource/texture.hpp"
#include "scene.hpp"
#include "shapes.hpp"

#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

namespace PT
{

// Owns the host arrays the descriptor points into.  pt_upload_scene copies everything, so a FlatScene may be
// destroyed as soon as that call has returned.
struct FlatScene
{
    std::vector< PtBvhNode > nodes;
    std::vector< PtPrim > prims;
    std::vector< PtTriangle > triangles;
    std::vector< PtSphere > spheres;
    std::vector< PtMesh > meshes;
    std::vector< float > vertices, normals, tangents, uvs;
    std::vector< PtMaterial > materials;
    std::vector< PtTexture > textures;
    std::vector< uint8_t > texels;
    std::vector< PtLight > lights;
    PtSceneDesc desc{};

    FlatScene()                   = default;
    FlatScene( const FlatScene& ) = delete; // desc points into the vectors
    FlatScene& operator=( const FlatScene& ) = delete;
};

inline void Flatten( Scene& scene, FlatScene& out, int device = 0 )
{
    static_assert( sizeof( LinearBVHNode ) == sizeof( PtBvhNode ), "PtBvhNode is LinearBVHNode, byte for byte" );
    const auto& shapes = scene.bvh.shapes; // in leaf order (src/bvh.cpp:217-221)
    if ( shapes.empty() ) throw std::runtime_error( "Flatten: the scene has no shapes" );

    // nodes: the pre-order array ends with the leaf that holds the last shape
    size_t numNodes = 0, shapesSeen = 0;
    while ( shapesSeen < shapes.size() ) shapesSeen += scene.bvh.nodes[numNodes++].numShapes;
    out.nodes.resize( numNodes );
    std::memcpy( static_cast< void* >( out.nodes.data() ), scene.bvh.nodes, numNodes * sizeof( PtBvhNode ) );
    for ( PtBvhNode& n : out.nodes ) n.pad = 0; // the reference never writes this byte

    std::map< const Texture*, int32_t > textureIndex;
    std::map< const Material*, int32_t > materialIndex;
    std::map< const MeshInstance*, uint32_t > meshIndex;
    std::map< const Shape*, int32_t > primIndex;

    auto material = [&]( const std::shared_ptr< Material >& m ) -> int32_t
    {
        auto found = materialIndex.find( m.get() );
        if ( found != materialIndex.end() ) return found->second;
        PtMaterial pm{};
        for ( int c = 0; c < 3; ++c )
        {
            pm.albedo[c] = m->albedo[c];
            pm.Ke[c]     = m->Ke[c];
        }
        pm.albedo_texture = -1;
        if ( const Texture* t = m->albedoTexture.get() )
        {
            auto known = textureIndex.find( t );
            if ( known == textureIndex.end() )
            {
                PtTexture pt{};
                pt.width        = t->GetWidth();
                pt.height       = t->GetHeight();
                pt.texel_offset = out.texels.size() / 4;
                const uint8_t* px = reinterpret_cast< const uint8_t* >( t->GetPixels() ); // RGBA8 rows as Texture::Load left them
                out.texels.insert( out.texels.end(), px, px + static_cast< size_t >( pt.width ) * pt.height * 4 );
                known = textureIndex.emplace( t, static_cast< int32_t >( out.textures.size() ) ).first;
                out.textures.push_back( pt );
            }
            pm.albedo_texture = known->second;
        }
        const int32_t index = static_cast< int32_t >( out.materials.size() );
        out.materials.push_back( pm );
        materialIndex.emplace( m.get(), index );
        return index;
    };

    auto mesh = [&]( const std::shared_ptr< MeshInstance >& instance ) -> uint32_t
    {
        auto found = meshIndex.find( instance.get() );
        if ( found != meshIndex.end() ) return found->second;
        const Mesh& data = instance->data; // world space already (src/resource/model.cpp:277-288)
        PtMesh pm{};
        pm.vertex_offset = static_cast< uint32_t >( out.vertices.size() / 3 );
        pm.num_vertices  = static_cast< uint32_t >( data.vertices.size() );
        pm.material      = material( data.material );
        for ( size_t v = 0; v < data.vertices.size(); ++v )
        {
            for ( int c = 0; c < 3; ++c )
            {
                out.vertices.push_back( data.vertices[v][c] );
                out.normals.push_back( data.normals[v][c] );
                out.tangents.push_back( data.tangents[v][c] );
            }
            out.uvs.push_back( data.uvs[v].x );
            out.uvs.push_back( data.uvs[v].y );
        }
        const uint32_t index = static_cast< uint32_t >( out.meshes.size() );
        out.meshes.push_back( pm );
        meshIndex.emplace( instance.get(), index );
        return index;
    };

    for ( const auto& shape : shapes )
    {
        PtPrim prim{};
        if ( const auto* tri = dynamic_cast< const Triangle* >( shape.get() ) )
        {
            prim.type  = PT_PRIM_TRIANGLE;
            prim.index = static_cast< uint32_t >( out.triangles.size() );
            out.triangles.push_back( PtTriangle{ tri->i0, tri->i1, tri->i2, mesh( tri->mesh ) } );
        }
        else if ( const auto* sphere = dynamic_cast< const Sphere* >( shape.get() ) )
        {
            prim.type  = PT_PRIM_SPHERE;
            prim.index = static_cast< uint32_t >( out.spheres.size() );
            PtSphere ps{};
            for ( int c = 0; c < 3; ++c ) ps.position[c] = sphere->position[c];
            ps.radius = sphere->radius;
            std::memcpy( ps.world_to_local, &sphere->worldToLocal.matrix[0][0], 64 ); // column-major, as glm stores it
            ps.material = material( sphere->material );
            out.spheres.push_back( ps );
        }
        else
            throw std::runtime_error( "Flatten: a shape that is neither Triangle nor Sphere" );
        primIndex.emplace( shape.get(), static_cast< int32_t >( out.prims.size() ) );
        out.prims.push_back( prim );
    }

    for ( const Light* light : scene.lights ) // every light is sampled at every bounce, in this order
    {
        PtLight pl{};
        pl.n_samples = light->nSamples;
        pl.prim      = -1;
        for ( int c = 0; c < 3; ++c ) pl.Lemit[c] = light->Lemit[c];
        if ( const auto* point = dynamic_cast< const PointLight* >( light ) )
        {
            pl.type = PT_LIGHT_POINT;
            for ( int c = 0; c < 3; ++c ) pl.position[c] = point->position[c];
        }
        else if ( const auto* directional = dynamic_cast< const DirectionalLight* >( light ) )
        {
            pl.type = PT_LIGHT_DIRECTIONAL;
            for ( int c = 0; c < 3; ++c ) pl.direction[c] = directional->direction[c];
        }
        else if ( const auto* area = dynamic_cast< const AreaLight* >( light ) )
        {
            pl.type = PT_LIGHT_AREA;
            pl.prim = primIndex.at( area->shape.get() );
        }
        else
            throw std::runtime_error( "Flatten: unknown light type" );
        out.lights.push_back( pl );
    }

    PtSceneDesc& d = out.desc;
    d              = PtSceneDesc{};
    d.struct_size  = sizeof( PtSceneDesc );
    d.version      = PT_API_VERSION;
    d.nodes = out.nodes.data();             d.num_nodes = out.nodes.size();
    d.prims = out.prims.data();             d.num_prims = out.prims.size();
    d.triangles = out.triangles.data();     d.num_triangles = out.triangles.size();
    d.spheres = out.spheres.data();         d.num_spheres = out.spheres.size();
    d.meshes = out.meshes.data();           d.num_meshes = out.meshes.size();
    d.vertices = out.vertices.data();       d.normals = out.normals.data();
    d.tangents = out.tangents.data();       d.uvs = out.uvs.data();
    d.num_vertices = out.vertices.size() / 3;
    d.materials = out.materials.data();     d.num_materials = out.materials.size();
    d.textures = out.textures.data();       d.num_textures = out.textures.size();
    d.texels = out.texels.data();           d.num_texels = out.texels.size() / 4;
    d.lights = out.lights.data();           d.num_lights = out.lights.size();
#ifndef PT_FLATTEN_NO_SKYBOX
    if ( scene.skybox )
    {
        d.skybox        = reinterpret_cast< const float* >( scene.skybox->GetPixels() ); // 6 faces of float4
        d.skybox_width  = scene.skybox->GetWidth();
        d.skybox_height = scene.skybox->GetHeight();
    }
#endif
    const Camera& cam = scene.camera;
    for ( int c = 0; c < 3; ++c )
    {
        d.background_radiance[c] = scene.backgroundRadiance[c];
        d.camera.position[c]     = cam.position[c];
        d.camera.view_dir[c]     = cam.GetViewDir()[c];
        d.camera.up_dir[c]       = cam.GetUpDir()[c];
        d.camera.right_dir[c]    = cam.GetRightDir()[c];
    }
    d.camera.vfov         = cam.vfov;
    d.camera.aspect_ratio = cam.aspectRatio;
    d.camera.exposure     = cam.exposure;
    d.camera.gamma        = cam.gamma;
    d.image_width         = scene.imageResolution.x;
    d.image_height        = scene.imageResolution.y;
    d.max_depth           = scene.maxDepth;
    d.device              = device;
}

} // namespace PT
