#include "host_scene.hpp"
#include <stdexcept>

namespace pth
{

static void Store3( float* dst, vec3 v )
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

HostScene::HostScene()
{
    // Camera defaults, src/camera.hpp:16-32
    Store3( camera.position, vec3( 0 ) );
    Store3( camera.view_dir, vec3( 0, 0, -1 ) );
    Store3( camera.up_dir, vec3( 0, 1, 0 ) );
    Store3( camera.right_dir, vec3( 1, 0, 0 ) );
    camera.vfov         = radians( 45.0f );
    camera.aspect_ratio = 16.0f / 9.0f;
    camera.exposure     = 1.0f;
    camera.gamma        = 1.0f;
}

void HostScene::UpdateCameraOrientation()
{
    mat4 rot( 1.0f );
    rot       = rotate( rot, cameraRotation.y, vec3( 0, 1, 0 ) );
    rot       = rotate( rot, cameraRotation.x, vec3( 1, 0, 0 ) );
    vec3 view = xyz( rot * vec4( 0, 0, -1, 0 ) );
    vec3 up   = xyz( rot * vec4( 0, 1, 0, 0 ) );
    Store3( camera.view_dir, view );
    Store3( camera.up_dir, up );
    Store3( camera.right_dir, cross( view, up ) );
}

int HostScene::AddMaterial( const std::string& name, const PtMaterial& m )
{
    // ResourceManager keeps one entry per name; a later material with the same name replaces the earlier
    // one for later lookups (src/resource/resource_manager.cpp).  Instances already emitted keep their index.
    materials.push_back( m );
    materialNames.push_back( name );
    return static_cast< int >( materials.size() ) - 1;
}

int HostScene::FindMaterial( const std::string& name ) const
{
    for ( int i = static_cast< int >( materialNames.size() ) - 1; i >= 0; --i )
        if ( materialNames[i] == name ) return i;
    return -1;
}

int HostScene::AddTexture( const std::string& name, int width, int height, const uint8_t* rgba )
{
    PtTexture t{};
    t.width        = width;
    t.height       = height;
    t.texel_offset = texels.size() / 4;
    texels.insert( texels.end(), rgba, rgba + static_cast< size_t >( width ) * height * 4 );
    textures.push_back( t );
    textureNames.push_back( name );
    return static_cast< int >( textures.size() ) - 1;
}

int HostScene::FindTexture( const std::string& name ) const
{
    for ( int i = static_cast< int >( textureNames.size() ) - 1; i >= 0; --i )
        if ( textureNames[i] == name ) return i;
    return -1;
}

const LocalModel* HostScene::FindModel( const std::string& name ) const
{
    for ( int i = static_cast< int >( models.size() ) - 1; i >= 0; --i )
        if ( models[i].name == name ) return &models[i];
    return nullptr;
}

void RecalculateNormals( LocalMesh& mesh )
{
    mesh.normals.assign( mesh.vertices.size(), vec3( 0 ) );
    for ( size_t i = 0; i + 2 < mesh.indices.size(); i += 3 )
    {
        vec3 a = mesh.vertices[mesh.indices[i + 0]];
        vec3 b = mesh.vertices[mesh.indices[i + 1]];
        vec3 c = mesh.vertices[mesh.indices[i + 2]];
        vec3 n = normalize( cross( b - a, c - a ) );
        mesh.normals[mesh.indices[i + 0]] += n;
        mesh.normals[mesh.indices[i + 1]] += n;
        mesh.normals[mesh.indices[i + 2]] += n;
    }
    for ( auto& n : mesh.normals ) n = normalize( n );
}

void GenerateFallbackTangents( LocalMesh& mesh )
{
    mesh.tangents.assign( mesh.vertices.size(), vec3( 0 ) );
    for ( size_t i = 0; i + 2 < mesh.indices.size(); i += 3 )
    {
        uint32_t i0 = mesh.indices[i + 0];
        uint32_t i1 = mesh.indices[i + 1];
        vec3 n0     = mesh.normals[i0];
        vec3 n1     = mesh.normals[i1];
        vec3 t      = normalize( mesh.vertices[i1] - mesh.vertices[i0] );
        vec3 t0     = normalize( t - n0 * dot( n0, t ) );
        vec3 t1     = normalize( t - n1 * dot( n1, t ) );
        if ( mesh.tangents[i0] == vec3( 0 ) ) mesh.tangents[i0] = t0;
        if ( mesh.tangents[i1] == vec3( 0 ) ) mesh.tangents[i1] = t1;
    }
}

void HostScene::AddMeshInstance( const LocalMesh& mesh, const mat4& localToWorld, int materialOverride )
{
    const size_t n = mesh.vertices.size();
    if ( mesh.normals.size() != n || mesh.tangents.size() != n || mesh.uvs.size() != n )
        throw std::runtime_error( "mesh '" + mesh.name + "': attribute arrays differ in length" );
    const int material = materialOverride >= 0 ? materialOverride : mesh.material;
    if ( material < 0 || material >= static_cast< int >( materials.size() ) )
        throw std::runtime_error( "mesh '" + mesh.name + "' has no material" );

    const mat4 worldToLocal    = inverse( localToWorld );
    const mat4 normalTransform = transpose( worldToLocal );

    PtMesh m{};
    m.vertex_offset = static_cast< uint32_t >( vertices.size() / 3 );
    m.num_vertices  = static_cast< uint32_t >( n );
    m.material      = material;
    const uint32_t meshIndex = static_cast< uint32_t >( meshes.size() );
    meshes.push_back( m );

    for ( size_t i = 0; i < n; ++i )
    {
        vec3 p  = transformPoint( localToWorld, mesh.vertices[i] );
        vec3 nn = normalize( transformVector( normalTransform, mesh.normals[i] ) );
        vec3 t  = transformVector( localToWorld, mesh.tangents[i] );
        vertices.insert( vertices.end(), { p.x, p.y, p.z } );
        normals.insert( normals.end(), { nn.x, nn.y, nn.z } );
        tangents.insert( tangents.end(), { t.x, t.y, t.z } );
        uvs.insert( uvs.end(), { mesh.uvs[i].x, mesh.uvs[i].y } );
    }

    const PtMaterial& mat = materials[material];
    const bool emissive   = !( mat.Ke[0] == 0 && mat.Ke[1] == 0 && mat.Ke[2] == 0 );
    for ( size_t f = 0; f + 2 < mesh.indices.size(); f += 3 )
    {
        PtTriangle tri{ mesh.indices[f], mesh.indices[f + 1], mesh.indices[f + 2], meshIndex };
        PtPrim shape{ PT_PRIM_TRIANGLE, static_cast< uint32_t >( triangles.size() ) };
        triangles.push_back( tri );
        if ( emissive )
        {
            PtLight l{};
            l.type      = PT_LIGHT_AREA;
            l.n_samples = 1;
            l.Lemit[0]  = mat.Ke[0];
            l.Lemit[1]  = mat.Ke[1];
            l.Lemit[2]  = mat.Ke[2];
            l.prim      = -1;
            lights.push_back( l );
            lightShape.push_back( static_cast< int32_t >( shapes.size() ) );
        }
        shapes.push_back( shape );
    }
    accelBuilt = false;
}

void HostScene::AddSphere( vec3 position, vec3 rotation, float radius, int material )
{
    PtSphere s{};
    Store3( s.position, position );
    s.radius   = radius;
    mat4 w2l   = inverse( makeTransform( position, rotation, vec3( radius ) ) );
    for ( int c = 0; c < 4; ++c )
        for ( int r = 0; r < 4; ++r ) s.world_to_local[4 * c + r] = w2l[c][r];
    s.material = material;
    shapes.push_back( PtPrim{ PT_PRIM_SPHERE, static_cast< uint32_t >( spheres.size() ) } );
    spheres.push_back( s );
    accelBuilt = false;
}

void HostScene::BuildAccel()
{
    std::vector< Aabb > bounds( shapes.size() );
    for ( size_t i = 0; i < shapes.size(); ++i )
    {
        Aabb b;
        if ( shapes[i].type == PT_PRIM_TRIANGLE ) // Triangle::WorldSpaceAABB, src/shapes.cpp:143-150
        {
            const PtTriangle& t = triangles[shapes[i].index];
            const float* v      = &vertices[3 * static_cast< size_t >( meshes[t.mesh].vertex_offset )];
            b.Union( vec3( v[3 * t.i0], v[3 * t.i0 + 1], v[3 * t.i0 + 2] ) );
            b.Union( vec3( v[3 * t.i1], v[3 * t.i1 + 1], v[3 * t.i1 + 2] ) );
            b.Union( vec3( v[3 * t.i2], v[3 * t.i2 + 1], v[3 * t.i2 + 2] ) );
        }
        else // Sphere::WorldSpaceAABB, src/shapes.cpp:84-88
        {
            const PtSphere& s = spheres[shapes[i].index];
            vec3 p( s.position[0], s.position[1], s.position[2] );
            b.min = p - vec3( s.radius );
            b.max = p + vec3( s.radius );
        }
        bounds[i] = b;
    }

    BvhResult r = BuildSceneBvh( bounds, splitMethod );
    nodes       = std::move( r.nodes );
    primToShape = std::move( r.order );
    prims.resize( primToShape.size() );
    std::vector< int32_t > shapeToPrim( shapes.size(), -1 );
    for ( size_t k = 0; k < primToShape.size(); ++k )
    {
        prims[k]                    = shapes[primToShape[k]];
        shapeToPrim[primToShape[k]] = static_cast< int32_t >( k );
    }
    for ( size_t i = 0; i < lights.size(); ++i )
    {
        if ( lights[i].type == PT_LIGHT_AREA )
        {
            lights[i].prim      = shapeToPrim[lightShape[i]];
            lights[i].n_samples = numSamplesPerAreaLight; // src/scene.cpp:333-339
        }
    }
    accelBuilt = true;
}

PtSceneDesc HostScene::Desc( int device ) const
{
    if ( !accelBuilt ) throw std::runtime_error( "HostScene::Desc called before BuildAccel" );
    PtSceneDesc d{};
    d.struct_size   = sizeof( PtSceneDesc );
    d.version       = PT_API_VERSION;
    d.nodes         = nodes.data();
    d.num_nodes     = nodes.size();
    d.prims         = prims.data();
    d.num_prims     = prims.size();
    d.triangles     = triangles.data();
    d.num_triangles = triangles.size();
    d.spheres       = spheres.data();
    d.num_spheres   = spheres.size();
    d.meshes        = meshes.data();
    d.num_meshes    = meshes.size();
    d.vertices      = vertices.data();
    d.normals       = normals.data();
    d.tangents      = tangents.data();
    d.uvs           = uvs.data();
    d.num_vertices  = vertices.size() / 3;
    d.materials     = materials.data();
    d.num_materials = materials.size();
    d.textures      = textures.data();
    d.num_textures  = textures.size();
    d.texels        = texels.data();
    d.num_texels    = texels.size() / 4;
    d.lights        = lights.data();
    d.num_lights    = lights.size();
    d.skybox        = skybox.empty() ? nullptr : skybox.data();
    d.skybox_width  = skyboxWidth;
    d.skybox_height = skyboxHeight;
    d.background_radiance[0] = backgroundRadiance.x;
    d.background_radiance[1] = backgroundRadiance.y;
    d.background_radiance[2] = backgroundRadiance.z;
    d.camera        = camera;
    d.image_width   = imageWidth;
    d.image_height  = imageHeight;
    d.max_depth     = maxDepth;
    d.device        = device;
    return d;
}

} // namespace pth
