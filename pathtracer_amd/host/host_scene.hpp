// HostScene: the host-side scene in the flattened form the C-ABI consumes (include/pt_api.h).
//
// It plays the role of the reference's Scene (src/scene.hpp:15-38) for everything above the drop-in
// boundary: it is filled by the JSON scene loader, the OBJ loader or a procedural generator, it runs the
// BVH build, and desc() hands the result to pt_upload_scene.
#pragma once

#include "bvh_builder.hpp"
#include "hmath.hpp"
#include "pt_api.h"
#include <string>
#include <vector>

namespace pth
{

// Mesh in model space, as the reference's Mesh (src/resource/model.hpp:23-35).
struct LocalMesh
{
    std::string name;
    int material = -1;
    std::vector< vec3 > vertices, normals, tangents;
    std::vector< vec2 > uvs;
    std::vector< uint32_t > indices;
};

struct LocalModel
{
    std::string name;
    std::vector< LocalMesh > meshes;
};

struct HostScene
{
    // ---- settings (src/scene.hpp:27-37) ----
    PtCamera camera{};
    vec3 cameraRotation = vec3( 0 );
    std::string outputImageFilename = "rendered.png";
    int imageWidth = 1280, imageHeight = 720;
    int maxDepth = 5;
    int numSamplesPerAreaLight = 1;
    std::vector< int > numSamplesPerPixel = { 32 };
    SplitMethod splitMethod = SplitMethod::Middle; // src/bvh.hpp:36
    vec3 backgroundRadiance = vec3( 0 );

    // ---- resources ----
    std::vector< PtMaterial > materials;
    std::vector< std::string > materialNames;
    std::vector< PtTexture > textures;
    std::vector< std::string > textureNames;
    std::vector< uint8_t > texels;
    std::vector< LocalModel > models;
    std::vector< float > skybox;
    int skyboxWidth = 0, skyboxHeight = 0;

    // ---- shapes in scene order (Scene::shapes, before the BVH build) ----
    std::vector< PtPrim > shapes;
    std::vector< PtTriangle > triangles;
    std::vector< PtSphere > spheres;
    std::vector< PtMesh > meshes;
    std::vector< float > vertices, normals, tangents, uvs;
    std::vector< PtLight > lights;      // Scene::lights order; PtLight::prim is filled in by BuildAccel()
    std::vector< int32_t > lightShape;  // parallel to lights: scene-order shape index of an area light, else -1
    void AddLight( const PtLight& l ) { lights.push_back( l ); lightShape.push_back( -1 ); }

    // ---- acceleration structure ----
    std::vector< PtBvhNode > nodes;
    std::vector< PtPrim > prims;         // BVH order
    std::vector< uint32_t > primToShape; // BVH position -> scene-order shape index
    bool accelBuilt = false;

    HostScene();

    void UpdateCameraOrientation(); // src/camera.cpp:6-14

    int AddMaterial( const std::string& name, const PtMaterial& m );
    int FindMaterial( const std::string& name ) const;
    int AddTexture( const std::string& name, int width, int height, const uint8_t* rgba );
    int FindTexture( const std::string& name ) const;
    const LocalModel* FindModel( const std::string& name ) const;

    // MeshInstance constructor + EmitTrianglesAndLights (src/resource/model.cpp:271-316): bakes the mesh to
    // world space, appends one triangle per face and one area light per face if the material emits.
    void AddMeshInstance( const LocalMesh& mesh, const mat4& localToWorld, int materialOverride );

    // ParseSphere (src/scene.cpp:260-274)
    void AddSphere( vec3 position, vec3 rotation, float radius, int material );

    // BVH::Build + area-light nSamples assignment (src/scene.cpp:330-339)
    void BuildAccel();

    PtSceneDesc Desc( int device = 0 ) const;
};

// Mesh::RecalculateNormals (src/resource/model.cpp:131-156)
void RecalculateNormals( LocalMesh& mesh );

// Tangent basis for a mesh without usable uv derivatives (src/resource/model.cpp:227-251)
void GenerateFallbackTangents( LocalMesh& mesh );

// Per-vertex tangents from uv derivatives, re-orthogonalised against the normals (defined in obj_loader.cpp)
void ComputeUvTangents( LocalMesh& mesh );

} // namespace pth
