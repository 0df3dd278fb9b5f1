// Wavefront OBJ / MTL loader for the host scene path.
//
// The reference loads models through Assimp 5.0 with aiProcess_Triangulate | GenNormals |
// JoinIdenticalVertices | CalcTangentSpace | RemoveRedundantMaterials and then post-processes each mesh
// itself (src/resource/model.cpp:158-261).  Assimp is not carried here; this loader produces the same kind
// of result for the OBJ subset the reference's scenes use (v / vt / vn / f with negative indices,
// g / o / usemtl splitting, mtllib with Kd / Ks / Ke / Ns / map_Kd), then applies the reference's own
// post-processing steps.
#pragma once

#include "host_scene.hpp"
#include <string>

namespace pth
{

struct ModelLoadOptions
{
    std::string name;
    std::string filename;
    bool recalculateNormals = true;    // src/resource/model.hpp:20
    std::string resourceDir;           // searched for textures named by map_Kd (src/resource/model.cpp:37-52)
};

// Loads the model, registers its materials and textures in `scene`, appends it to scene.models.
// Returns false and sets `error` on failure.
bool LoadObjModel( HostScene& scene, const ModelLoadOptions& opts, std::string& error );

} // namespace pth
