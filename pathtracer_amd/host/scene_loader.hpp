// JSON scene file -> HostScene, accepting the reference's scene schema (src/scene.cpp:294-368).
#pragma once

#include "host_scene.hpp"
#include <functional>
#include <string>

namespace pth
{

struct SceneLoadOptions
{
    std::string resourceDir;                       // the reference's compiled-in RESOURCE_DIR (cmake/configuration.hpp.in:11), with trailing '/'
    std::function< void( const std::string& ) > log; // receives warnings such as unknown keys
};

// Members are dispatched in document order, duplicate keys included, and unknown keys are reported as
// "WARNING '<key>' not found in mapping" and skipped (src/utils/json_parsing.hpp:47-61).  Ends with the
// BVH build.  Returns false and sets `error` if the file cannot be read or a model fails to load.
bool LoadSceneFile( HostScene& scene, const std::string& filename, const SceneLoadOptions& opts, std::string& error );

} // namespace pth
