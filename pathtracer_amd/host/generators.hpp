// Procedural scenes for the benchmark configurations whose assets are not available.
//
// The reference's sponza.obj, dragon.obj and eight sponza textures are not part of its checkout
// (.MISSING_LARGE_BLOBS:3-13) and nothing can be downloaded, so the benchmark configurations that name
// them run on stand-ins built here with the same triangle counts, the same cameras and lights as the
// reference's scene files, and comparable occlusion structure.  Everything is seeded and deterministic.
#pragma once

#include "host_scene.hpp"
#include <cstdint>

namespace pth
{

// Atrium stand-in for resources/scenes/sponza.json: a long two-storey colonnaded courtyard with an open
// roof, curtains and props, `targetTriangles` triangles (default 262,267 like Crytek Sponza), 24 materials
// with procedural `textureSize`^2 RGBA8 diffuse textures, Kd 0.4704 like sponza.mtl.  Camera, light,
// depth and model scale follow resources/scenes/sponza.json:4-24.
void GenerateAtrium( HostScene& scene, uint32_t seed, int targetTriangles, int textureSize );

// Closed displaced torus-knot mesh stand-in for resources/scenes/dragon.json (871,414 triangles in the
// original) on a ground plane, one directional light.
void GenerateKnot( HostScene& scene, uint32_t seed, int targetTriangles );

// Random triangle soup: centres uniform in [-1,1]^3, edge vectors uniform in [-edge,edge]^3, one of 64
// Lambert albedos uniform in [0.2,0.8] per 4096-triangle mesh, inside a procedural LDR cubemap.
void GenerateSoup( HostScene& scene, uint32_t seed, uint64_t numTriangles, float edge, int skyboxSize );

} // namespace pth
