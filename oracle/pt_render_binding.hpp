// pt_render_binding.hpp - the reference-side binding, second half: the body that replaces
// PathTracer::Render (src/path_tracer.cpp:158-222) when the path runs on the GPU library.
//
// A maintainer of the reference includes this file in src/path_tracer.cpp and replaces the sample loop of
// Render (lines 163-208) by
//     RenderOnGpu( *this, scene, samplesPerPixelIteration, std::random_device{}(), LinkedPtApi(), false );
// which leaves the mean linear radiance in renderedImage, as the loop did; the tonemap pass that follows
// (lines 210-221) stays and the saved picture is the CPU path's to the last bit.  With `true` as the last
// argument the library tonemaps on the GPU and lines 210-221 go as well (its pow differs from the C library's
// powf by one unit in the last place for 0.06 % of values).  Link with -lptamd.  The checking harness cannot link the GPU library (it has to load on machines without
// one), so it defines PT_BINDING_NO_LINK and passes the entry points it was handed at run time; everything
// else is the same code.  TEST INFRASTRUCTURE in this repository; nothing under pathtracer_amd/ includes it.
#pragma once

#include "pt_api.h"
#include "pt_flatten.hpp"

#include "path_tracer.hpp"
#include "utils/logger.hpp"

namespace PT
{

struct PtApi
{
    int ( *upload_scene )( const PtSceneDesc*, PtScene** );
    int ( *render )( PtScene*, const PtRenderParams* );
    int ( *readback )( PtScene*, uint32_t, float*, float*, PtStats* );
    int ( *destroy_scene )( PtScene* );
    const char* ( *last_error )( void );
};

#ifndef PT_BINDING_NO_LINK
inline PtApi LinkedPtApi() { return PtApi{ pt_upload_scene, pt_render, pt_readback, pt_destroy_scene, pt_last_error }; }
#endif

// Fills tracer.renderedImage (Image::m_pixels: W * H * 3 floats, row-major) with the picture of
// numSamplesPerPixel[samplesPerPixelIteration] samples per pixel: the mean linear radiance the CPU loop leaves
// there, or the tonemapped picture Render ends with.  SaveImage and main need no change.  The reference renders each "SamplesPerPixel" entry with a fresh PathTracer
// (src/main.cpp:32-35), so the scene is uploaded per call here; a host that renders many times keeps the PtScene.
inline bool RenderOnGpu( PathTracer& tracer, Scene* scene, int samplesPerPixelIteration, uint64_t seed, const PtApi& api, bool tonemapOnGpu )
{
    tracer.renderedImage      = Image( scene->imageResolution.x, scene->imageResolution.y );
    const int samplesPerPixel = scene->numSamplesPerPixel[samplesPerPixelIteration];

    PtScene* gpuScene = nullptr;
    {
        FlatScene flat;
        Flatten( *scene, flat );
        if ( api.upload_scene( &flat.desc, &gpuScene ) != PT_OK )
        {
            LOG_ERR( api.last_error() );
            return false;
        }
    } // the library has copied everything

    PtRenderParams params{};
    params.struct_size  = sizeof( params );
    params.spp_total    = samplesPerPixel;
    params.sample_begin = 0;
    params.sample_end   = samplesPerPixel;
    params.rng_mode     = PT_RNG_PHILOX;
    params.seed         = seed;
    float* pixels = reinterpret_cast< float* >( tracer.renderedImage.GetPixels() );
    bool ok       = api.render( gpuScene, &params ) == PT_OK &&
              api.readback( gpuScene, PT_TONEMAP_UNCHARTED2, tonemapOnGpu ? nullptr : pixels, tonemapOnGpu ? pixels : nullptr, nullptr ) == PT_OK;
    if ( !ok ) LOG_ERR( api.last_error() );
    api.destroy_scene( gpuScene );
    return ok;
}

} // namespace PT
