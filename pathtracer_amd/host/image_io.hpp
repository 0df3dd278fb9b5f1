// Image file readers / writers used by the host (textures in, rendered image out).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pth
{

struct ImageRGBA8
{
    int width = 0, height = 0;
    std::vector< uint8_t > pixels; // 4 bytes per pixel, row 0 first
};

// Reads TGA (true colour, grey and colour-mapped, raw and run-length coded, 8 to 32 bit), binary PPM/PGM (P5, P6) and uncompressed BMP (1 to 32 bit) into
// RGBA8 with the top row first, then flips the rows if `flipVertically` - the contract of
// stbi_load(..., 4) with stbi_set_flip_vertically_on_load as used by Texture::Load
// (src/resource/texture.cpp:19-33).  Also PNG (png_decoder.cpp) and baseline or progressive JPEG
// (jpeg_decoder.cpp).
bool ReadImageRGBA8( const std::string& filename, bool flipVertically, ImageRGBA8& out, std::string& error );

// A picture as floats, the way the reference loads cube map faces (stbi_loadf with its gamma and scale set to 1,
// src/resource/skybox.cpp:33-37): a Radiance .hdr file's radiance as stored (hdr_decoder.cpp), any other format's
// bytes divided by 255.  RGBA, row 0 first, rows flipped if asked.
bool ReadImageRGBA32F( const std::string& filename, bool flipVertically, int& width, int& height, std::vector< float >& pixels, std::string& error );
bool IsRadianceFile( const std::vector< uint8_t >& file );
bool ReadRadianceRGBA32F( const std::vector< uint8_t >& file, int& width, int& height, std::vector< float >& pixels, std::string& error );

// Baseline or progressive JPEG with the arithmetic of the reference's decoder (jpeg_decoder.cpp); rows as stored, top row first.
bool ReadJpegRGBA8( const std::string& filename, ImageRGBA8& out, std::string& error );

// PNG of any colour type, bit depth and interlacing, converted to 8-bit RGBA the way the reference's decoder does.
bool ReadPngRGBA8( const std::vector< uint8_t >& file, ImageRGBA8& out, std::string& error );

// The pass PathTracer::Render ends with (src/path_tracer.cpp:211-219, src/tonemap.cpp:6-9, 29-53), in place, on
// `count` floats: Uncharted2 (op 0) or Reinhard (op 1) at `exposure`, pow( x, 1 / gamma ) with the C library's powf,
// clamp to [0, 1].  Bit for bit the reference's result on the same machine; the device's tonemap kernel
// differs from it by one unit in the last place where powf is not correctly rounded.
void TonemapRGB( float* rgb, size_t count, float exposure, float gamma, int op );

// The reference's quantiser (src/image.cpp:57-68): 255 * clamp(c, 0, 1) truncated to uint8.
std::vector< uint8_t > QuantizeRGB8( const float* rgb, int width, int height );

// Writes 8-bit RGB as .png, .tga, .bmp, .jpg / .jpeg (or, beyond the reference, .ppm) by extension.  The first four
// come out byte for byte as Image::Save writes them (src/image.cpp:43-100, stb_image_write 1.14): PNG with its
// per-row filter choice and fixed-Huffman deflate, run-length TGA, 24-bit BMP, quality-95 4:4:4 baseline JPEG.
// Width and height must be at most 65535 for .tga and .jpg.
std::vector< uint8_t > EncodeJpegRGB8( const uint8_t* rgb, int width, int height );
bool WriteImageRGB8( const std::string& filename, const uint8_t* rgb, int width, int height, std::string& error );

} // namespace pth
