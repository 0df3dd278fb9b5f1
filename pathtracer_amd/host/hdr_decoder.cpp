// Radiance RGBE pictures (.hdr) to float RGBA, for cube map faces: the reference loads its sky with stbi_loadf
// (Skybox::Load, src/resource/skybox.cpp:30-58), which hands such a file's radiance on unscaled - mantissa byte
// times 2^(exponent - 136), alpha 1, a zero exponent byte meaning black - and that is what this reader returns,
// with that loader's reading of the container: the "#?RADIANCE" or "#?RGBE" line, header lines up to an empty one
// of which one must be FORMAT=32-bit_rle_rgbe, a "-Y height +X width" line, then per row either the new run-length
// coding (2, 2, width high, width low, then each of the four bytes of all pixels as runs and literals) or, when the
// width is below 8 or at least 32768 or a row does not start that way, flat four-byte pixels.
#include "image_io.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace pth
{

namespace
{

    struct ByteReader
    {
        const std::vector< uint8_t >& data;
        size_t pos = 0;
        bool AtEnd() const { return pos >= data.size(); }
        int Byte() { return pos < data.size() ? data[pos++] : 0; }
        // the rest of the line, without its newline; at most 1022 characters are kept
        std::string Line()
        {
            std::string line;
            char c = static_cast< char >( Byte() );
            while ( !AtEnd() && c != '\n' )
            {
                line.push_back( c );
                if ( line.size() == 1023 )
                {
                    while ( !AtEnd() && Byte() != '\n' ) {}
                    break;
                }
                c = static_cast< char >( Byte() );
            }
            return line;
        }
    };

    void Convert( const uint8_t rgbe[4], float* out )
    {
        if ( rgbe[3] != 0 )
        {
            const float scale = static_cast< float >( std::ldexp( 1.0f, rgbe[3] - ( 128 + 8 ) ) );
            out[0] = rgbe[0] * scale;
            out[1] = rgbe[1] * scale;
            out[2] = rgbe[2] * scale;
        }
        else
            out[0] = out[1] = out[2] = 0;
        out[3] = 1;
    }

} // namespace

bool IsRadianceFile( const std::vector< uint8_t >& file )
{
    auto startsWith = [&]( const char* s ) { return file.size() >= std::strlen( s ) && std::memcmp( file.data(), s, std::strlen( s ) ) == 0; };
    return startsWith( "#?RADIANCE\n" ) || startsWith( "#?RGBE\n" );
}

bool ReadRadianceRGBA32F( const std::vector< uint8_t >& file, int& width, int& height, std::vector< float >& pixels, std::string& error )
{
    ByteReader in{ file };
    const std::string first = in.Line();
    if ( first != "#?RADIANCE" && first != "#?RGBE" )
    {
        error = "HDR: not a Radiance picture";
        return false;
    }
    bool rgbe = false;
    for ( ;; )
    {
        const std::string line = in.Line();
        if ( line.empty() ) break;
        if ( line == "FORMAT=32-bit_rle_rgbe" ) rgbe = true;
    }
    if ( !rgbe )
    {
        error = "HDR: unsupported format";
        return false;
    }
    const std::string size = in.Line();
    const char* token      = size.c_str();
    char* next             = nullptr;
    if ( std::strncmp( token, "-Y ", 3 ) != 0 )
    {
        error = "HDR: unsupported data layout";
        return false;
    }
    const long h = std::strtol( token + 3, &next, 10 );
    while ( *next == ' ' ) ++next;
    if ( std::strncmp( next, "+X ", 3 ) != 0 )
    {
        error = "HDR: unsupported data layout";
        return false;
    }
    const long w = std::strtol( next + 3, nullptr, 10 );
    if ( w <= 0 || h <= 0 || w > ( 1 << 24 ) || h > ( 1 << 24 ) || static_cast< uint64_t >( w ) * h > ( 1ull << 28 ) )
    {
        error = "HDR: bad picture size";
        return false;
    }
    // every pixel takes at least one byte of file, run-length coded or not, except inside runs (128 per two bytes)
    if ( static_cast< uint64_t >( w ) * h / 64 > file.size() )
    {
        error = "HDR: file too short for its picture size";
        return false;
    }
    width  = static_cast< int >( w );
    height = static_cast< int >( h );
    pixels.assign( static_cast< size_t >( w ) * h * 4, 0.0f );

    auto flatFrom = [&]( size_t firstPixel ) {
        for ( size_t p = firstPixel; p < static_cast< size_t >( w ) * h; ++p )
        {
            uint8_t px[4];
            for ( uint8_t& b : px ) b = static_cast< uint8_t >( in.Byte() );
            Convert( px, &pixels[4 * p] );
        }
    };
    if ( w < 8 || w >= 32768 )
    {
        flatFrom( 0 );
        return true;
    }
    std::vector< uint8_t > scanline( static_cast< size_t >( w ) * 4 );
    for ( long j = 0; j < h; ++j )
    {
        const int c1 = in.Byte(), c2 = in.Byte();
        int length = in.Byte();
        if ( c1 != 2 || c2 != 2 || ( length & 0x80 ) )
        {
            // no run-length header: these four bytes are taken for the picture's first pixel and every other
            // pixel is read flat from here on, whichever row this happens in
            const uint8_t px[4] = { static_cast< uint8_t >( c1 ), static_cast< uint8_t >( c2 ), static_cast< uint8_t >( length ), static_cast< uint8_t >( in.Byte() ) };
            Convert( px, &pixels[0] );
            flatFrom( 1 );
            return true;
        }
        length = length << 8 | in.Byte();
        if ( length != w )
        {
            error = "HDR: invalid decoded scanline length";
            return false;
        }
        for ( int k = 0; k < 4; ++k )
        {
            long i = 0, left;
            while ( ( left = w - i ) > 0 )
            {
                int count = in.Byte();
                if ( count > 128 )
                {
                    const uint8_t value = static_cast< uint8_t >( in.Byte() );
                    count -= 128;
                    if ( count > left )
                    {
                        error = "HDR: bad run-length data";
                        return false;
                    }
                    for ( int z = 0; z < count; ++z ) scanline[4 * i++ + k] = value;
                }
                else
                {
                    if ( count > left || ( count == 0 && in.AtEnd() ) )
                    {
                        error = "HDR: bad run-length data";
                        return false;
                    }
                    for ( int z = 0; z < count; ++z ) scanline[4 * i++ + k] = static_cast< uint8_t >( in.Byte() );
                }
            }
        }
        for ( long i = 0; i < w; ++i ) Convert( &scanline[4 * i], &pixels[4 * ( static_cast< size_t >( j ) * w + i )] );
    }
    return true;
}

} // namespace pth
