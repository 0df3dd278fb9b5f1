#include "image_io.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>

namespace pth
{

namespace
{

    std::string Extension( const std::string& filename )
    {
        size_t dot = filename.find_last_of( '.' );
        if ( dot == std::string::npos ) return "";
        std::string e = filename.substr( dot + 1 );
        std::transform( e.begin(), e.end(), e.begin(), []( unsigned char c ) { return std::tolower( c ); } );
        return e;
    }

    bool ReadFile( const std::string& filename, std::vector< uint8_t >& data )
    {
        std::ifstream in( filename, std::ios::binary );
        if ( !in ) return false;
        in.seekg( 0, std::ios::end );
        std::streamoff n = in.tellg();
        in.seekg( 0 );
        data.resize( static_cast< size_t >( n ) );
        in.read( reinterpret_cast< char* >( data.data() ), n );
        return static_cast< bool >( in );
    }

    // Truevision TGA as the reference's decoder reads it (stb_image 2.25 behind Texture::Load,
    // src/resource/texture.cpp:19-33): true colour 24 / 32 bit, 15 / 16 bit as 5-5-5 without alpha (each field
    // times 255 / 31, truncated), grey 8 bit, grey with alpha 16 bit, colour-mapped with 8 or 16 bit indices into a
    // palette of any of those formats (an 8 bit palette entry is a grey level; an index past the palette means
    // entry 0; the "first entry" field counts bytes to skip before the palette), each raw or run-length coded with
    // packets that may run across rows; bottom row first unless bit 5 of the descriptor is set.
    bool ReadTga( const std::vector< uint8_t >& f, ImageRGBA8& out, std::string& error )
    {
        auto fail = [&]( const char* what ) { error = what; return false; };
        if ( f.size() < 18 ) return fail( "TGA header truncated" );
        const int idLength = f[0], indexed = f[1];
        int type = f[2];
        const int paletteStart = f[3] | ( f[4] << 8 ), paletteLength = f[5] | ( f[6] << 8 ), paletteBits = f[7];
        const int width = f[12] | ( f[13] << 8 ), height = f[14] | ( f[15] << 8 );
        const int bpp = f[16], descriptor = f[17];
        if ( indexed > 1 ) return fail( "bad TGA colour map type" );
        const bool rle = type >= 8;
        if ( rle ) type -= 8;
        if ( indexed ? type != 1 : ( type != 2 && type != 3 ) ) return fail( "unsupported TGA image type" );
        if ( indexed && bpp != 8 && bpp != 16 ) return fail( "unsupported TGA index size" );
        if ( width <= 0 || height <= 0 ) return fail( "empty TGA picture" );

        // 1: grey, 2: grey + alpha, 3: colour, 4: colour + alpha; 15 / 16 bit colour is unpacked to three bytes
        const int formatBits = indexed ? paletteBits : bpp;
        int channels = 0;
        bool packed16 = false;
        if ( formatBits == 8 ) channels = 1;
        else if ( formatBits == 16 && !indexed && type == 3 ) channels = 2;
        else if ( formatBits == 15 || formatBits == 16 ) { channels = 3; packed16 = true; }
        else if ( formatBits == 24 || formatBits == 32 ) channels = formatBits / 8;
        else return fail( "unsupported TGA pixel depth" );

        size_t pos = 18 + static_cast< size_t >( idLength );
        // reads one pixel or palette entry in the file's format into `to` (colour ends up as R, G, B[, A])
        auto readEntry = [&]( uint8_t* to ) -> bool
        {
            if ( packed16 )
            {
                if ( pos + 2 > f.size() ) return false;
                const int v = f[pos] | ( f[pos + 1] << 8 );
                pos += 2;
                to[0] = static_cast< uint8_t >( ( ( v >> 10 ) & 31 ) * 255 / 31 );
                to[1] = static_cast< uint8_t >( ( ( v >> 5 ) & 31 ) * 255 / 31 );
                to[2] = static_cast< uint8_t >( ( v & 31 ) * 255 / 31 );
                return true;
            }
            if ( pos + channels > f.size() ) return false;
            for ( int k = 0; k < channels; ++k ) to[k] = f[pos + k];
            if ( channels >= 3 ) std::swap( to[0], to[2] ); // stored blue first
            pos += channels;
            return true;
        };
        std::vector< uint8_t > palette;
        if ( indexed )
        {
            pos += paletteStart;
            palette.assign( static_cast< size_t >( paletteLength ) * channels + 4, 0 );
            for ( int i = 0; i < paletteLength; ++i )
                if ( !readEntry( &palette[static_cast< size_t >( i ) * channels] ) ) return fail( "TGA palette truncated" );
        }
        const size_t total = static_cast< size_t >( width ) * height;
        if ( total / 128 > f.size() ) return fail( "TGA data truncated" ); // a packet of two bytes or more yields at most 128 pixels
        out.width  = width;
        out.height = height;
        out.pixels.assign( total * 4, 255 );
        uint8_t raw[4]  = { 0, 0, 0, 0 };
        int packetLeft  = 0;
        bool repeating  = false, readNext = true;
        for ( size_t i = 0; i < total; ++i )
        {
            if ( rle )
            {
                if ( packetLeft == 0 )
                {
                    if ( pos >= f.size() ) return fail( "TGA data truncated" );
                    const int command = f[pos++];
                    packetLeft        = 1 + ( command & 127 );
                    repeating         = ( command >> 7 ) != 0;
                    readNext          = true;
                }
                else if ( !repeating )
                    readNext = true;
            }
            else
                readNext = true;
            if ( readNext )
            {
                if ( indexed )
                {
                    const int indexBytes = bpp / 8;
                    if ( pos + indexBytes > f.size() ) return fail( "TGA data truncated" );
                    int index = indexBytes == 1 ? f[pos] : f[pos] | ( f[pos + 1] << 8 );
                    pos += indexBytes;
                    if ( index >= paletteLength ) index = 0;
                    for ( int k = 0; k < channels; ++k ) raw[k] = palette[static_cast< size_t >( index ) * channels + k];
                }
                else if ( !readEntry( raw ) )
                    return fail( "TGA data truncated" );
                readNext = false;
            }
            uint8_t* o = &out.pixels[4 * i];
            if ( channels <= 2 )
            {
                o[0] = o[1] = o[2] = raw[0];
                if ( channels == 2 ) o[3] = raw[1];
            }
            else
            {
                o[0] = raw[0];
                o[1] = raw[1];
                o[2] = raw[2];
                if ( channels == 4 ) o[3] = raw[3];
            }
            --packetLeft;
        }
        const bool topOrigin = ( descriptor & 0x20 ) != 0;
        if ( !topOrigin ) // stored bottom row first: bring the top row to the front
        {
            for ( int r = 0; r < height / 2; ++r )
                std::swap_ranges( out.pixels.begin() + 4 * static_cast< size_t >( r ) * width,
                                  out.pixels.begin() + 4 * static_cast< size_t >( r + 1 ) * width,
                                  out.pixels.begin() + 4 * static_cast< size_t >( height - 1 - r ) * width );
        }
        return true;
    }

    bool ReadPnm( const std::vector< uint8_t >& f, ImageRGBA8& out, std::string& error )
    {
        size_t pos = 2;
        auto nextInt = [&]() -> int
        {
            while ( pos < f.size() )
            {
                if ( f[pos] == '#' ) while ( pos < f.size() && f[pos] != '\n' ) ++pos;
                else if ( std::isspace( f[pos] ) ) ++pos;
                else break;
            }
            int v = 0;
            while ( pos < f.size() && std::isdigit( f[pos] ) ) v = v * 10 + ( f[pos++] - '0' );
            return v;
        };
        const int channels = f[1] == '6' ? 3 : 1;
        const int w = nextInt(), h = nextInt(), maxv = nextInt();
        ++pos;
        if ( w <= 0 || h <= 0 || maxv > 255 || pos + static_cast< size_t >( w ) * h * channels > f.size() ) // samples are taken as stored, whatever the maximum says
        {
            error = "unsupported or truncated PNM";
            return false;
        }
        out.width  = w;
        out.height = h;
        out.pixels.resize( static_cast< size_t >( w ) * h * 4 );
        for ( size_t i = 0; i < static_cast< size_t >( w ) * h; ++i )
        {
            for ( int c = 0; c < 3; ++c ) out.pixels[4 * i + c] = f[pos + i * channels + ( channels == 3 ? c : 0 )];
            out.pixels[4 * i + 3] = 255;
        }
        return true;
    }

    // A `bits`-wide field that was moved so that its top bit is bit 7, widened to eight bits by repeating it
    // (stb_image's stbi__shiftsigned: multiply-and-shift tables per width).
    int WidenField( uint32_t v, int shift, int bits )
    {
        static const uint32_t multiplier[9] = { 0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01 };
        static const int shiftBack[9]       = { 0, 0, 0, 1, 0, 2, 4, 6, 0 };
        if ( shift < 0 ) v <<= -shift;
        else v >>= shift;
        v &= 0xff;
        v >>= ( 8 - bits );
        return static_cast< int >( v * multiplier[bits] ) >> shiftBack[bits];
    }

    int HighBit( uint32_t v )
    {
        int n = -1;
        for ( ; v; v >>= 1 ) ++n;
        return n;
    }

    int BitCount( uint32_t v )
    {
        int n = 0;
        for ( ; v; v >>= 1 ) n += v & 1;
        return n;
    }

    // Windows / OS2 bitmaps as the reference's decoder reads them (stb_image 2.25 behind Texture::Load,
    // src/resource/texture.cpp:19-33): core (12 byte), info (40, 56), V4 and V5 headers; 1, 4 and 8 bit with a
    // palette, 16 and 32 bit through channel masks (5-5-5 and 8-8-8-8 by default, BI_BITFIELDS otherwise), 24 bit;
    // no run-length compression; rows bottom-up unless the height is negative; a 32 bit file under a 40 or 56 byte
    // header whose alpha bytes are all zero is opaque.
    bool ReadBmp( const std::vector< uint8_t >& f, ImageRGBA8& out, std::string& error )
    {
        size_t pos = 0;
        auto u8  = [&]() -> uint32_t { return pos < f.size() ? f[pos++] : 0; };
        auto u16 = [&]() -> uint32_t { uint32_t lo = u8(); return lo | u8() << 8; };
        auto u32 = [&]() -> uint32_t { uint32_t lo = u16(); return lo | u16() << 16; };
        auto fail = [&]( const char* what ) { error = what; return false; };
        pos = 10;
        const int offset = static_cast< int >( u32() ), headerSize = static_cast< int >( u32() );
        if ( headerSize != 12 && headerSize != 40 && headerSize != 56 && headerSize != 108 && headerSize != 124 ) return fail( "unknown BMP header" );
        int w, hRaw;
        if ( headerSize == 12 ) { w = static_cast< int >( u16() ); hRaw = static_cast< int >( u16() ); }
        else { w = static_cast< int32_t >( u32() ); hRaw = static_cast< int32_t >( u32() ); }
        if ( u16() != 1 ) return fail( "bad BMP (planes)" );
        const int bpp = static_cast< int >( u16() );
        uint32_t maskR = 0, maskG = 0, maskB = 0, maskA = 0, alphaSeen = 255;
        int headerRead = 14;
        if ( headerSize != 12 )
        {
            const uint32_t compression = u32();
            if ( compression == 1 || compression == 2 ) return fail( "run-length compressed BMP is not decoded" );
            pos += 20; // image size, resolution, colour counts
            if ( headerSize == 40 || headerSize == 56 )
            {
                if ( headerSize == 56 ) pos += 16;
                if ( bpp == 16 || bpp == 32 )
                {
                    if ( compression == 0 )
                    {
                        if ( bpp == 32 ) { maskR = 0xffu << 16; maskG = 0xffu << 8; maskB = 0xffu; maskA = 0xffu << 24; alphaSeen = 0; }
                        else { maskR = 31u << 10; maskG = 31u << 5; maskB = 31u; }
                    }
                    else if ( compression == 3 )
                    {
                        maskR = u32(); maskG = u32(); maskB = u32();
                        headerRead += 12;
                        if ( maskR == maskG && maskG == maskB ) return fail( "bad BMP (masks)" );
                    }
                    else return fail( "bad BMP (compression)" );
                }
            }
            else
            {
                maskR = u32(); maskG = u32(); maskB = u32(); maskA = u32();
                pos += 4 + 48 + ( headerSize == 124 ? 16 : 0 );
            }
        }
        const bool bottomUp = hRaw > 0;
        const int h = hRaw < 0 ? -hRaw : hRaw;
        if ( w <= 0 || h <= 0 || w > ( 1 << 24 ) || h > ( 1 << 24 ) ) return fail( "bad BMP size" );
        int paletteSize = 0;
        if ( headerSize == 12 ) { if ( bpp < 24 ) paletteSize = ( offset - headerRead - 12 ) / 3; } // three-byte entries
        else if ( bpp < 16 ) paletteSize = ( offset - headerRead - headerSize ) >> 2;
        if ( bpp != 1 && bpp != 4 && bpp != 8 && bpp != 16 && bpp != 24 && bpp != 32 ) return fail( "bad BMP bit depth" );
        const size_t rowBytes = ( ( static_cast< size_t >( w ) * bpp + 7 ) / 8 + 3 ) & ~size_t( 3 );
        if ( offset < 0 || static_cast< size_t >( offset ) > f.size() || rowBytes * h > f.size() - static_cast< size_t >( offset ) ) return fail( "BMP data truncated" );
        out.width  = w;
        out.height = h;
        out.pixels.assign( static_cast< size_t >( w ) * h * 4, 255 );
        uint8_t* o = out.pixels.data();
        if ( bpp < 16 )
        {
            if ( paletteSize <= 0 || paletteSize > 256 ) return fail( "bad BMP palette" );
            uint8_t palette[256][3] = {};
            for ( int i = 0; i < paletteSize; ++i )
            {
                palette[i][2] = static_cast< uint8_t >( u8() );
                palette[i][1] = static_cast< uint8_t >( u8() );
                palette[i][0] = static_cast< uint8_t >( u8() );
                if ( headerSize != 12 ) u8();
            }
            for ( int j = 0; j < h; ++j )
            {
                const uint8_t* row = &f[offset + rowBytes * j];
                for ( int i = 0; i < w; ++i, o += 4 )
                {
                    int index;
                    if ( bpp == 1 ) index = ( row[i >> 3] >> ( 7 - ( i & 7 ) ) ) & 1;
                    else if ( bpp == 4 ) index = ( i & 1 ) ? row[i >> 1] & 15 : row[i >> 1] >> 4;
                    else index = row[i];
                    o[0] = palette[index][0];
                    o[1] = palette[index][1];
                    o[2] = palette[index][2];
                }
            }
        }
        else
        {
            const bool plain = bpp == 24 || ( bpp == 32 && maskB == 0xff && maskG == 0xff00 && maskR == 0x00ff0000 && maskA == 0xff000000 );
            int shiftR = 0, shiftG = 0, shiftB = 0, shiftA = 0, bitsR = 0, bitsG = 0, bitsB = 0, bitsA = 0;
            if ( !plain )
            {
                if ( !maskR || !maskG || !maskB ) return fail( "bad BMP (masks)" );
                shiftR = HighBit( maskR ) - 7; bitsR = BitCount( maskR );
                shiftG = HighBit( maskG ) - 7; bitsG = BitCount( maskG );
                shiftB = HighBit( maskB ) - 7; bitsB = BitCount( maskB );
                shiftA = HighBit( maskA ) - 7; bitsA = BitCount( maskA );
                if ( bitsR > 8 || bitsG > 8 || bitsB > 8 || bitsA > 8 ) return fail( "bad BMP (masks)" );
            }
            for ( int j = 0; j < h; ++j )
            {
                const uint8_t* p = &f[offset + rowBytes * j];
                for ( int i = 0; i < w; ++i, o += 4 )
                {
                    if ( plain )
                    {
                        o[2] = p[0];
                        o[1] = p[1];
                        o[0] = p[2];
                        o[3] = bpp == 32 ? p[3] : 255;
                        p += bpp / 8;
                    }
                    else
                    {
                        uint32_t v = p[0] | p[1] << 8;
                        if ( bpp == 32 ) v |= static_cast< uint32_t >( p[2] ) << 16 | static_cast< uint32_t >( p[3] ) << 24;
                        p += bpp / 8;
                        o[0] = static_cast< uint8_t >( WidenField( v & maskR, shiftR, bitsR ) );
                        o[1] = static_cast< uint8_t >( WidenField( v & maskG, shiftG, bitsG ) );
                        o[2] = static_cast< uint8_t >( WidenField( v & maskB, shiftB, bitsB ) );
                        o[3] = static_cast< uint8_t >( maskA ? WidenField( v & maskA, shiftA, bitsA ) : 255 );
                    }
                    alphaSeen |= o[3];
                }
            }
        }
        if ( alphaSeen == 0 )
            for ( size_t i = 3; i < out.pixels.size(); i += 4 ) out.pixels[i] = 255;
        if ( bottomUp )
        {
            const size_t row = static_cast< size_t >( w ) * 4;
            for ( int r = 0; r < h / 2; ++r )
                std::swap_ranges( out.pixels.begin() + row * r, out.pixels.begin() + row * ( r + 1 ), out.pixels.begin() + row * ( h - 1 - r ) );
        }
        return true;
    }

    uint32_t Crc32( const uint8_t* p, size_t n, uint32_t crc = 0 )
    {
        static uint32_t table[256];
        static bool init = false;
        if ( !init )
        {
            for ( uint32_t i = 0; i < 256; ++i )
            {
                uint32_t c = i;
                for ( int k = 0; k < 8; ++k ) c = ( c & 1 ) ? 0xEDB88320u ^ ( c >> 1 ) : c >> 1;
                table[i] = c;
            }
            init = true;
        }
        crc = ~crc;
        for ( size_t i = 0; i < n; ++i ) crc = table[( crc ^ p[i] ) & 0xff] ^ ( crc >> 8 );
        return ~crc;
    }

    void PutBE32( std::vector< uint8_t >& v, uint32_t x )
    {
        v.push_back( x >> 24 );
        v.push_back( x >> 16 );
        v.push_back( x >> 8 );
        v.push_back( x );
    }

    void PngChunk( std::vector< uint8_t >& file, const char* tag, const std::vector< uint8_t >& data )
    {
        PutBE32( file, static_cast< uint32_t >( data.size() ) );
        std::vector< uint8_t > body( tag, tag + 4 );
        body.insert( body.end(), data.begin(), data.end() );
        file.insert( file.end(), body.begin(), body.end() );
        PutBE32( file, Crc32( body.data(), body.size() ) );
    }

    // ---- PNG as the reference writes it (stb_image_write 1.14 behind Image::Save, src/image.cpp:43-100), so that a
    // rendered picture is the same file byte for byte: per row the filter whose output has the smallest sum of
    // absolute values (first wins a tie; on the first row "up" degenerates to "none" and "average" / "Paeth" to
    // their left-only forms, the filter byte still says what was asked for), then one fixed-Huffman deflate block
    // whose matches come from 16,384 hash buckets of at most 16 positions (the older 8 dropped when full), the
    // longest match winning and the latest of equal ones, deferred by one byte when the next position has a longer.

    int PaethPredictor( int a, int b, int c )
    {
        const int p = a + b - c, pa = std::abs( p - a ), pb = std::abs( p - b ), pc = std::abs( p - c );
        return ( pa <= pb && pa <= pc ) ? a : ( pb <= pc ? b : c );
    }

    void FilterRow( const uint8_t* row, const uint8_t* above, int bytes, int filter, int8_t* out )
    {
        // operation: 0 none, 1 sub, 2 up, 3 average, 4 Paeth; 5 / 6 = average / Paeth of a first row
        const int op = above ? filter : ( filter == 2 ? 0 : ( filter == 3 ? 5 : ( filter == 4 ? 6 : filter ) ) );
        for ( int i = 0; i < bytes; ++i )
        {
            const int left = i >= 3 ? row[i - 3] : 0, up = above ? above[i] : 0, upLeft = ( above && i >= 3 ) ? above[i - 3] : 0;
            int predicted = 0;
            switch ( op )
            {
                case 1: predicted = left; break;
                case 2: predicted = up; break;
                case 3: predicted = ( left + up ) >> 1; break;
                case 4: predicted = PaethPredictor( left, up, upLeft ); break;
                case 5: predicted = left >> 1; break;
                case 6: predicted = left; break;
                default: break;
            }
            out[i] = static_cast< int8_t >( static_cast< uint8_t >( row[i] - predicted ) );
        }
    }

    class FixedDeflater
    {
    public:
        std::vector< uint8_t > out;

        void Compress( const uint8_t* data, int size, int quality )
        {
            static const uint16_t lengthBase[30] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 259 };
            static const uint8_t lengthExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
            static const uint16_t distanceBase[31] = { 1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,  193,
                                                       257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 32768 };
            static const uint8_t distanceExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
            constexpr int kBuckets = 16384;
            std::vector< std::vector< int > > buckets( kBuckets );
            out = { 0x78, 0x5e };
            Add( 1, 1 ); // last block
            Add( 1, 2 ); // fixed Huffman codes
            int i = 0;
            while ( i < size - 3 )
            {
                int best = 3, bestAt = -1;
                std::vector< int >& bucket = buckets[Hash( data + i ) & ( kBuckets - 1 )];
                for ( int at : bucket )
                    if ( at > i - 32768 )
                    {
                        const int n = MatchLength( data + at, data + i, size - i );
                        if ( n >= best )
                        {
                            best   = n;
                            bestAt = at;
                        }
                    }
                if ( static_cast< int >( bucket.size() ) == 2 * quality ) bucket.erase( bucket.begin(), bucket.begin() + quality );
                bucket.push_back( i );
                if ( bestAt >= 0 )
                    for ( int at : buckets[Hash( data + i + 1 ) & ( kBuckets - 1 )] )
                        if ( at > i - 32767 && MatchLength( data + at, data + i + 1, size - i - 1 ) > best )
                        {
                            bestAt = -1; // the next byte starts a longer match: this one goes out as a literal
                            break;
                        }
                if ( bestAt >= 0 )
                {
                    const int distance = i - bestAt;
                    int j = 0;
                    while ( best > lengthBase[j + 1] - 1 ) ++j;
                    Symbol( j + 257 );
                    if ( lengthExtra[j] ) Add( static_cast< uint32_t >( best - lengthBase[j] ), lengthExtra[j] );
                    j = 0;
                    while ( distance > distanceBase[j + 1] - 1 ) ++j;
                    Add( Reverse( j, 5 ), 5 );
                    if ( distanceExtra[j] ) Add( static_cast< uint32_t >( distance - distanceBase[j] ), distanceExtra[j] );
                    i += best;
                }
                else
                    Symbol( data[i++] );
            }
            for ( ; i < size; ++i ) Symbol( data[i] );
            Symbol( 256 );
            while ( m_count ) Add( 0, 1 );
            uint32_t a = 1, b = 0;
            for ( int k = 0; k < size; ++k )
            {
                a = ( a + data[k] ) % 65521;
                b = ( b + a ) % 65521;
            }
            for ( uint32_t v : { b >> 8, b, a >> 8, a } ) out.push_back( static_cast< uint8_t >( v ) );
        }

    private:
        uint32_t m_bits = 0;
        int m_count     = 0;

        void Add( uint32_t code, int bits )
        {
            m_bits |= code << m_count;
            m_count += bits;
            while ( m_count >= 8 )
            {
                out.push_back( static_cast< uint8_t >( m_bits ) );
                m_bits >>= 8;
                m_count -= 8;
            }
        }

        static uint32_t Reverse( int code, int bits )
        {
            uint32_t r = 0;
            while ( bits-- )
            {
                r = r << 1 | ( code & 1 );
                code >>= 1;
            }
            return r;
        }

        void Symbol( int n ) // literal / length alphabet of the fixed code
        {
            if ( n <= 143 ) Add( Reverse( 0x30 + n, 8 ), 8 );
            else if ( n <= 255 ) Add( Reverse( 0x190 + n - 144, 9 ), 9 );
            else if ( n <= 279 ) Add( Reverse( n - 256, 7 ), 7 );
            else Add( Reverse( 0xc0 + n - 280, 8 ), 8 );
        }

        static int MatchLength( const uint8_t* a, const uint8_t* b, int limit )
        {
            int n = 0;
            while ( n < limit && n < 258 && a[n] == b[n] ) ++n;
            return n;
        }

        static uint32_t Hash( const uint8_t* d )
        {
            uint32_t h = d[0] + ( d[1] << 8 ) + ( d[2] << 16 );
            h ^= h << 3;
            h += h >> 5;
            h ^= h << 4;
            h += h >> 17;
            h ^= h << 25;
            h += h >> 6;
            return h;
        }
    };

    std::vector< uint8_t > EncodePng( const uint8_t* rgb, int w, int h )
    {
        std::vector< uint8_t > file = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
        std::vector< uint8_t > ihdr;
        PutBE32( ihdr, w );
        PutBE32( ihdr, h );
        ihdr.insert( ihdr.end(), { 8, 2, 0, 0, 0 } );
        PngChunk( file, "IHDR", ihdr );

        const int bytes = 3 * w;
        std::vector< uint8_t > raw( static_cast< size_t >( h ) * ( bytes + 1 ) );
        std::vector< int8_t > line( bytes ), bestLine( bytes );
        for ( int r = 0; r < h; ++r )
        {
            const uint8_t* row   = rgb + static_cast< size_t >( r ) * bytes;
            const uint8_t* above = r ? row - bytes : nullptr;
            int best = 0, bestSum = 0x7fffffff;
            for ( int filter = 0; filter < 5; ++filter )
            {
                FilterRow( row, above, bytes, filter, line.data() );
                int sum = 0;
                for ( int i = 0; i < bytes; ++i ) sum += std::abs( static_cast< int >( line[i] ) );
                if ( sum < bestSum )
                {
                    bestSum = sum;
                    best    = filter;
                    bestLine.swap( line );
                }
            }
            uint8_t* o = &raw[static_cast< size_t >( r ) * ( bytes + 1 )];
            o[0]       = static_cast< uint8_t >( best );
            std::memcpy( o + 1, bestLine.data(), bytes );
        }
        FixedDeflater deflater;
        deflater.Compress( raw.data(), static_cast< int >( raw.size() ), 8 );
        PngChunk( file, "IDAT", deflater.out );
        PngChunk( file, "IEND", {} );
        return file;
    }

} // namespace

bool ReadImageRGBA8( const std::string& filename, bool flipVertically, ImageRGBA8& out, std::string& error )
{
    std::vector< uint8_t > f;
    if ( !ReadFile( filename, f ) )
    {
        error = "Failed to load image '" + filename + "'";
        return false;
    }
    bool ok = false;
    const std::string ext = Extension( filename );
    if ( f.size() > 2 && f[0] == 'P' && ( f[1] == '5' || f[1] == '6' ) ) ok = ReadPnm( f, out, error );
    else if ( f.size() > 2 && f[0] == 'B' && f[1] == 'M' ) ok = ReadBmp( f, out, error );
    else if ( f.size() > 2 && f[0] == 0xFF && f[1] == 0xD8 ) ok = ReadJpegRGBA8( filename, out, error );
    else if ( f.size() > 4 && f[0] == 0x89 && f[1] == 'P' && f[2] == 'N' && f[3] == 'G' ) ok = ReadPngRGBA8( f, out, error );
    else if ( IsRadianceFile( f ) )
    {
        // a Radiance picture asked for as bytes: the reference's decoder applies gamma 1 / 2.2 and rounds
        std::vector< float > radiance;
        ok = ReadRadianceRGBA32F( f, out.width, out.height, radiance, error );
        if ( ok )
        {
            out.pixels.resize( radiance.size() );
            const float inverseGamma = 1.0f / 2.2f, scale = 1.0f;
            for ( size_t i = 0; i < radiance.size(); ++i )
            {
                float z = ( i & 3 ) == 3 ? radiance[i] * 255 + 0.5f
                                         : static_cast< float >( std::pow( static_cast< double >( radiance[i] * scale ), static_cast< double >( inverseGamma ) ) ) * 255 + 0.5f;
                z = z < 0 ? 0 : ( z > 255 ? 255 : z );
                out.pixels[i] = static_cast< uint8_t >( static_cast< int >( z ) );
            }
        }
    }
    else if ( ext == "tga" ) ok = ReadTga( f, out, error );
    else error = "image format of '" + filename + "' is not decoded by this host (TGA, PNG, JPEG, BMP, PPM, HDR only)";
    if ( !ok )
    {
        error = "Failed to load image '" + filename + "': " + error;
        return false;
    }
    if ( flipVertically )
    {
        const size_t row = static_cast< size_t >( out.width ) * 4;
        for ( int r = 0; r < out.height / 2; ++r )
            std::swap_ranges( out.pixels.begin() + row * r, out.pixels.begin() + row * ( r + 1 ),
                              out.pixels.begin() + row * ( out.height - 1 - r ) );
    }
    return true;
}

bool ReadImageRGBA32F( const std::string& filename, bool flipVertically, int& width, int& height, std::vector< float >& pixels, std::string& error )
{
    std::vector< uint8_t > f;
    if ( !ReadFile( filename, f ) )
    {
        error = "Failed to load image '" + filename + "'";
        return false;
    }
    if ( !IsRadianceFile( f ) )
    {
        f.clear();
        ImageRGBA8 bytes;
        if ( !ReadImageRGBA8( filename, flipVertically, bytes, error ) ) return false;
        width  = bytes.width;
        height = bytes.height;
        pixels.resize( bytes.pixels.size() );
        for ( size_t i = 0; i < pixels.size(); ++i ) pixels[i] = bytes.pixels[i] / 255.0f;
        return true;
    }
    if ( !ReadRadianceRGBA32F( f, width, height, pixels, error ) )
    {
        error = "Failed to load image '" + filename + "': " + error;
        return false;
    }
    if ( flipVertically )
    {
        const size_t row = static_cast< size_t >( width ) * 4;
        for ( int r = 0; r < height / 2; ++r )
            std::swap_ranges( pixels.begin() + row * r, pixels.begin() + row * ( r + 1 ), pixels.begin() + row * ( height - 1 - r ) );
    }
    return true;
}

namespace
{
    // Uncharted2TonemapHelper, src/tonemap.cpp:36-45
    float FilmicCurve( float x )
    {
        const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
        return ( ( x * ( A * x + C * B ) + D * E ) / ( x * ( A * x + B ) + D * F ) ) - E / F;
    }
} // namespace

void TonemapRGB( float* rgb, size_t count, float exposure, float gamma, int op )
{
    const float whiteScale = 1.0f / FilmicCurve( 11.2f ), inverseGamma = 1.0f / gamma;
    for ( size_t i = 0; i < count; ++i )
    {
        float x = rgb[i];
        if ( op == 0 )
            x = FilmicCurve( exposure * x ) * whiteScale;
        else
        {
            const float adjusted = exposure * x;
            x                    = adjusted / ( 1.0f + adjusted );
        }
        x      = std::pow( x, inverseGamma ); // float overload: the C library's powf, as glm::pow calls it
        x      = x < 0.0f ? 0.0f : x;         // glm::clamp = min( max( x, 0 ), 1 )
        rgb[i] = 1.0f < x ? 1.0f : x;
    }
}

std::vector< uint8_t > QuantizeRGB8( const float* rgb, int width, int height )
{
    std::vector< uint8_t > out( static_cast< size_t >( width ) * height * 3 );
    for ( size_t i = 0; i < out.size(); ++i )
    {
        float c = rgb[i];
        c       = c < 0.0f ? 0.0f : c;  // glm::clamp = min(max(x, 0), 1)
        c       = 1.0f < c ? 1.0f : c;
        out[i]  = static_cast< uint8_t >( 255.0f * c );
    }
    return out;
}

bool WriteImageRGB8( const std::string& filename, const uint8_t* rgb, int width, int height, std::string& error )
{
    const std::string ext = Extension( filename );
    std::vector< uint8_t > file;
    if ( ext == "png" )
        file = EncodePng( rgb, width, height );
    else if ( ext == "jpg" || ext == "jpeg" )
        file = EncodeJpegRGB8( rgb, width, height );
    else if ( ext == "ppm" )
    {
        std::string header = "P6\n" + std::to_string( width ) + " " + std::to_string( height ) + "\n255\n";
        file.assign( header.begin(), header.end() );
        file.insert( file.end(), rgb, rgb + static_cast< size_t >( width ) * height * 3 );
    }
    else if ( ext == "tga" )
    {
        // Run-length encoded true colour (type 10), bottom row first, as the reference's writer emits it
        // (stb_image_write 1.14 behind Image::Save, src/image.cpp:43-100).  Packets never cross a row.  A run
        // packet repeats one pixel up to 128 times.  A raw packet starts when the first two pixels differ and is
        // extended pixel by pixel while pixel k differs from pixel k - 2 (sic: two back, not one); when they are
        // equal the packet ends one pixel early, so that pixel k - 1 can open the next packet.
        file.assign( 18, 0 );
        file[2]  = 10;
        file[12] = width & 0xff;
        file[13] = ( width >> 8 ) & 0xff;
        file[14] = height & 0xff;
        file[15] = ( height >> 8 ) & 0xff;
        file[16] = 24;
        auto same = [&]( const uint8_t* a, const uint8_t* b ) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
        auto put  = [&]( const uint8_t* p ) { file.insert( file.end(), { p[2], p[1], p[0] } ); };
        for ( int r = height - 1; r >= 0; --r )
        {
            const uint8_t* row = rgb + 3 * static_cast< size_t >( r ) * width;
            int length = 1;
            for ( int i = 0; i < width; i += length )
            {
                const uint8_t* begin = row + 3 * i;
                bool run = false;
                length   = 1;
                if ( i < width - 1 )
                {
                    length = 2;
                    run    = same( begin, begin + 3 );
                    if ( !run )
                    {
                        const uint8_t* twoBack = begin;
                        for ( int k = i + 2; k < width && length < 128; ++k )
                        {
                            if ( same( twoBack, row + 3 * k ) )
                            {
                                --length;
                                break;
                            }
                            twoBack += 3;
                            ++length;
                        }
                    }
                    else
                        for ( int k = i + 2; k < width && length < 128 && same( begin, row + 3 * k ); ++k ) ++length;
                }
                if ( run )
                {
                    file.push_back( static_cast< uint8_t >( length - 129 ) );
                    put( begin );
                }
                else
                {
                    file.push_back( static_cast< uint8_t >( length - 1 ) );
                    for ( int k = 0; k < length; ++k ) put( begin + 3 * k );
                }
            }
        }
    }
    else if ( ext == "bmp" )
    {
        // 24 bit, bottom row first, rows padded to four bytes, the size-of-image and resolution fields left zero
        const size_t stride = ( static_cast< size_t >( width ) * 3 + 3 ) & ~size_t( 3 );
        const uint32_t size = static_cast< uint32_t >( 54 + stride * height );
        file.assign( 54, 0 );
        file[0] = 'B';
        file[1] = 'M';
        auto put32 = [&]( size_t o, uint32_t v ) { for ( int k = 0; k < 4; ++k ) file[o + k] = ( v >> ( 8 * k ) ) & 0xff; };
        put32( 2, size );
        put32( 10, 54 );
        put32( 14, 40 );
        put32( 18, width );
        put32( 22, height );
        file[26] = 1;
        file[28] = 24;
        for ( int r = height - 1; r >= 0; --r )
        {
            size_t start = file.size();
            for ( int c = 0; c < width; ++c )
            {
                const uint8_t* p = rgb + 3 * ( static_cast< size_t >( r ) * width + c );
                file.insert( file.end(), { p[2], p[1], p[0] } );
            }
            file.resize( start + stride, 0 );
        }
    }
    else
    {
        error = "Saving image as filetype '" + ext + "' is not supported";
        return false;
    }
    std::ofstream out( filename, std::ios::binary );
    if ( !out || !out.write( reinterpret_cast< const char* >( file.data() ), file.size() ) )
    {
        error = "Failed to write image: '" + filename + "'";
        return false;
    }
    return true;
}

} // namespace pth
