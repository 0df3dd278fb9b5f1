// PNG to RGBA8 for textures (the reference loads them through stb_image 2.25: Texture::Load,
// src/resource/texture.cpp:19-33).  PNG is lossless; what has to agree with the reference's decoder is how other
// pixel formats become 8-bit RGBA: 16-bit samples keep their high byte, grey samples of 1 / 2 / 4 bits are
// spread over 0..255 (x255, x85, x17), palette entries are used as they are, a tRNS colour key makes exactly
// the matching pixels transparent (compared before any depth conversion), grey becomes r = g = b, and gamma,
// colour profile and background chunks are ignored.  All colour types, bit depths and Adam7 interlacing.
#include "image_io.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

namespace pth
{

namespace
{

    // ---- inflate (RFC 1951) of a zlib stream (RFC 1950)

    class Inflater
    {
    public:
        Inflater( const std::vector< uint8_t >& in, std::vector< uint8_t >& out ) : m_in( in ), m_out( out ) {}

        bool Run( std::string& error )
        {
            if ( m_in.size() < 2 || ( m_in[0] & 15 ) != 8 || ( ( m_in[0] << 8 | m_in[1] ) % 31 ) != 0 || ( m_in[1] & 32 ) )
            {
                error = "bad zlib header";
                return false;
            }
            m_pos = 2;
            bool last = false;
            while ( !last )
            {
                last           = Bits( 1 ) != 0;
                const int type = Bits( 2 );
                bool ok        = false;
                if ( type == 0 ) ok = Stored();
                else if ( type == 1 ) ok = FixedBlock();
                else if ( type == 2 ) ok = DynamicBlock();
                if ( !ok || m_overrun )
                {
                    error = "corrupt deflate stream";
                    return false;
                }
            }
            return true;
        }

    private:
        struct Codes
        {
            uint16_t count[16]   = {};
            uint16_t symbol[288] = {};
        };

        const std::vector< uint8_t >& m_in;
        std::vector< uint8_t >& m_out;
        size_t m_pos = 0;
        uint32_t m_hold = 0;
        int m_held      = 0;
        bool m_overrun  = false;

        int Bits( int n )
        {
            while ( m_held < n )
            {
                if ( m_pos >= m_in.size() )
                {
                    m_overrun = true;
                    return 0;
                }
                m_hold |= static_cast< uint32_t >( m_in[m_pos++] ) << m_held;
                m_held += 8;
            }
            const int v = static_cast< int >( m_hold & ( ( 1u << n ) - 1 ) );
            m_hold >>= n;
            m_held -= n;
            return v;
        }

        static bool Build( Codes& c, const uint8_t* lengths, int n )
        {
            std::memset( c.count, 0, sizeof( c.count ) );
            for ( int i = 0; i < n; ++i ) c.count[lengths[i]]++;
            c.count[0] = 0;
            int left = 1;
            for ( int len = 1; len < 16; ++len )
            {
                left = ( left << 1 ) - c.count[len];
                if ( left < 0 ) return false; // over-subscribed
            }
            uint16_t offset[16] = {};
            for ( int len = 1; len < 15; ++len ) offset[len + 1] = static_cast< uint16_t >( offset[len] + c.count[len] );
            for ( int i = 0; i < n; ++i )
                if ( lengths[i] ) c.symbol[offset[lengths[i]]++] = static_cast< uint16_t >( i );
            return true;
        }

        int Decode( const Codes& c )
        {
            int code = 0, first = 0, index = 0;
            for ( int len = 1; len < 16; ++len )
            {
                code |= Bits( 1 );
                const int count = c.count[len];
                if ( code - count < first ) return c.symbol[index + ( code - first )];
                index += count;
                first = ( first + count ) << 1;
                code <<= 1;
                if ( m_overrun ) return -1;
            }
            return -1;
        }

        bool Stored()
        {
            m_hold = 0;
            m_held = 0;
            if ( m_pos + 4 > m_in.size() ) return false;
            const unsigned len = m_in[m_pos] | m_in[m_pos + 1] << 8, inverse = m_in[m_pos + 2] | m_in[m_pos + 3] << 8;
            m_pos += 4;
            if ( len != ( ~inverse & 0xffffu ) || m_pos + len > m_in.size() ) return false;
            m_out.insert( m_out.end(), m_in.begin() + static_cast< std::ptrdiff_t >( m_pos ), m_in.begin() + static_cast< std::ptrdiff_t >( m_pos + len ) );
            m_pos += len;
            return true;
        }

        bool Body( const Codes& lengthCodes, const Codes& distanceCodes )
        {
            static const uint16_t lengthBase[29]   = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
            static const uint16_t lengthExtra[29]  = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
            static const uint16_t distanceBase[30] = { 1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                                       193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
            static const uint16_t distanceExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
            for ( ;; )
            {
                int symbol = Decode( lengthCodes );
                if ( symbol < 0 ) return false;
                if ( symbol < 256 )
                {
                    m_out.push_back( static_cast< uint8_t >( symbol ) );
                    continue;
                }
                if ( symbol == 256 ) return true;
                symbol -= 257;
                if ( symbol >= 29 ) return false;
                const int length = lengthBase[symbol] + Bits( lengthExtra[symbol] );
                const int d      = Decode( distanceCodes );
                if ( d < 0 || d >= 30 ) return false;
                const size_t distance = distanceBase[d] + static_cast< size_t >( Bits( distanceExtra[d] ) );
                if ( distance > m_out.size() ) return false;
                for ( int k = 0; k < length; ++k ) m_out.push_back( m_out[m_out.size() - distance] );
                if ( m_overrun ) return false;
            }
        }

        bool FixedBlock()
        {
            uint8_t lengths[288];
            for ( int i = 0; i < 288; ++i ) lengths[i] = i < 144 ? 8 : ( i < 256 ? 9 : ( i < 280 ? 7 : 8 ) );
            Codes lit, dist;
            Build( lit, lengths, 288 );
            for ( int i = 0; i < 30; ++i ) lengths[i] = 5;
            Build( dist, lengths, 30 );
            return Body( lit, dist );
        }

        bool DynamicBlock()
        {
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            const int nLit = Bits( 5 ) + 257, nDist = Bits( 5 ) + 1, nCode = Bits( 4 ) + 4;
            if ( nLit > 286 || nDist > 30 ) return false;
            uint8_t lengths[320] = {};
            for ( int i = 0; i < nCode; ++i ) lengths[order[i]] = static_cast< uint8_t >( Bits( 3 ) );
            Codes codeCodes;
            if ( !Build( codeCodes, lengths, 19 ) ) return false;
            std::memset( lengths, 0, sizeof( lengths ) );
            for ( int i = 0; i < nLit + nDist; )
            {
                const int symbol = Decode( codeCodes );
                if ( symbol < 0 ) return false;
                if ( symbol < 16 )
                {
                    lengths[i++] = static_cast< uint8_t >( symbol );
                    continue;
                }
                int repeat, value = 0;
                if ( symbol == 16 )
                {
                    if ( i == 0 ) return false;
                    value  = lengths[i - 1];
                    repeat = 3 + Bits( 2 );
                }
                else if ( symbol == 17 )
                    repeat = 3 + Bits( 3 );
                else
                    repeat = 11 + Bits( 7 );
                if ( i + repeat > nLit + nDist ) return false;
                while ( repeat-- ) lengths[i++] = static_cast< uint8_t >( value );
            }
            if ( lengths[256] == 0 ) return false;
            Codes lit, dist;
            if ( !Build( lit, lengths, nLit ) || !Build( dist, lengths + nLit, nDist ) ) return false;
            return Body( lit, dist );
        }
    };

    // ---- PNG

    uint32_t Big32( const uint8_t* p ) { return static_cast< uint32_t >( p[0] ) << 24 | p[1] << 16 | p[2] << 8 | p[3]; }

    int Paeth( int a, int b, int c )
    {
        const int p = a + b - c, pa = std::abs( p - a ), pb = std::abs( p - b ), pc = std::abs( p - c );
        return ( pa <= pb && pa <= pc ) ? a : ( pb <= pc ? b : c );
    }

    struct Header
    {
        uint32_t width = 0, height = 0;
        int depth = 0, colourType = 0, interlace = 0, channels = 0;
    };

    // Undoes the row filters of one (sub)image of w x h pixels in `raw` (one filter byte and the packed row per
    // line) and hands every pixel to `put` as `channels` samples of 8 or 16 bits, lower depths unpacked unscaled.
    template < typename Put >
    bool Unfilter( const Header& hd, const uint8_t* raw, size_t rawSize, uint32_t w, uint32_t h, Put put, size_t* consumed )
    {
        const size_t bitsPerPixel = static_cast< size_t >( hd.depth ) * hd.channels;
        const size_t rowBytes = ( w * bitsPerPixel + 7 ) / 8, step = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
        if ( rawSize < ( rowBytes + 1 ) * h ) return false;
        std::vector< uint8_t > previous( rowBytes, 0 ), current( rowBytes );
        std::vector< uint16_t > samples( hd.channels );
        for ( uint32_t y = 0; y < h; ++y )
        {
            const uint8_t* line = raw + ( rowBytes + 1 ) * y;
            const int filter    = line[0];
            if ( filter > 4 ) return false;
            for ( size_t i = 0; i < rowBytes; ++i )
            {
                const int a = i >= step ? current[i - step] : 0, b = previous[i], c = i >= step ? previous[i - step] : 0;
                int predicted = 0;
                if ( filter == 1 ) predicted = a;
                else if ( filter == 2 ) predicted = b;
                else if ( filter == 3 ) predicted = ( a + b ) >> 1;
                else if ( filter == 4 ) predicted = Paeth( a, b, c );
                current[i] = static_cast< uint8_t >( line[1 + i] + predicted );
            }
            for ( uint32_t x = 0; x < w; ++x )
            {
                for ( int k = 0; k < hd.channels; ++k )
                {
                    const size_t s = static_cast< size_t >( x ) * hd.channels + k;
                    if ( hd.depth == 16 ) samples[k] = static_cast< uint16_t >( current[2 * s] << 8 | current[2 * s + 1] );
                    else if ( hd.depth == 8 ) samples[k] = current[s];
                    else
                    {
                        const size_t bit = s * hd.depth;
                        samples[k] = static_cast< uint16_t >( ( current[bit / 8] >> ( 8 - hd.depth - bit % 8 ) ) & ( ( 1 << hd.depth ) - 1 ) );
                    }
                }
                put( x, y, samples.data() );
            }
            previous.swap( current );
        }
        *consumed = ( rowBytes + 1 ) * h;
        return true;
    }

    bool Refuse( std::string& error, const char* what )
    {
        error = std::string( "PNG: " ) + what;
        return false;
    }

} // namespace

bool ReadPngRGBA8( const std::vector< uint8_t >& file, ImageRGBA8& out, std::string& error )
{
    static const uint8_t signature[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if ( file.size() < 8 + 25 || std::memcmp( file.data(), signature, 8 ) != 0 ) return Refuse( error, "bad signature" );
    Header hd;
    std::vector< uint8_t > compressed, palette( 256 * 4, 255 );
    int paletteEntries = 0;
    bool haveKey = false, first = true;
    uint16_t key[3] = {};
    for ( size_t pos = 8; pos + 12 <= file.size(); )
    {
        const uint32_t length = Big32( &file[pos] );
        const uint8_t* type   = &file[pos + 4];
        const uint8_t* data   = &file[pos + 8];
        if ( length > file.size() - pos - 12 ) return Refuse( error, "chunk runs past the end of the file" );
        if ( first && std::memcmp( type, "IHDR", 4 ) != 0 ) return Refuse( error, "first chunk is not IHDR" );
        first = false;
        if ( std::memcmp( type, "IHDR", 4 ) == 0 )
        {
            if ( length != 13 ) return Refuse( error, "bad IHDR" );
            hd.width      = Big32( data );
            hd.height     = Big32( data + 4 );
            hd.depth      = data[8];
            hd.colourType = data[9];
            hd.interlace  = data[12];
            if ( hd.width == 0 || hd.height == 0 || hd.width > ( 1u << 24 ) || hd.height > ( 1u << 24 ) ) return Refuse( error, "bad size" );
            if ( data[10] != 0 || data[11] != 0 || hd.interlace > 1 ) return Refuse( error, "unknown compression, filter or interlace method" );
            static const int channelsOf[7] = { 1, 0, 3, 1, 2, 0, 4 };
            if ( hd.colourType > 6 || channelsOf[hd.colourType] == 0 ) return Refuse( error, "bad colour type" );
            hd.channels      = channelsOf[hd.colourType];
            const bool depthOk = hd.depth == 8 || ( hd.depth == 16 && hd.colourType != 3 ) ||
                                 ( ( hd.depth == 1 || hd.depth == 2 || hd.depth == 4 ) && ( hd.colourType == 0 || hd.colourType == 3 ) );
            if ( !depthOk ) return Refuse( error, "bad bit depth" );
        }
        else if ( std::memcmp( type, "PLTE", 4 ) == 0 )
        {
            if ( length > 768 || length % 3 ) return Refuse( error, "bad PLTE" );
            paletteEntries = static_cast< int >( length / 3 );
            for ( int i = 0; i < paletteEntries; ++i ) std::memcpy( &palette[4 * i], data + 3 * i, 3 );
        }
        else if ( std::memcmp( type, "tRNS", 4 ) == 0 )
        {
            if ( !compressed.empty() ) return Refuse( error, "tRNS after IDAT" );
            if ( hd.colourType == 3 )
            {
                if ( static_cast< int >( length ) > paletteEntries ) return Refuse( error, "bad tRNS" );
                for ( uint32_t i = 0; i < length; ++i ) palette[4 * i + 3] = data[i];
            }
            else if ( hd.colourType == 0 || hd.colourType == 2 )
            {
                if ( length != static_cast< uint32_t >( 2 * hd.channels ) ) return Refuse( error, "bad tRNS" );
                for ( int k = 0; k < hd.channels; ++k ) key[k] = static_cast< uint16_t >( hd.depth == 16 ? data[2 * k] << 8 | data[2 * k + 1] : data[2 * k + 1] );
                haveKey = true;
            }
            else
                return Refuse( error, "tRNS with an alpha channel" );
        }
        else if ( std::memcmp( type, "IDAT", 4 ) == 0 )
        {
            if ( hd.width == 0 ) return Refuse( error, "IDAT before IHDR" );
            compressed.insert( compressed.end(), data, data + length );
        }
        else if ( std::memcmp( type, "IEND", 4 ) == 0 )
            break;
        else if ( !( type[0] & 32 ) )
            return Refuse( error, "unknown critical chunk" );
        pos += 12 + static_cast< size_t >( length );
    }
    if ( hd.width == 0 || compressed.empty() ) return Refuse( error, "no image data" );
    if ( hd.colourType == 3 && paletteEntries == 0 ) return Refuse( error, "palette image without PLTE" );
    if ( static_cast< uint64_t >( hd.width ) * hd.height > ( 1ull << 28 ) ) return Refuse( error, "image too large" );

    std::vector< uint8_t > raw;
    raw.reserve( ( static_cast< size_t >( hd.width ) * hd.channels * hd.depth / 8 + 1 ) * hd.height );
    Inflater inflater( compressed, raw );
    if ( !inflater.Run( error ) )
    {
        error = "PNG: " + error;
        return false;
    }

    out.width  = static_cast< int >( hd.width );
    out.height = static_cast< int >( hd.height );
    out.pixels.assign( static_cast< size_t >( hd.width ) * hd.height * 4, 255 );
    const int greyScale = hd.depth == 1 ? 255 : ( hd.depth == 2 ? 85 : ( hd.depth == 4 ? 17 : 1 ) );
    auto narrow = [&]( uint16_t v ) { return static_cast< uint8_t >( hd.depth == 16 ? v >> 8 : v ); };
    bool badIndex = false;
    auto store = [&]( uint32_t x, uint32_t y, const uint16_t* s ) {
        uint8_t* o = &out.pixels[( static_cast< size_t >( y ) * hd.width + x ) * 4];
        switch ( hd.colourType )
        {
            case 0:
                o[0] = o[1] = o[2] = static_cast< uint8_t >( hd.depth < 8 ? s[0] * greyScale : narrow( s[0] ) );
                if ( haveKey && s[0] == key[0] ) o[3] = 0;
                break;
            case 2:
                o[0] = narrow( s[0] );
                o[1] = narrow( s[1] );
                o[2] = narrow( s[2] );
                if ( haveKey && s[0] == key[0] && s[1] == key[1] && s[2] == key[2] ) o[3] = 0;
                break;
            case 3:
                if ( s[0] >= paletteEntries ) badIndex = true;
                std::memcpy( o, &palette[4 * ( s[0] & 255 )], 4 );
                break;
            case 4:
                o[0] = o[1] = o[2] = narrow( s[0] );
                o[3]               = narrow( s[1] );
                break;
            default:
                for ( int k = 0; k < 4; ++k ) o[k] = narrow( s[k] );
                break;
        }
    };

    size_t used = 0;
    if ( !hd.interlace )
    {
        if ( !Unfilter( hd, raw.data(), raw.size(), hd.width, hd.height, store, &used ) ) return Refuse( error, "not enough pixel data" );
    }
    else
    {
        static const int x0[7] = { 0, 4, 0, 2, 0, 1, 0 }, y0[7] = { 0, 0, 4, 0, 2, 0, 1 }, dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
        size_t offset = 0;
        for ( int pass = 0; pass < 7; ++pass )
        {
            const uint32_t w = ( hd.width - x0[pass] + dx[pass] - 1 ) / dx[pass], h = ( hd.height - y0[pass] + dy[pass] - 1 ) / dy[pass];
            if ( hd.width <= static_cast< uint32_t >( x0[pass] ) || hd.height <= static_cast< uint32_t >( y0[pass] ) || w == 0 || h == 0 ) continue;
            auto place = [&]( uint32_t x, uint32_t y, const uint16_t* s ) { store( x0[pass] + x * dx[pass], y0[pass] + y * dy[pass], s ); };
            if ( !Unfilter( hd, raw.data() + offset, raw.size() - offset, w, h, place, &used ) ) return Refuse( error, "not enough pixel data" );
            offset += used;
        }
    }
    if ( badIndex ) return Refuse( error, "palette index out of range" );
    return true;
}

} // namespace pth
