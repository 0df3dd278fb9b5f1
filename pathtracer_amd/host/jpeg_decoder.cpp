// Baseline and progressive JPEG (Huffman-coded DCT, 8 bit; grey, YCbCr / RGB, Adobe CMYK / YCCK) to RGBA8, for the textures and cube maps the reference's scenes
// name (resources/images/*.jpg).  The reference decodes them with stb_image 2.25 (Texture::Load,
// src/resource/texture.cpp:19-33; Skybox::Load, src/resource/skybox.cpp:30-58), and a texel only equals the
// reference's if three choices the JPEG standard leaves open are made the same way:
//   * the inverse DCT: the 13-bit fixed point "slow integer" transform of the IJG library with two extra bits kept
//     between the column and the row pass (10 and 17 bit descaling shifts), coefficients dequantised in 16 bit;
//   * chroma upsampling: "triangle" interpolation centred like JFIF, 3/4 near + 1/4 far in each direction, the
//     vertical pass first (3 * near + far) and the horizontal pass on those sums ((3 a + b + 8) >> 4), the two
//     end samples from the vertical sums alone;
//   * YCbCr to RGB in 20-bit fixed point whose factors are first rounded to 12 bits, with the Cb term of green
//     truncated to its upper 16 bits before the sum.
// A progressive file's coefficients are collected over its scans (spectral selection, successive approximation) and
// dequantised with the tables in force at the end of the file, in 16-bit arithmetic, before the same inverse DCT.
// The tests compare with the reference's own decode of every JPEG file it ships and of generated files.
#include "image_io.hpp"

#include <cstring>
#include <fstream>
#include <iterator>

namespace pth
{

namespace
{

    // position k of the zig-zag scan -> index in the 8x8 block, row-major
    const uint8_t kZigZag[64] = { 0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

    struct HuffmanTable
    {
        bool present = false;
        int firstCode[17] = {};  // code of the first symbol of each length
        int firstIndex[17] = {}; // its position in `symbols`
        int count[17]      = {};
        uint8_t symbols[256] = {};
    };

    struct Component
    {
        int id = 0, h = 1, v = 1, quant = 0, dcTable = 0, acTable = 0;
        int width = 0, height = 0;   // samples that carry picture
        int stride = 0, rows = 0;    // samples allocated: whole MCUs
        int dcPrediction = 0;
        std::vector< uint8_t > plane;
        std::vector< int16_t > coefficients; // progressive files: 64 per block, stride / 8 blocks per row
    };

    uint8_t Clamp8( long long x ) { return x < 0 ? 0 : ( x > 255 ? 255 : static_cast< uint8_t >( x ) ); }

    // 13-bit constants of the slow integer DCT
    constexpr long long Fixed( double x ) { return static_cast< int >( x * 4096 + 0.5 ); }

    // One dimension of the inverse transform on s[0..7]; leaves the even part in x[0..3] and the odd part in t[0..3],
    // both scaled by 4096, for the caller to combine, bias and shift.  32 bits hold every intermediate of a valid
    // file; 64 are used so that the coefficients of a damaged one cannot overflow.
    inline void InverseDct1D( const int s[8], long long x[4], long long t[4] )
    {
        long long p2 = s[2], p3 = s[6];
        long long p1 = ( p2 + p3 ) * Fixed( 0.5411961f );
        const long long t2e = p1 + p3 * Fixed( -1.847759065f );
        const long long t3e = p1 + p2 * Fixed( 0.765366865f );
        const long long t0e = ( static_cast< long long >( s[0] ) + s[4] ) * 4096;
        const long long t1e = ( static_cast< long long >( s[0] ) - s[4] ) * 4096;
        x[0] = t0e + t3e;
        x[3] = t0e - t3e;
        x[1] = t1e + t2e;
        x[2] = t1e - t2e;
        long long t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
        p3 = t0 + t2;
        long long p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const long long p5 = ( p3 + p4 ) * Fixed( 1.175875602f );
        t0 = t0 * Fixed( 0.298631336f );
        t1 = t1 * Fixed( 2.053119869f );
        t2 = t2 * Fixed( 3.072711026f );
        t3 = t3 * Fixed( 1.501321110f );
        p1 = p5 + p1 * Fixed( -0.899976223f );
        p2 = p5 + p2 * Fixed( -2.562915447f );
        p3 = p3 * Fixed( -1.961570560f );
        p4 = p4 * Fixed( -0.390180644f );
        t[3] = t3 + p1 + p4;
        t[2] = t2 + p2 + p3;
        t[1] = t1 + p2 + p4;
        t[0] = t0 + p1 + p3;
    }

    int Narrow( long long x ) { return x < -0x40000000ll ? -0x40000000 : ( x > 0x3fffffffll ? 0x3fffffff : static_cast< int >( x ) ); }

    void InverseDctBlock( const int16_t coefficients[64], uint8_t* out, int stride )
    {
        int mid[64];
        for ( int c = 0; c < 8; ++c ) // columns, two extra bits of precision kept
        {
            int s[8];
            long long x[4], t[4];
            bool acZero = true;
            for ( int r = 0; r < 8; ++r )
            {
                s[r] = coefficients[8 * r + c];
                if ( r && s[r] ) acZero = false;
            }
            if ( acZero )
            {
                for ( int r = 0; r < 8; ++r ) mid[8 * r + c] = s[0] * 4;
                continue;
            }
            InverseDct1D( s, x, t );
            for ( int k = 0; k < 4; ++k )
            {
                mid[8 * k + c]         = Narrow( ( x[k] + 512 + t[3 - k] ) >> 10 );
                mid[8 * ( 7 - k ) + c] = Narrow( ( x[k] + 512 - t[3 - k] ) >> 10 );
            }
        }
        for ( int r = 0; r < 8; ++r ) // rows: 12 bits of the constants, 2 kept above, 3 of the two sqrt(8) scalings
        {
            long long x[4], t[4];
            InverseDct1D( mid + 8 * r, x, t );
            uint8_t* o           = out + static_cast< size_t >( r ) * stride;
            const long long bias = 65536 + ( 128 << 17 ); // rounding, and the level shift back to 0..255
            for ( int k = 0; k < 4; ++k )
            {
                o[k]     = Clamp8( ( x[k] + bias + t[3 - k] ) >> 17 );
                o[7 - k] = Clamp8( ( x[k] + bias - t[3 - k] ) >> 17 );
            }
        }
    }

    class Decoder
    {
    public:
        Decoder( const std::vector< uint8_t >& bytes ) : m_data( bytes ) {}

        bool Decode( ImageRGBA8& out, std::string& error )
        {
            if ( !ParseHeaders( error ) ) return false;
            if ( m_progressive ) TransformCollectedBlocks();
            Output( out );
            return true;
        }

    private:
        const std::vector< uint8_t >& m_data;
        size_t m_pos = 0;
        int m_width = 0, m_height = 0, m_hMax = 1, m_vMax = 1, m_mcusX = 0, m_mcusY = 0, m_restartInterval = 0;
        bool m_jfif = false, m_progressive = false;
        int m_spectralStart = 0, m_spectralEnd = 63, m_approxHigh = 0, m_approxLow = 0, m_endOfBandRun = 0;
        int m_adobeTransform = -1;
        std::vector< Component > m_components;
        uint16_t m_quant[4][64] = {};
        HuffmanTable m_dc[4], m_ac[4];

        // entropy-coded segment reader
        uint32_t m_bits = 0;
        int m_bitCount  = 0;
        bool m_hitMarker = false;

        bool Fail( std::string& error, const char* what )
        {
            error = std::string( "JPEG: " ) + what;
            return false;
        }

        int Byte() { return m_pos < m_data.size() ? m_data[m_pos++] : 0; }
        int Word()
        {
            int hi = Byte();
            return hi << 8 | Byte();
        }

        bool ParseHeaders( std::string& error )
        {
            if ( m_data.size() < 4 || m_data[0] != 0xFF || m_data[1] != 0xD8 ) return Fail( error, "no SOI marker" );
            m_pos = 2;
            bool haveFrame = false;
            while ( m_pos + 4 <= m_data.size() )
            {
                if ( Byte() != 0xFF ) return Fail( error, "marker expected" );
                int marker = Byte();
                while ( marker == 0xFF ) marker = Byte(); // fill bytes
                if ( marker == 0xD9 ) break;
                if ( marker == 0x01 || ( marker >= 0xD0 && marker <= 0xD7 ) ) continue; // no payload
                const size_t start = m_pos;
                const int length   = Word();
                if ( length < 2 || start + length > m_data.size() ) return Fail( error, "segment runs past the end of the file" );
                const size_t end = start + length;
                switch ( marker )
                {
                    case 0xE0: // APP0
                        if ( length >= 7 && std::memcmp( &m_data[m_pos], "JFIF\0", 5 ) == 0 ) m_jfif = true;
                        break;
                    case 0xEE: // APP14
                        if ( length >= 14 && std::memcmp( &m_data[m_pos], "Adobe\0", 6 ) == 0 ) m_adobeTransform = m_data[m_pos + 11];
                        break;
                    case 0xDB: // DQT
                        while ( m_pos < end )
                        {
                            const int info = Byte(), table = info & 15;
                            const bool wide = ( info >> 4 ) != 0;
                            if ( table > 3 ) return Fail( error, "bad quantisation table index" );
                            for ( int k = 0; k < 64; ++k ) m_quant[table][kZigZag[k]] = static_cast< uint16_t >( wide ? Word() : Byte() );
                        }
                        break;
                    case 0xC4: // DHT
                        while ( m_pos < end )
                        {
                            const int info = Byte(), index = info & 15;
                            if ( index > 3 || ( info >> 4 ) > 1 ) return Fail( error, "bad Huffman table header" );
                            HuffmanTable& h = ( info >> 4 ) ? m_ac[index] : m_dc[index];
                            int total = 0, code = 0;
                            for ( int len = 1; len <= 16; ++len )
                            {
                                h.count[len]      = Byte();
                                h.firstIndex[len] = total;
                                h.firstCode[len]  = code;
                                total += h.count[len];
                                code = ( code + h.count[len] ) << 1;
                            }
                            if ( total > 256 ) return Fail( error, "bad Huffman table" );
                            for ( int k = 0; k < total; ++k ) h.symbols[k] = static_cast< uint8_t >( Byte() );
                            h.present = true;
                        }
                        break;
                    case 0xDD: // DRI
                        m_restartInterval = Word();
                        break;
                    case 0xC0: // SOF0, SOF1: sequential, Huffman
                    case 0xC1:
                        m_progressive = false;
                        if ( !ParseFrame( error ) ) return false;
                        haveFrame = true;
                        break;
                    case 0xC2: // SOF2: progressive, Huffman
                        m_progressive = true;
                        if ( !ParseFrame( error ) ) return false;
                        haveFrame = true;
                        break;
                    case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                        return Fail( error, "only Huffman-coded sequential and progressive DCT files are decoded" );
                    case 0xDA: // SOS
                        if ( !haveFrame ) return Fail( error, "scan before frame header" );
                        if ( !ParseScan( error ) ) return false;
                        continue; // the scan leaves m_pos at the next marker
                    default:
                        break;
                }
                m_pos = end;
            }
            if ( !haveFrame ) return Fail( error, "no frame header" );
            return true;
        }

        bool ParseFrame( std::string& error )
        {
            if ( Byte() != 8 ) return Fail( error, "only 8-bit samples" );
            m_height = Word();
            m_width  = Word();
            const int n = Byte();
            if ( m_width <= 0 || m_height <= 0 ) return Fail( error, "empty picture" );
            if ( n != 1 && n != 3 && n != 4 ) return Fail( error, "1, 3 or 4 components expected" );
            m_components.assign( n, Component() );
            for ( Component& c : m_components )
            {
                c.id        = Byte();
                const int s = Byte();
                c.h = s >> 4;
                c.v = s & 15;
                c.quant = Byte();
                if ( c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.quant > 3 ) return Fail( error, "bad component" );
                m_hMax = c.h > m_hMax ? c.h : m_hMax;
                m_vMax = c.v > m_vMax ? c.v : m_vMax;
            }
            for ( const Component& c : m_components )
                if ( m_hMax % c.h || m_vMax % c.v ) return Fail( error, "fractional sampling ratios are not decoded" );
            m_mcusX = ( m_width + 8 * m_hMax - 1 ) / ( 8 * m_hMax );
            m_mcusY = ( m_height + 8 * m_vMax - 1 ) / ( 8 * m_vMax );
            for ( Component& c : m_components )
            {
                c.width  = ( m_width * c.h + m_hMax - 1 ) / m_hMax;
                c.height = ( m_height * c.v + m_vMax - 1 ) / m_vMax;
                c.stride = m_mcusX * c.h * 8;
                c.rows   = m_mcusY * c.v * 8;
                c.plane.assign( static_cast< size_t >( c.stride ) * c.rows, 0 );
                if ( m_progressive ) c.coefficients.assign( static_cast< size_t >( c.stride ) * c.rows, 0 );
            }
            return true;
        }

        // ---- entropy-coded data

        void FillBits()
        {
            while ( m_bitCount <= 24 )
            {
                int b = 0;
                if ( !m_hitMarker && m_pos < m_data.size() )
                {
                    b = m_data[m_pos];
                    if ( b == 0xFF )
                    {
                        const int next = m_pos + 1 < m_data.size() ? m_data[m_pos + 1] : 0xD9;
                        if ( next == 0 )
                            m_pos += 2; // stuffed zero
                        else
                        {
                            m_hitMarker = true; // leave the marker where it is; zeros from here on
                            b           = 0;
                        }
                    }
                    else
                        ++m_pos;
                }
                m_bits |= static_cast< uint32_t >( b ) << ( 24 - m_bitCount );
                m_bitCount += 8;
            }
        }

        int Bits( int n )
        {
            if ( n == 0 ) return 0;
            if ( m_bitCount < n ) FillBits();
            const int v = static_cast< int >( m_bits >> ( 32 - n ) );
            m_bits <<= n;
            m_bitCount -= n;
            return v;
        }

        int Symbol( const HuffmanTable& h )
        {
            int code = 0;
            for ( int len = 1; len <= 16; ++len )
            {
                code = code << 1 | Bits( 1 );
                const int offset = code - h.firstCode[len];
                if ( offset >= 0 && offset < h.count[len] ) return h.symbols[h.firstIndex[len] + offset];
            }
            return -1;
        }

        // the `size`-bit two's complement style difference of the standard (F.2.2.1)
        int Extend( int size )
        {
            if ( size == 0 ) return 0;
            const int v = Bits( size );
            return v < ( 1 << ( size - 1 ) ) ? v - ( 1 << size ) + 1 : v;
        }

        bool DecodeBlock( Component& c, int16_t block[64] )
        {
            std::memset( block, 0, 64 * sizeof( int16_t ) );
            const uint16_t* q = m_quant[c.quant];
            int size = Symbol( m_dc[c.dcTable] );
            if ( size < 0 || size > 15 ) return false;
            c.dcPrediction = WrapAdd( c.dcPrediction, Extend( size ) );
            block[0]       = Int16( WrapMul( c.dcPrediction, q[0] ) );
            for ( int k = 1; k < 64; )
            {
                const int rs = Symbol( m_ac[c.acTable] );
                if ( rs < 0 ) return false;
                const int run = rs >> 4, bits = rs & 15;
                if ( bits == 0 )
                {
                    if ( run != 15 ) break; // end of block
                    k += 16;
                    continue;
                }
                k += run;
                if ( k > 63 ) return false;
                const int at = kZigZag[k++];
                block[at]    = static_cast< int16_t >( Extend( bits ) * q[at] );
            }
            return true;
        }

        // arithmetic that wraps like the 32- and 16-bit registers it models, whatever a damaged file feeds it
        static int WrapAdd( int a, int b ) { return static_cast< int >( static_cast< uint32_t >( a ) + static_cast< uint32_t >( b ) ); }
        static int WrapMul( int a, int b ) { return static_cast< int >( static_cast< uint32_t >( a ) * static_cast< uint32_t >( b ) ); }
        static int16_t Int16( int a ) { return static_cast< int16_t >( static_cast< uint16_t >( static_cast< uint32_t >( a ) ) ); }

        // DC scan of a progressive file: the first one carries the prediction differences scaled down by 2^low,
        // a refining one a single further bit per block
        bool DecodeProgressiveDc( Component& c, int16_t block[64] )
        {
            if ( m_approxHigh == 0 )
            {
                std::memset( block, 0, 64 * sizeof( int16_t ) );
                const int size = Symbol( m_dc[c.dcTable] );
                if ( size < 0 || size > 15 ) return false;
                c.dcPrediction = WrapAdd( c.dcPrediction, Extend( size ) );
                block[0]       = Int16( WrapMul( c.dcPrediction, 1 << m_approxLow ) );
            }
            else if ( Bits( 1 ) )
                block[0] = Int16( block[0] + ( 1 << m_approxLow ) );
            return true;
        }

        // one more bit of a coefficient that is already non-zero
        void Refine( int16_t& value, int bit )
        {
            if ( Bits( 1 ) && ( value & bit ) == 0 ) value = Int16( value > 0 ? value + bit : value - bit );
        }

        // AC scan of a progressive file over the band [m_spectralStart, m_spectralEnd] of one block
        bool DecodeProgressiveAc( const Component& c, int16_t block[64] )
        {
            const HuffmanTable& table = m_ac[c.acTable];
            if ( m_approxHigh == 0 )
            {
                if ( m_endOfBandRun )
                {
                    --m_endOfBandRun;
                    return true;
                }
                int k = m_spectralStart;
                do
                {
                    const int rs = Symbol( table );
                    if ( rs < 0 ) return false;
                    const int run = rs >> 4, bits = rs & 15;
                    if ( bits == 0 )
                    {
                        if ( run < 15 ) // this band ends here, in this block and the next `run length - 1`
                        {
                            m_endOfBandRun = ( 1 << run ) + ( run ? Bits( run ) : 0 ) - 1;
                            break;
                        }
                        k += 16;
                    }
                    else
                    {
                        k += run;
                        const int at = k < 64 ? kZigZag[k] : 63;
                        ++k;
                        block[at] = Int16( WrapMul( Extend( bits ), 1 << m_approxLow ) );
                    }
                } while ( k <= m_spectralEnd );
                return true;
            }

            const int bit = 1 << m_approxLow;
            if ( m_endOfBandRun )
            {
                --m_endOfBandRun;
                for ( int k = m_spectralStart; k <= m_spectralEnd; ++k )
                    if ( block[kZigZag[k]] != 0 ) Refine( block[kZigZag[k]], bit );
                return true;
            }
            int k = m_spectralStart;
            do
            {
                const int rs = Symbol( table );
                if ( rs < 0 ) return false;
                int run = rs >> 4, value = 0;
                if ( ( rs & 15 ) == 0 )
                {
                    if ( run < 15 )
                    {
                        m_endOfBandRun = ( 1 << run ) - 1 + ( run ? Bits( run ) : 0 );
                        run            = 64; // no new coefficient in the rest of the band, only refinements
                    }
                    // run 15: sixteen zeros, the sixteenth "set" to zero below
                }
                else
                {
                    if ( ( rs & 15 ) != 1 ) return false;
                    value = Bits( 1 ) ? bit : -bit;
                }
                while ( k <= m_spectralEnd )
                {
                    int16_t& coefficient = block[kZigZag[k++]];
                    if ( coefficient != 0 )
                        Refine( coefficient, bit );
                    else if ( run-- == 0 )
                    {
                        coefficient = static_cast< int16_t >( value );
                        break;
                    }
                }
            } while ( k <= m_spectralEnd );
            return true;
        }

        // progressive files: dequantise what the scans collected, with 16-bit products, and transform
        void TransformCollectedBlocks()
        {
            for ( Component& c : m_components )
            {
                const uint16_t* q = m_quant[c.quant];
                const int bx = ( c.width + 7 ) / 8, by = ( c.height + 7 ) / 8, perRow = c.stride / 8;
                for ( int y = 0; y < by; ++y )
                    for ( int x = 0; x < bx; ++x )
                    {
                        int16_t* block = &c.coefficients[64 * ( static_cast< size_t >( y ) * perRow + x )];
                        for ( int i = 0; i < 64; ++i ) block[i] = Int16( WrapMul( block[i], q[i] ) );
                        InverseDctBlock( block, &c.plane[static_cast< size_t >( 8 * y ) * c.stride + 8 * x], c.stride );
                    }
            }
        }

        void RestartDecoder()
        {
            m_bits = 0;
            m_bitCount = 0;
            m_hitMarker = false;
            m_endOfBandRun = 0;
            for ( Component& c : m_components ) c.dcPrediction = 0;
        }

        bool ParseScan( std::string& error )
        {
            const int n = Byte();
            if ( n < 1 || n > static_cast< int >( m_components.size() ) ) return Fail( error, "bad scan header" );
            std::vector< Component* > scan;
            for ( int i = 0; i < n; ++i )
            {
                const int id = Byte(), tables = Byte();
                Component* found = nullptr;
                for ( Component& c : m_components )
                    if ( c.id == id ) found = &c;
                if ( !found ) return Fail( error, "scan names an unknown component" );
                found->dcTable = tables >> 4;
                found->acTable = tables & 15;
                if ( found->dcTable > 3 || found->acTable > 3 ) return Fail( error, "bad Huffman table index" );
                scan.push_back( found );
            }
            m_spectralStart = Byte();
            m_spectralEnd   = Byte();
            const int approximation = Byte();
            m_approxHigh = approximation >> 4;
            m_approxLow  = approximation & 15;
            if ( m_progressive )
            {
                if ( m_spectralStart > 63 || m_spectralEnd > 63 || m_spectralStart > m_spectralEnd || m_approxHigh > 13 || m_approxLow > 13 )
                    return Fail( error, "bad scan header" );
                if ( m_spectralStart == 0 ? m_spectralEnd != 0 : n != 1 ) return Fail( error, "DC and AC coefficients in one progressive scan" );
            }
            for ( const Component* c : scan )
            {
                const bool needsDc = !m_progressive || ( m_spectralStart == 0 && m_approxHigh == 0 ), needsAc = !m_progressive || m_spectralStart != 0;
                if ( ( needsDc && !m_dc[c->dcTable].present ) || ( needsAc && !m_ac[c->acTable].present ) )
                    return Fail( error, "scan uses a Huffman table that was not defined" );
            }
            RestartDecoder();

            int16_t block[64];
            int untilRestart = m_restartInterval ? m_restartInterval : 0x7fffffff;
            auto afterUnit   = [&]() -> bool {
                if ( --untilRestart > 0 ) return true;
                // a restart marker is due: skip to it, start over
                if ( !m_hitMarker )
                    while ( m_pos + 1 < m_data.size() && !( m_data[m_pos] == 0xFF && m_data[m_pos + 1] != 0 ) ) ++m_pos;
                if ( m_pos + 1 < m_data.size() && m_data[m_pos + 1] >= 0xD0 && m_data[m_pos + 1] <= 0xD7 )
                {
                    m_pos += 2;
                    RestartDecoder();
                    untilRestart = m_restartInterval;
                    return true;
                }
                return false; // some other marker: the scan is over
            };

            if ( scan.size() == 1 )
            {
                // not interleaved: the component's own blocks, only those that carry picture
                Component& c  = *scan[0];
                const int bx = ( c.width + 7 ) / 8, by = ( c.height + 7 ) / 8;
                for ( int y = 0; y < by; ++y )
                    for ( int x = 0; x < bx; ++x )
                    {
                        if ( m_progressive )
                        {
                            int16_t* collected = &c.coefficients[64 * ( static_cast< size_t >( y ) * ( c.stride / 8 ) + x )];
                            if ( !( m_spectralStart == 0 ? DecodeProgressiveDc( c, collected ) : DecodeProgressiveAc( c, collected ) ) )
                                return Fail( error, "bad Huffman code" );
                        }
                        else
                        {
                            if ( !DecodeBlock( c, block ) ) return Fail( error, "bad Huffman code" );
                            InverseDctBlock( block, &c.plane[static_cast< size_t >( 8 * y ) * c.stride + 8 * x], c.stride );
                        }
                        if ( !afterUnit() ) goto done;
                    }
            }
            else
            {
                for ( int my = 0; my < m_mcusY; ++my )
                    for ( int mx = 0; mx < m_mcusX; ++mx )
                    {
                        for ( Component* c : scan )
                            for ( int v = 0; v < c->v; ++v )
                                for ( int h = 0; h < c->h; ++h )
                                {
                                    const int x = ( mx * c->h + h ) * 8, y = ( my * c->v + v ) * 8;
                                    if ( m_progressive ) // only DC scans interleave
                                    {
                                        if ( !DecodeProgressiveDc( *c, &c->coefficients[64 * ( static_cast< size_t >( y / 8 ) * ( c->stride / 8 ) + x / 8 )] ) )
                                            return Fail( error, "bad Huffman code" );
                                        continue;
                                    }
                                    if ( !DecodeBlock( *c, block ) ) return Fail( error, "bad Huffman code" );
                                    InverseDctBlock( block, &c->plane[static_cast< size_t >( y ) * c->stride + x], c->stride );
                                }
                        if ( !afterUnit() ) goto done;
                    }
            }
        done:
            // continue at the marker that ended the entropy-coded data
            if ( !m_hitMarker )
                while ( m_pos + 1 < m_data.size() && !( m_data[m_pos] == 0xFF && m_data[m_pos + 1] != 0 && !( m_data[m_pos + 1] >= 0xD0 && m_data[m_pos + 1] <= 0xD7 ) ) ) ++m_pos;
            return true;
        }

        // ---- upsampling and colour

        // one output row of a component: `near` is the input row this output row lies in, `far` its neighbour on
        // the side the output row is closer to; `width` input samples
        static const uint8_t* Upsample( const Component& c, int hs, int vs, const uint8_t* near, const uint8_t* far, int width, std::vector< uint8_t >& line )
        {
            if ( hs == 1 && vs == 1 ) return near;
            if ( hs == 1 && vs == 2 )
            {
                for ( int i = 0; i < width; ++i ) line[i] = static_cast< uint8_t >( ( 3 * near[i] + far[i] + 2 ) >> 2 );
            }
            else if ( hs == 2 && vs == 1 )
            {
                if ( width == 1 )
                    line[0] = line[1] = near[0];
                else
                {
                    line[0] = near[0];
                    line[1] = static_cast< uint8_t >( ( near[0] * 3 + near[1] + 2 ) >> 2 );
                    for ( int i = 1; i < width - 1; ++i )
                    {
                        const int n     = 3 * near[i] + 2;
                        line[2 * i]     = static_cast< uint8_t >( ( n + near[i - 1] ) >> 2 );
                        line[2 * i + 1] = static_cast< uint8_t >( ( n + near[i + 1] ) >> 2 );
                    }
                    line[2 * width - 2] = static_cast< uint8_t >( ( near[width - 2] * 3 + near[width - 1] + 2 ) >> 2 );
                    line[2 * width - 1] = near[width - 1];
                }
            }
            else if ( hs == 2 && vs == 2 )
            {
                int current = 3 * near[0] + far[0];
                if ( width == 1 )
                    line[0] = line[1] = static_cast< uint8_t >( ( current + 2 ) >> 2 );
                else
                {
                    line[0] = static_cast< uint8_t >( ( current + 2 ) >> 2 );
                    for ( int i = 1; i < width; ++i )
                    {
                        const int previous = current;
                        current            = 3 * near[i] + far[i];
                        line[2 * i - 1]    = static_cast< uint8_t >( ( 3 * previous + current + 8 ) >> 4 );
                        line[2 * i]        = static_cast< uint8_t >( ( 3 * current + previous + 8 ) >> 4 );
                    }
                    line[2 * width - 1] = static_cast< uint8_t >( ( current + 2 ) >> 2 );
                }
            }
            else // other ratios: nearest sample
            {
                for ( int i = 0; i < width; ++i )
                    for ( int k = 0; k < hs; ++k ) line[i * hs + k] = near[i];
            }
            (void)c;
            return line.data();
        }

        void Output( ImageRGBA8& out )
        {
            out.width  = m_width;
            out.height = m_height;
            out.pixels.assign( static_cast< size_t >( m_width ) * m_height * 4, 255 );
            const size_t n = m_components.size();
            // components named R, G, B, or an Adobe file that says "no transform" outside JFIF, hold RGB already
            int namedRgb = 0;
            for ( size_t k = 0; k < n && k < 3; ++k ) namedRgb += m_components[k].id == "RGB"[k];
            const bool isRgb = n == 3 && ( namedRgb == 3 || ( m_adobeTransform == 0 && !m_jfif ) );

            struct RowState
            {
                int hs, vs, step, row, inputWidth;
                const uint8_t *line0, *line1;
                std::vector< uint8_t > buffer;
            };
            std::vector< RowState > state( n );
            for ( size_t k = 0; k < n; ++k )
            {
                const Component& c = m_components[k];
                RowState& r = state[k];
                r.hs = m_hMax / c.h;
                r.vs = m_vMax / c.v;
                r.step = r.vs >> 1;
                r.row = 0;
                r.inputWidth = ( m_width + r.hs - 1 ) / r.hs;
                r.line0 = r.line1 = c.plane.data();
                r.buffer.assign( static_cast< size_t >( m_width ) + 3 * 4 + 8, 0 );
            }
            const uint8_t* rows[4] = { nullptr, nullptr, nullptr, nullptr };
            // a * b / 255, rounded, for bytes
            auto scale = []( int a, int b ) { const unsigned t = static_cast< unsigned >( a * b + 128 ); return static_cast< uint8_t >( ( t + ( t >> 8 ) ) >> 8 ); };
            for ( int y = 0; y < m_height; ++y )
            {
                for ( size_t k = 0; k < n; ++k )
                {
                    RowState& r   = state[k];
                    const bool lower = r.step >= ( r.vs >> 1 ); // the output row lies in the lower half of its input row
                    rows[k] = Upsample( m_components[k], r.hs, r.vs, lower ? r.line1 : r.line0, lower ? r.line0 : r.line1, r.inputWidth, r.buffer );
                    if ( ++r.step >= r.vs )
                    {
                        r.step  = 0;
                        r.line0 = r.line1;
                        if ( ++r.row < m_components[k].height ) r.line1 += m_components[k].stride;
                    }
                }
                uint8_t* o = &out.pixels[static_cast< size_t >( y ) * m_width * 4];
                if ( n == 1 )
                    for ( int x = 0; x < m_width; ++x, o += 4 ) o[0] = o[1] = o[2] = rows[0][x];
                else if ( isRgb )
                    for ( int x = 0; x < m_width; ++x, o += 4 )
                    {
                        o[0] = rows[0][x];
                        o[1] = rows[1][x];
                        o[2] = rows[2][x];
                    }
                else if ( n == 4 && m_adobeTransform == 0 ) // CMYK as Adobe writes it (inverted): ink times black
                    for ( int x = 0; x < m_width; ++x, o += 4 )
                    {
                        const int black = rows[3][x];
                        o[0] = scale( rows[0][x], black );
                        o[1] = scale( rows[1][x], black );
                        o[2] = scale( rows[2][x], black );
                    }
                else
                    for ( int x = 0; x < m_width; ++x, o += 4 )
                    {
                        // factors rounded to 12 bits, then carried in 20: 1.402, 0.71414, 0.34414, 1.772
                        const int luma = ( rows[0][x] << 20 ) + ( 1 << 19 );
                        const int cb = rows[1][x] - 128, cr = rows[2][x] - 128;
                        const int r = luma + cr * ( Fixed12( 1.40200f ) << 8 );
                        const int g = luma + cr * -( Fixed12( 0.71414f ) << 8 ) + static_cast< int >( static_cast< unsigned >( cb * -( Fixed12( 0.34414f ) << 8 ) ) & 0xffff0000u );
                        const int b = luma + cb * ( Fixed12( 1.77200f ) << 8 );
                        o[0] = Clamp8( r >> 20 );
                        o[1] = Clamp8( g >> 20 );
                        o[2] = Clamp8( b >> 20 );
                        if ( n == 4 && m_adobeTransform == 2 ) // YCCK: the colour is inverted, then scaled by black
                        {
                            const int black = rows[3][x];
                            o[0] = scale( 255 - o[0], black );
                            o[1] = scale( 255 - o[1], black );
                            o[2] = scale( 255 - o[2], black );
                        } // four components without that marker: the fourth is ignored
                    }
            }
        }

        static constexpr int Fixed12( float x ) { return static_cast< int >( x * 4096.0f + 0.5f ); }
    };

} // namespace

bool ReadJpegRGBA8( const std::string& filename, ImageRGBA8& out, std::string& error )
{
    std::ifstream f( filename, std::ios::binary );
    if ( !f )
    {
        error = "could not open file";
        return false;
    }
    const std::vector< uint8_t > bytes( ( std::istreambuf_iterator< char >( f ) ), std::istreambuf_iterator< char >() );
    Decoder decoder( bytes );
    return decoder.Decode( out, error );
}

} // namespace pth
