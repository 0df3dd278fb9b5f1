// Baseline JPEG writer for rendered pictures named *.jpg / *.jpeg, producing the file the reference produces
// (Image::Save, src/image.cpp:43-100, calls stb_image_write 1.14 with quality 95): 4:4:4, the standard's example
// quantisation tables (T.81 K.1, K.2) scaled by 10 %, its example Huffman tables (K.3-K.6), the single precision
// AAN forward DCT of the IJG library with the scale factors folded into the quantiser, JFIF colour conversion in
// single precision, values rounded half away from zero, blocks past the edge filled with the last row / column.
// Every float operation below is in the order that writer performs it; the host library is built without
// floating point contraction, so the coefficients - and with them every byte - come out the same.
#include "image_io.hpp"

#include <cstring>

namespace pth
{

namespace
{

    // natural (row-major) index -> position in the zig-zag scan
    const uint8_t kScanPosition[64] = { 0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                        41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                        46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63 };

    // T.81 tables K.1 and K.2, row by row
    const int kLumaQuant[64]   = { 16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
    const int kChromaQuant[64] = { 17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };

    // T.81 tables K.3 - K.6: number of codes of each length 1..16, then the symbols in code order
    const uint8_t kDcLumaCounts[16]   = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
    const uint8_t kDcChromaCounts[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
    const uint8_t kDcSymbols[12]      = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
    const uint8_t kAcLumaCounts[16]   = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
    const uint8_t kAcLumaSymbols[162] = {
        0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
        0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
        0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
        0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
        0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
        0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
    const uint8_t kAcChromaCounts[16]   = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
    const uint8_t kAcChromaSymbols[162] = {
        0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
        0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
        0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
        0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
        0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
        0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };

    struct Code
    {
        uint16_t bits   = 0;
        uint16_t length = 0;
    };

    // canonical codes of a table given as counts per length and symbols in code order (T.81 annex C)
    void BuildCodes( const uint8_t counts[16], const uint8_t* symbols, Code codes[256] )
    {
        int code = 0, k = 0;
        for ( int length = 1; length <= 16; ++length )
        {
            for ( int i = 0; i < counts[length - 1]; ++i ) codes[symbols[k++]] = Code{ static_cast< uint16_t >( code++ ), static_cast< uint16_t >( length ) };
            code <<= 1;
        }
    }

    class BitWriter
    {
    public:
        explicit BitWriter( std::vector< uint8_t >& out ) : m_out( out ) {}
        void Put( Code c )
        {
            m_count += c.length;
            m_bits |= static_cast< uint32_t >( c.bits ) << ( 24 - m_count );
            while ( m_count >= 8 )
            {
                const uint8_t byte = ( m_bits >> 16 ) & 255;
                m_out.push_back( byte );
                if ( byte == 255 ) m_out.push_back( 0 );
                m_bits <<= 8;
                m_count -= 8;
            }
        }

    private:
        std::vector< uint8_t >& m_out;
        uint32_t m_bits = 0;
        int m_count     = 0;
    };

    // the value's size category and the bits that follow the category's code (T.81 F.1.2.1)
    Code Magnitude( int value )
    {
        int magnitude = value < 0 ? -value : value, size = 1;
        while ( magnitude >>= 1 ) ++size;
        const int low = value < 0 ? value - 1 : value;
        return Code{ static_cast< uint16_t >( low & ( ( 1 << size ) - 1 ) ), static_cast< uint16_t >( size ) };
    }

    // Arai-Agui-Nakajima forward DCT on eight values `stride` apart, in place; outputs are scaled by the factors
    // that EncodeBlock's table divides out again
    void ForwardDct8( float* d, int stride )
    {
        float& d0 = d[0];
        float& d1 = d[stride];
        float& d2 = d[2 * stride];
        float& d3 = d[3 * stride];
        float& d4 = d[4 * stride];
        float& d5 = d[5 * stride];
        float& d6 = d[6 * stride];
        float& d7 = d[7 * stride];
        const float s07 = d0 + d7, m07 = d0 - d7, s16 = d1 + d6, m16 = d1 - d6, s25 = d2 + d5, m25 = d2 - d5, s34 = d3 + d4, m34 = d3 - d4;

        // even part
        const float e0 = s07 + s34, e3 = s07 - s34, e1 = s16 + s25, e2 = s16 - s25;
        const float z1 = ( e2 + e3 ) * 0.707106781f;
        // odd part, with the rotation arranged to avoid negations
        const float o0 = m34 + m25, o1 = m25 + m16, o2 = m16 + m07;
        const float z5 = ( o0 - o2 ) * 0.382683433f;
        const float z2 = o0 * 0.541196100f + z5;
        const float z4 = o2 * 1.306562965f + z5;
        const float z3 = o1 * 0.707106781f;
        const float z11 = m07 + z3, z13 = m07 - z3;

        d5 = z13 + z2;
        d3 = z13 - z2;
        d1 = z11 + z4;
        d7 = z11 - z4;
        d0 = e0 + e1;
        d4 = e0 - e1;
        d2 = e3 + z1;
        d6 = e3 - z1;
    }

    // transforms, quantises and entropy-codes one 8 x 8 block; returns its DC value for the next block's prediction
    int EncodeBlock( BitWriter& bits, float block[64], const float scale[64], int previousDc, const Code dc[256], const Code ac[256] )
    {
        for ( int r = 0; r < 8; ++r ) ForwardDct8( block + 8 * r, 1 );
        for ( int c = 0; c < 8; ++c ) ForwardDct8( block + c, 8 );
        int scan[64];
        for ( int k = 0; k < 64; ++k )
        {
            const float v           = block[k] * scale[k];
            scan[kScanPosition[k]] = static_cast< int >( v < 0 ? v - 0.5f : v + 0.5f );
        }

        const int difference = scan[0] - previousDc;
        if ( difference == 0 )
            bits.Put( dc[0] );
        else
        {
            const Code m = Magnitude( difference );
            bits.Put( dc[m.length] );
            bits.Put( m );
        }
        int last = 63;
        while ( last > 0 && scan[last] == 0 ) --last;
        if ( last == 0 )
        {
            bits.Put( ac[0x00] ); // end of block
            return scan[0];
        }
        for ( int i = 1; i <= last; ++i )
        {
            const int start = i;
            while ( scan[i] == 0 && i <= last ) ++i;
            int zeros = i - start;
            for ( int k = 0; k < zeros / 16; ++k ) bits.Put( ac[0xF0] ); // sixteen zeros
            zeros &= 15;
            const Code m = Magnitude( scan[i] );
            bits.Put( ac[( zeros << 4 ) + m.length] );
            bits.Put( m );
        }
        if ( last != 63 ) bits.Put( ac[0x00] );
        return scan[0];
    }

} // namespace

std::vector< uint8_t > EncodeJpegRGB8( const uint8_t* rgb, int width, int height )
{
    const int quality = 200 - 95 * 2; // the scale the reference's quality of 95 turns into, in percent
    uint8_t lumaTable[64], chromaTable[64]; // in scan order, as the file stores them
    for ( int i = 0; i < 64; ++i )
    {
        const int y = ( kLumaQuant[i] * quality + 50 ) / 100, c = ( kChromaQuant[i] * quality + 50 ) / 100;
        lumaTable[kScanPosition[i]]   = static_cast< uint8_t >( y < 1 ? 1 : ( y > 255 ? 255 : y ) );
        chromaTable[kScanPosition[i]] = static_cast< uint8_t >( c < 1 ? 1 : ( c > 255 ? 255 : c ) );
    }
    static const float aan[8] = { 1.0f * 2.828427125f, 1.387039845f * 2.828427125f, 1.306562965f * 2.828427125f, 1.175875602f * 2.828427125f,
                                  1.0f * 2.828427125f, 0.785694958f * 2.828427125f, 0.541196100f * 2.828427125f, 0.275899379f * 2.828427125f };
    float lumaScale[64], chromaScale[64];
    for ( int row = 0, k = 0; row < 8; ++row )
        for ( int col = 0; col < 8; ++col, ++k )
        {
            lumaScale[k]   = 1 / ( lumaTable[kScanPosition[k]] * aan[row] * aan[col] );
            chromaScale[k] = 1 / ( chromaTable[kScanPosition[k]] * aan[row] * aan[col] );
        }

    static Code dcLuma[256], acLuma[256], dcChroma[256], acChroma[256];
    BuildCodes( kDcLumaCounts, kDcSymbols, dcLuma );
    BuildCodes( kAcLumaCounts, kAcLumaSymbols, acLuma );
    BuildCodes( kDcChromaCounts, kDcSymbols, dcChroma );
    BuildCodes( kAcChromaCounts, kAcChromaSymbols, acChroma );

    std::vector< uint8_t > file;
    auto put = [&]( std::initializer_list< int > bytes ) { for ( int b : bytes ) file.push_back( static_cast< uint8_t >( b ) ); };
    auto putArray = [&]( const uint8_t* p, size_t n ) { file.insert( file.end(), p, p + n ); };
    put( { 0xFF, 0xD8 } );                                                                  // start of image
    put( { 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 } );         // JFIF 1.1, aspect ratio 1:1
    put( { 0xFF, 0xDB, 0, 0x84, 0 } );                                                      // both quantisation tables
    putArray( lumaTable, 64 );
    put( { 1 } );
    putArray( chromaTable, 64 );
    put( { 0xFF, 0xC0, 0, 17, 8, height >> 8, height & 255, width >> 8, width & 255, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1 } ); // baseline frame, 4:4:4
    put( { 0xFF, 0xC4, 0x01, 0xA2, 0x00 } );                                                // the four Huffman tables
    putArray( kDcLumaCounts, 16 );
    putArray( kDcSymbols, 12 );
    put( { 0x10 } );
    putArray( kAcLumaCounts, 16 );
    putArray( kAcLumaSymbols, 162 );
    put( { 0x01 } );
    putArray( kDcChromaCounts, 16 );
    putArray( kDcSymbols, 12 );
    put( { 0x11 } );
    putArray( kAcChromaCounts, 16 );
    putArray( kAcChromaSymbols, 162 );
    put( { 0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 0x3F, 0 } );                 // scan of all three components

    BitWriter bits( file );
    int dcY = 0, dcU = 0, dcV = 0;
    for ( int y = 0; y < height; y += 8 )
        for ( int x = 0; x < width; x += 8 )
        {
            float Y[64], U[64], V[64];
            for ( int row = y, at = 0; row < y + 8; ++row )
            {
                const uint8_t* line = rgb + 3 * static_cast< size_t >( row < height ? row : height - 1 ) * width;
                for ( int col = x; col < x + 8; ++col, ++at )
                {
                    const uint8_t* p = line + 3 * ( col < width ? col : width - 1 );
                    const float r = p[0], g = p[1], b = p[2];
                    Y[at] = +0.29900f * r + 0.58700f * g + 0.11400f * b - 128;
                    U[at] = -0.16874f * r - 0.33126f * g + 0.50000f * b;
                    V[at] = +0.50000f * r - 0.41869f * g - 0.08131f * b;
                }
            }
            dcY = EncodeBlock( bits, Y, lumaScale, dcY, dcLuma, acLuma );
            dcU = EncodeBlock( bits, U, chromaScale, dcU, dcChroma, acChroma );
            dcV = EncodeBlock( bits, V, chromaScale, dcV, dcChroma, acChroma );
        }
    bits.Put( Code{ 0x7F, 7 } ); // pad the last byte with ones
    put( { 0xFF, 0xD9 } );
    return file;
}

} // namespace pth
