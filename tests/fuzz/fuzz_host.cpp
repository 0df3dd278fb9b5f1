// Damaged input files through the host library's readers, built with -fsanitize=address,undefined by
// tests/test_fuzz_host.py: whatever the bytes, a reader returns an image / a scene or an error message.
//   fuzz_host image FILE ITERATIONS          mutates FILE (TGA, BMP, PPM, JPEG, PNG, HDR) and decodes it
//   fuzz_host scene DIR ITERATIONS           DIR holds scenes/base.json and the model files it names, writable
#include "image_io.hpp"
#include "scene_loader.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

namespace
{

std::vector< char > Slurp( const std::string& name )
{
    std::ifstream f( name, std::ios::binary );
    return std::vector< char >( ( std::istreambuf_iterator< char >( f ) ), std::istreambuf_iterator< char >() );
}

void Spit( const std::string& name, const std::vector< char >& bytes )
{
    std::ofstream o( name, std::ios::binary );
    o.write( bytes.data(), static_cast< std::streamsize >( bytes.size() ) );
}

void Damage( std::vector< char >& bytes, std::mt19937& rng, size_t headerBytes, bool text )
{
    static const char printable[] = "{}[]\",:0123456789.-eE abcflmnrstuv\n/#";
    const int edits = 1 + static_cast< int >( rng() % 8 );
    for ( int e = 0; e < edits && !bytes.empty(); ++e )
    {
        const size_t span = ( rng() % 3 == 0 && headerBytes < bytes.size() ) ? headerBytes : bytes.size();
        bytes[rng() % span] = ( text && ( rng() & 1 ) ) ? printable[rng() % ( sizeof( printable ) - 1 )] : static_cast< char >( rng() );
    }
    if ( rng() % 11 == 0 && !bytes.empty() ) bytes.resize( rng() % bytes.size() );
}

} // namespace

int main( int argc, char** argv )
{
    if ( argc != 4 ) return 2;
    const std::string mode = argv[1], path = argv[2];
    const int iterations = std::atoi( argv[3] );
    std::mt19937 rng( 7 );
    int accepted = 0, refused = 0;
    if ( mode == "image" )
    {
        const std::vector< char > original = Slurp( path );
        const std::string copy = path.substr( 0, path.rfind( '.' ) ) + "_damaged" + path.substr( path.rfind( '.' ) );
        for ( int it = 0; it < iterations; ++it )
        {
            std::vector< char > bytes = original;
            Damage( bytes, rng, 64, false );
            Spit( copy, bytes );
            std::string error;
            if ( copy.size() > 4 && copy.substr( copy.size() - 4 ) == ".hdr" )
            {
                std::vector< float > pixels;
                int w = 0, h = 0;
                ( pth::ReadImageRGBA32F( copy, true, w, h, pixels, error ) ? accepted : refused )++;
                continue;
            }
            pth::ImageRGBA8 image;
            ( pth::ReadImageRGBA8( copy, true, image, error ) ? accepted : refused )++;
        }
    }
    else
    {
        const std::vector< char > json = Slurp( path + "/scenes/base.json" ), obj = Slurp( path + "/models/base.obj" ), mtl = Slurp( path + "/models/base.mtl" );
        for ( int it = 0; it < iterations; ++it )
        {
            std::vector< char > j = json, o = obj, m = mtl;
            Damage( it % 3 == 0 ? j : ( it % 3 == 1 ? o : m ), rng, 200, true );
            Spit( path + "/scenes/damaged.json", j );
            Spit( path + "/models/model.obj", o );
            Spit( path + "/models/model.mtl", m );
            pth::HostScene scene;
            pth::SceneLoadOptions options;
            options.resourceDir = path + "/";
            options.log         = []( const std::string& ) {};
            std::string error;
            try
            {
                ( pth::LoadSceneFile( scene, path + "/scenes/damaged.json", options, error ) ? accepted : refused )++;
            }
            catch ( const std::exception& )
            {
                ++refused; // the C API turns these into error codes
            }
        }
    }
    std::printf( "%d accepted, %d refused\n", accepted, refused );
    return 0;
}
