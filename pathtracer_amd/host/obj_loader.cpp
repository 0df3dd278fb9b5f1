#include "obj_loader.hpp"
#include "image_io.hpp"
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

namespace pth
{

namespace
{

    struct Corner
    {
        int v = -1, vt = -1, vn = -1;
    };

    // One group of faces that ends up as one mesh: Assimp's OBJ importer starts a new mesh whenever the
    // object / group or the active material changes.
    struct FaceGroup
    {
        std::string name;
        std::string material;
        std::vector< std::vector< Corner > > faces;
    };

    struct MtlEntry
    {
        std::string name;
        vec3 Kd = vec3( 0.6f ), Ks = vec3( 0 ), Ke = vec3( 0 ); // Assimp's OBJ default diffuse is 0.6
        float Ns = 0;
        std::string mapKd;
    };

    std::string Trim( const std::string& s )
    {
        size_t a = s.find_first_not_of( " \t\r\n" );
        if ( a == std::string::npos ) return "";
        size_t b = s.find_last_not_of( " \t\r\n" );
        return s.substr( a, b - a + 1 );
    }

    std::string StripComment( const std::string& s )
    {
        size_t h = s.find( '#' );
        return h == std::string::npos ? s : s.substr( 0, h );
    }

    bool ParseMtl( const std::string& filename, std::vector< MtlEntry >& out )
    {
        std::ifstream in( filename );
        if ( !in ) return false;
        std::string line;
        while ( std::getline( in, line ) )
        {
            line = Trim( StripComment( line ) );
            if ( line.empty() ) continue;
            std::istringstream ss( line );
            std::string key;
            ss >> key;
            if ( key == "newmtl" )
            {
                out.emplace_back();
                std::string rest;
                std::getline( ss, rest );
                out.back().name = Trim( rest );
                continue;
            }
            if ( out.empty() ) continue;
            MtlEntry& m = out.back();
            auto read3  = [&]( vec3& v ) { ss >> v.x >> v.y >> v.z; };
            if ( key == "Kd" ) read3( m.Kd );
            else if ( key == "Ks" ) read3( m.Ks );
            else if ( key == "Ke" ) read3( m.Ke );
            else if ( key == "Ns" ) ss >> m.Ns;
            else if ( key == "map_Kd" )
            {
                std::string rest;
                std::getline( ss, rest );
                m.mapKd = Trim( rest );
            }
        }
        return true;
    }

    // Texture lookup of LoadAssimpTexture (src/resource/model.cpp:23-80): the path as written under the
    // resource directory first, otherwise the first file with the same basename anywhere below it.
    std::string FindTextureFile( const std::string& resourceDir, const std::string& objDir, const std::string& name )
    {
        namespace fs = std::filesystem;
        std::error_code ec;
        if ( !resourceDir.empty() && fs::exists( resourceDir + name, ec ) ) return resourceDir + name;
        if ( fs::exists( objDir + name, ec ) ) return objDir + name;
        std::string normalized = name;
        for ( auto& c : normalized )
            if ( c == '\\' ) c = '/';
        const std::string basename = fs::path( normalized ).filename().string();
        for ( const std::string& root : { resourceDir, objDir } )
        {
            if ( root.empty() || !fs::exists( root, ec ) ) continue;
            for ( auto it = fs::recursive_directory_iterator( root, ec ); !ec && it != fs::recursive_directory_iterator(); it.increment( ec ) )
                if ( it->path().filename().string() == basename ) return fs::absolute( it->path() ).string();
        }
        return "";
    }

    struct VertexKey
    {
        uint32_t bits[8];
        bool operator<( const VertexKey& o ) const { return std::memcmp( bits, o.bits, sizeof( bits ) ) < 0; }
    };

    vec3 FaceNormal( const std::vector< vec3 >& positions, const std::vector< Corner >& face )
    {
        // Newell's method, the polygon normal aiProcess_GenNormals assigns to every corner of a face
        vec3 n( 0 );
        for ( size_t i = 0; i < face.size(); ++i )
        {
            vec3 a = positions[face[i].v];
            vec3 b = positions[face[( i + 1 ) % face.size()].v];
            n.x += ( a.y - b.y ) * ( a.z + b.z );
            n.y += ( a.z - b.z ) * ( a.x + b.x );
            n.z += ( a.x - b.x ) * ( a.y + b.y );
        }
        return normalize( n );
    }

} // namespace

// Per-vertex tangents from uv derivatives (what aiProcess_CalcTangentSpace provides), followed by the
// reference's own re-orthogonalisation against the normal (src/resource/model.cpp:204-211).
void ComputeUvTangents( LocalMesh& mesh )
{
    std::vector< vec3 > acc( mesh.vertices.size(), vec3( 0 ) );
    for ( size_t f = 0; f + 2 < mesh.indices.size(); f += 3 )
    {
        uint32_t i0 = mesh.indices[f], i1 = mesh.indices[f + 1], i2 = mesh.indices[f + 2];
        vec3 v = mesh.vertices[i1] - mesh.vertices[i0];
        vec3 w = mesh.vertices[i2] - mesh.vertices[i0];
        float sx = mesh.uvs[i1].x - mesh.uvs[i0].x, sy = mesh.uvs[i1].y - mesh.uvs[i0].y;
        float tx = mesh.uvs[i2].x - mesh.uvs[i0].x, ty = mesh.uvs[i2].y - mesh.uvs[i0].y;
        float dirCorrection = ( tx * sy - ty * sx ) < 0.0f ? -1.0f : 1.0f;
        if ( sx * ty == sy * tx )
        {
            sx = 0.0f; sy = 1.0f;
            tx = 1.0f; ty = 0.0f;
        }
        vec3 tangent = ( w * sy - v * ty ) * dirCorrection;
        for ( uint32_t idx : { i0, i1, i2 } )
        {
            vec3 n     = mesh.normals[idx];
            vec3 local = tangent - n * dot( tangent, n );
            float len  = length( local );
            if ( len > 0 ) acc[idx] += local * ( 1.0f / len );
        }
    }
    mesh.tangents.resize( mesh.vertices.size() );
    for ( size_t i = 0; i < acc.size(); ++i )
    {
        vec3 t = acc[i];
        if ( t == vec3( 0 ) ) t = vec3( 1, 0, 0 );
        vec3 n           = mesh.normals[i];
        mesh.tangents[i] = normalize( t - n * dot( n, t ) );
    }
}

bool LoadObjModel( HostScene& scene, const ModelLoadOptions& opts, std::string& error )
{
    namespace fs = std::filesystem;
    std::ifstream in( opts.filename );
    if ( !in )
    {
        error = "Error parsing model file '" + opts.filename + "': could not open file";
        return false;
    }
    const std::string objDir = fs::path( opts.filename ).parent_path().string() + "/";

    std::vector< vec3 > positions, fileNormals;
    std::vector< vec2 > texcoords;
    std::vector< FaceGroup > groups;
    std::vector< MtlEntry > mtl;
    std::string currentGroup = "default", currentMaterial;
    auto currentFaces = [&]() -> FaceGroup&
    {
        if ( groups.empty() || groups.back().name != currentGroup || groups.back().material != currentMaterial )
        {
            groups.emplace_back();
            groups.back().name     = currentGroup;
            groups.back().material = currentMaterial;
        }
        return groups.back();
    };

    std::string line;
    while ( std::getline( in, line ) )
    {
        line = Trim( StripComment( line ) );
        if ( line.empty() ) continue;
        std::istringstream ss( line );
        std::string key;
        ss >> key;
        if ( key == "v" )
        {
            vec3 p;
            ss >> p.x >> p.y >> p.z;
            positions.push_back( p );
        }
        else if ( key == "vt" )
        {
            vec2 t;
            ss >> t.x >> t.y;
            texcoords.push_back( t );
        }
        else if ( key == "vn" )
        {
            vec3 n;
            ss >> n.x >> n.y >> n.z;
            fileNormals.push_back( n );
        }
        else if ( key == "f" )
        {
            std::vector< Corner > face;
            std::string tok;
            while ( ss >> tok )
            {
                Corner c;
                int* slots[3] = { &c.v, &c.vt, &c.vn };
                const int counts[3] = { static_cast< int >( positions.size() ), static_cast< int >( texcoords.size() ),
                                        static_cast< int >( fileNormals.size() ) };
                size_t start = 0;
                for ( int k = 0; k < 3 && start <= tok.size(); ++k )
                {
                    size_t slash = tok.find( '/', start );
                    std::string part = tok.substr( start, slash == std::string::npos ? std::string::npos : slash - start );
                    if ( !part.empty() )
                    {
                        int idx   = std::stoi( part );
                        *slots[k] = idx < 0 ? counts[k] + idx : idx - 1;
                    }
                    if ( slash == std::string::npos ) break;
                    start = slash + 1;
                }
                if ( c.v < 0 || c.v >= static_cast< int >( positions.size() ) )
                {
                    error = "Error parsing model file '" + opts.filename + "': face references a missing vertex";
                    return false;
                }
                if ( c.vt >= static_cast< int >( texcoords.size() ) ) c.vt = -1;
                if ( c.vn >= static_cast< int >( fileNormals.size() ) ) c.vn = -1;
                face.push_back( c );
            }
            if ( face.size() >= 3 ) currentFaces().faces.push_back( std::move( face ) );
        }
        else if ( key == "g" || key == "o" )
        {
            std::string rest;
            std::getline( ss, rest );
            currentGroup = Trim( rest );
        }
        else if ( key == "usemtl" )
        {
            std::string rest;
            std::getline( ss, rest );
            currentMaterial = Trim( rest );
        }
        else if ( key == "mtllib" )
        {
            std::string rest;
            std::getline( ss, rest );
            ParseMtl( objDir + Trim( rest ), mtl );
        }
    }

    // materials (ParseMaterials, src/resource/model.cpp:82-129)
    std::map< std::string, int > materialIndex;
    for ( const MtlEntry& m : mtl )
    {
        PtMaterial pm{};
        pm.albedo[0] = m.Kd.x; pm.albedo[1] = m.Kd.y; pm.albedo[2] = m.Kd.z;
        pm.Ke[0] = m.Ke.x; pm.Ke[1] = m.Ke.y; pm.Ke[2] = m.Ke.z;
        pm.albedo_texture = -1;
        if ( !m.mapKd.empty() )
        {
            int tex = scene.FindTexture( m.mapKd );
            if ( tex < 0 )
            {
                std::string file = FindTextureFile( opts.resourceDir, objDir, m.mapKd );
                if ( file.empty() )
                {
                    error = "Could not find image file '" + m.mapKd + "'";
                    return false;
                }
                ImageRGBA8 img;
                if ( !ReadImageRGBA8( file, true, img, error ) ) return false;
                tex = scene.AddTexture( m.mapKd, img.width, img.height, img.pixels.data() );
            }
            pm.albedo_texture = tex;
        }
        materialIndex[m.name] = scene.AddMaterial( m.name, pm );
    }
    int defaultMaterial = -1;

    LocalModel model;
    model.name = opts.name;
    for ( const FaceGroup& g : groups )
    {
        if ( g.faces.empty() ) continue;
        LocalMesh mesh;
        mesh.name = g.name;
        auto it   = materialIndex.find( g.material );
        if ( it != materialIndex.end() )
            mesh.material = it->second;
        else
        {
            if ( defaultMaterial < 0 )
            {
                PtMaterial pm{};
                pm.albedo[0] = pm.albedo[1] = pm.albedo[2] = 0.6f;
                pm.albedo_texture = -1;
                defaultMaterial   = scene.AddMaterial( "DefaultMaterial", pm );
            }
            mesh.material = defaultMaterial;
        }

        bool hasUvs = false;
        std::map< VertexKey, uint32_t > unique; // aiProcess_JoinIdenticalVertices
        for ( const auto& face : g.faces )
        {
            vec3 faceN = FaceNormal( positions, face );
            std::vector< uint32_t > idx;
            for ( const Corner& c : face )
            {
                vec3 p = positions[c.v];
                vec3 n = c.vn >= 0 ? fileNormals[c.vn] : faceN;
                vec2 t = c.vt >= 0 ? texcoords[c.vt] : vec2{};
                hasUvs = hasUvs || c.vt >= 0;
                VertexKey key;
                const float vals[8] = { p.x, p.y, p.z, n.x, n.y, n.z, t.x, t.y };
                std::memcpy( key.bits, vals, sizeof( vals ) );
                auto found = unique.find( key );
                if ( found == unique.end() )
                {
                    found = unique.emplace( key, static_cast< uint32_t >( mesh.vertices.size() ) ).first;
                    mesh.vertices.push_back( p );
                    mesh.normals.push_back( n );
                    mesh.uvs.push_back( t );
                }
                idx.push_back( found->second );
            }
            for ( size_t k = 1; k + 1 < idx.size(); ++k ) // aiProcess_Triangulate, convex fan
            {
                mesh.indices.push_back( idx[0] );
                mesh.indices.push_back( idx[k] );
                mesh.indices.push_back( idx[k + 1] );
            }
        }

        if ( hasUvs ) ComputeUvTangents( mesh );
        if ( opts.recalculateNormals ) RecalculateNormals( mesh ); // src/resource/model.cpp:222-225
        if ( !hasUvs ) GenerateFallbackTangents( mesh );           // src/resource/model.cpp:227-251
        model.meshes.push_back( std::move( mesh ) );
    }
    if ( model.meshes.empty() )
    {
        error = "Error parsing model file '" + opts.filename + "': no faces";
        return false;
    }
    scene.models.push_back( std::move( model ) );
    return true;
}

} // namespace pth
