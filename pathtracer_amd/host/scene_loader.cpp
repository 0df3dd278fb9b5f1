#include "scene_loader.hpp"
#include "image_io.hpp"
#include "json_reader.hpp"
#include "obj_loader.hpp"
#include <map>
#include <stdexcept>

namespace pth
{

namespace
{

    using Log = std::function< void( const std::string& ) >;

    float Num( const JsonValue& v ) { return static_cast< float >( v.number ); } // rapidjson Get<float>: double -> float

    vec3 Vec3( const JsonValue& v )
    {
        if ( !v.IsArray() || v.array.size() != 3 ) throw std::runtime_error( "expected an array of 3 numbers" );
        return vec3( Num( v.array[0] ), Num( v.array[1] ), Num( v.array[2] ) );
    }

    // Dispatches the members of `obj` in order through `handlers`; unknown keys warn and are skipped.
    using Handler = std::function< void( const JsonValue& ) >;
    void ForEachMember( const JsonValue& obj, const std::map< std::string, Handler >& handlers, const Log& log )
    {
        for ( const auto& m : obj.members )
        {
            auto it = handlers.find( m.first );
            if ( it == handlers.end() )
            {
                if ( log ) log( "WARNING  '" + m.first + "' not found in mapping" );
            }
            else
                it->second( m.second );
        }
    }

    mat4 ParseTransform( const JsonValue& v, const Log& log ) // src/scene.cpp:21-39
    {
        vec3 position( 0 ), rotation( 0 ), scl( 1 );
        ForEachMember( v, {
            { "position", [&]( const JsonValue& x ) { position = Vec3( x ); } },
            { "rotation", [&]( const JsonValue& x ) { rotation = radians( Vec3( x ) ); } },
            { "scale",    [&]( const JsonValue& x ) { scl = Vec3( x ); } },
        }, log );
        return makeTransform( position, rotation, scl );
    }

} // namespace

bool LoadSceneFile( HostScene& scene, const std::string& filename, const SceneLoadOptions& opts, std::string& error )
{
    JsonValue doc;
    if ( !ParseJsonFile( filename, doc, error ) ) return false;
    if ( !doc.IsObject() )
    {
        error = "scene file is not a JSON object";
        return false;
    }
    const Log& log = opts.log;
    bool failed    = false;

    std::map< std::string, Handler > top = {
        { "BackgroundColor", [&]( const JsonValue& v ) { // src/scene.cpp:41-45
            if ( const JsonValue* c = v.Find( "color" ) ) scene.backgroundRadiance = Vec3( *c );
        } },
        { "BVH", [&]( const JsonValue& v ) { // src/scene.cpp:47-72
            ForEachMember( v, { { "splitMethod", [&]( const JsonValue& x ) {
                if ( x.string == "Middle" ) scene.splitMethod = SplitMethod::Middle;
                else if ( x.string == "EqualCounts" ) scene.splitMethod = SplitMethod::EqualCounts;
                else if ( x.string == "SAH" ) scene.splitMethod = SplitMethod::SAH;
                else if ( log ) log( "No SplitMethod with name '" + x.string + "' found! Using SAH" );
            } } }, log );
        } },
        { "Camera", [&]( const JsonValue& v ) { // src/scene.cpp:74-90
            ForEachMember( v, {
                { "position",     [&]( const JsonValue& x ) { vec3 p = Vec3( x ); scene.camera.position[0] = p.x; scene.camera.position[1] = p.y; scene.camera.position[2] = p.z; } },
                { "rotation",     [&]( const JsonValue& x ) { scene.cameraRotation = radians( Vec3( x ) ); } },
                { "vfov",         [&]( const JsonValue& x ) { scene.camera.vfov = radians( Num( x ) ); } },
                { "aspectRatio",  [&]( const JsonValue& x ) { scene.camera.aspect_ratio = Num( x ); } },
                { "exposure",     [&]( const JsonValue& x ) { scene.camera.exposure = Num( x ); } },
                { "gamma",        [&]( const JsonValue& x ) { scene.camera.gamma = Num( x ); } },
                { "antialiasing", [&]( const JsonValue& ) { /* parsed by the reference but never used by Render (src/path_tracer.cpp:188) */ } },
            }, log );
            scene.UpdateCameraOrientation();
        } },
        { "DirectionalLight", [&]( const JsonValue& v ) { // src/scene.cpp:92-102
            PtLight l{};
            l.type         = PT_LIGHT_DIRECTIONAL;
            l.n_samples    = 1;
            l.direction[1] = -1;
            l.prim         = -1;
            ForEachMember( v, {
                { "Lemit",     [&]( const JsonValue& x ) { vec3 e = Vec3( x ); l.Lemit[0] = e.x; l.Lemit[1] = e.y; l.Lemit[2] = e.z; } },
                { "direction", [&]( const JsonValue& x ) { vec3 d = normalize( Vec3( x ) ); l.direction[0] = d.x; l.direction[1] = d.y; l.direction[2] = d.z; } },
            }, log );
            scene.AddLight( l );
        } },
        { "LogFile", [&]( const JsonValue& ) { /* extra log sink (src/scene.cpp:104-107): left to the caller */ } },
        { "Material", [&]( const JsonValue& v ) { // src/scene.cpp:109-133
            PtMaterial m{};
            m.albedo_texture = -1;
            std::string name;
            ForEachMember( v, {
                { "name",   [&]( const JsonValue& x ) { name = x.string; } },
                { "albedo", [&]( const JsonValue& x ) { vec3 a = Vec3( x ); m.albedo[0] = a.x; m.albedo[1] = a.y; m.albedo[2] = a.z; } },
                { "Ks",     [&]( const JsonValue& ) {} },
                { "Ns",     [&]( const JsonValue& ) {} },
                { "Tr",     [&]( const JsonValue& ) {} },
                { "ior",    [&]( const JsonValue& ) {} },
                { "albedoTexture", [&]( const JsonValue& x ) {
                    m.albedo_texture = scene.FindTexture( x.string );
                    if ( m.albedo_texture < 0 && log ) log( "No texture with name '" + x.string + "' found in resource manager!" );
                } },
            }, log );
            scene.AddMaterial( name, m );
        } },
        { "MaxDepth", [&]( const JsonValue& v ) { scene.maxDepth = static_cast< int >( v.number ); } },
        { "Model", [&]( const JsonValue& v ) { // src/scene.cpp:140-157
            ModelLoadOptions mo;
            mo.resourceDir = opts.resourceDir;
            ForEachMember( v, {
                { "name",               [&]( const JsonValue& x ) { mo.name = x.string; } },
                { "filename",           [&]( const JsonValue& x ) { mo.filename = opts.resourceDir + x.string; } },
                { "recalculateNormals", [&]( const JsonValue& x ) { mo.recalculateNormals = x.boolean; } },
            }, log );
            std::string err;
            if ( !LoadObjModel( scene, mo, err ) && log ) log( "ERROR    " + err );
        } },
        { "ModelInstance", [&]( const JsonValue& v ) { // src/scene.cpp:159-191
            mat4 transform( 1.0f );
            std::string modelName, materialName;
            ForEachMember( v, {
                { "transform", [&]( const JsonValue& x ) { transform = ParseTransform( x, log ); } },
                { "model",     [&]( const JsonValue& x ) { modelName = x.string; } },
                { "material",  [&]( const JsonValue& x ) { materialName = x.string; } },
            }, log );
            const LocalModel* model = scene.FindModel( modelName );
            if ( !model )
            {
                // the reference dereferences a null model here (src/scene.cpp:178-186)
                error  = "ModelInstance refers to model '" + modelName + "', which was not loaded";
                failed = true;
                return;
            }
            int material = -1;
            if ( !materialName.empty() )
            {
                material = scene.FindMaterial( materialName );
                if ( material < 0 )
                {
                    error  = "ModelInstance refers to unknown material '" + materialName + "'";
                    failed = true;
                    return;
                }
            }
            for ( const LocalMesh& mesh : model->meshes ) scene.AddMeshInstance( mesh, transform, material );
        } },
        { "OutputImageData", [&]( const JsonValue& v ) { // src/scene.cpp:193-207
            ForEachMember( v, {
                { "filename",   [&]( const JsonValue& x ) { scene.outputImageFilename = x.string; } },
                { "resolution", [&]( const JsonValue& x ) {
                    if ( !x.IsArray() || x.array.size() < 2 ) throw std::runtime_error( "resolution needs two numbers" );
                    scene.imageWidth  = static_cast< int >( x.array[0].number );
                    scene.imageHeight = static_cast< int >( x.array[1].number );
                } },
            }, log );
        } },
        { "PointLight", [&]( const JsonValue& v ) { // src/scene.cpp:209-219
            PtLight l{};
            l.type      = PT_LIGHT_POINT;
            l.n_samples = 1;
            l.prim      = -1;
            ForEachMember( v, {
                { "Lemit",    [&]( const JsonValue& x ) { vec3 e = Vec3( x ); l.Lemit[0] = e.x; l.Lemit[1] = e.y; l.Lemit[2] = e.z; } },
                { "position", [&]( const JsonValue& x ) { vec3 p = Vec3( x ); l.position[0] = p.x; l.position[1] = p.y; l.position[2] = p.z; } },
            }, log );
            scene.AddLight( l );
        } },
        { "SamplesPerAreaLight", [&]( const JsonValue& v ) { scene.numSamplesPerAreaLight = static_cast< int >( v.number ); } },
        { "SamplesPerPixel", [&]( const JsonValue& v ) { // src/scene.cpp:226-233
            scene.numSamplesPerPixel.clear();
            for ( const auto& e : v.array ) scene.numSamplesPerPixel.push_back( static_cast< int >( e.number ) );
        } },
        { "Skybox", [&]( const JsonValue& v ) { // src/scene.cpp:235-258
            std::string faces[6];
            bool flip = true;
            ForEachMember( v, {
                { "name",           [&]( const JsonValue& ) {} },
                { "right",          [&]( const JsonValue& x ) { faces[0] = x.string; } },
                { "left",           [&]( const JsonValue& x ) { faces[1] = x.string; } },
                { "top",            [&]( const JsonValue& x ) { faces[2] = x.string; } },
                { "bottom",         [&]( const JsonValue& x ) { faces[3] = x.string; } },
                { "back",           [&]( const JsonValue& x ) { faces[4] = x.string; } },
                { "front",          [&]( const JsonValue& x ) { faces[5] = x.string; } },
                { "flipVertically", [&]( const JsonValue& x ) { flip = x.boolean; } },
            }, log );
            std::vector< float > pixels;
            int w = 0, h = 0;
            for ( int f = 0; f < 6; ++f ) // Skybox::Load, src/resource/skybox.cpp:19-61
            {
                std::vector< float > face;
                std::string err;
                int fw = 0, fh = 0;
                // stbi_loadf with hdr scale and gamma forced to 1: byte / 255, or a Radiance file's own floats
                if ( !ReadImageRGBA32F( opts.resourceDir + faces[f], flip, fw, fh, face, err ) )
                {
                    if ( log ) log( "ERROR    " + err );
                    return;
                }
                if ( f == 0 ) { w = fw; h = fh; }
                if ( fw != w || fh != h )
                {
                    if ( log ) log( "ERROR    skybox faces differ in size" );
                    return;
                }
                pixels.insert( pixels.end(), face.begin(), face.end() );
            }
            scene.skybox       = std::move( pixels );
            scene.skyboxWidth  = w;
            scene.skyboxHeight = h;
        } },
        { "Sphere", [&]( const JsonValue& v ) { // src/scene.cpp:260-274
            vec3 position( 0 ), rotation( 0 );
            float radius = 1;
            int material = -1;
            ForEachMember( v, {
                { "position", [&]( const JsonValue& x ) { position = Vec3( x ); } },
                { "rotation", [&]( const JsonValue& x ) { rotation = radians( Vec3( x ) ); } },
                { "radius",   [&]( const JsonValue& x ) { radius = Num( x ); } },
                { "material", [&]( const JsonValue& x ) { material = scene.FindMaterial( x.string ); } },
            }, log );
            if ( material < 0 )
            {
                error  = "Sphere has no (known) material"; // the reference would dereference a null material at the first hit
                failed = true;
                return;
            }
            scene.AddSphere( position, rotation, radius, material );
        } },
        { "Texture", [&]( const JsonValue& v ) { // src/scene.cpp:276-292
            std::string name, file;
            bool flip = true;
            ForEachMember( v, {
                { "name",           [&]( const JsonValue& x ) { name = x.string; } },
                { "filename",       [&]( const JsonValue& x ) { file = opts.resourceDir + x.string; } },
                { "flipVertically", [&]( const JsonValue& x ) { flip = x.boolean; } },
            }, log );
            ImageRGBA8 img;
            std::string err;
            if ( ReadImageRGBA8( file, flip, img, err ) )
                scene.AddTexture( name, img.width, img.height, img.pixels.data() );
            else if ( log )
                log( "ERROR    " + err );
        } },
    };

    try
    {
        for ( const auto& m : doc.members )
        {
            auto it = top.find( m.first );
            if ( it == top.end() )
            {
                if ( log ) log( "WARNING  '" + m.first + "' not found in mapping" );
                continue;
            }
            it->second( m.second );
            if ( failed ) return false;
        }
        if ( scene.shapes.empty() )
        {
            error = "scene has no shapes";
            return false;
        }
        scene.BuildAccel();
    }
    catch ( const std::exception& e )
    {
        error = e.what();
        return false;
    }
    return true;
}

} // namespace pth
