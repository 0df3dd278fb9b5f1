#include "generators.hpp"
#include <algorithm>
#include <cmath>
#include <functional>
#include <stdexcept>

namespace pth
{

namespace
{

    // ---------------------------------------------------------------- deterministic noise
    uint32_t Hash( uint32_t x )
    {
        x ^= x >> 16;
        x *= 0x7feb352du;
        x ^= x >> 15;
        x *= 0x846ca68bu;
        x ^= x >> 16;
        return x;
    }

    struct Rng
    {
        uint32_t state;
        explicit Rng( uint32_t seed ) : state( Hash( seed ^ 0x9e3779b9u ) | 1u ) {}
        uint32_t Next()
        {
            state = state * 747796405u + 2891336453u;
            return Hash( state );
        }
        float Uniform() { return ( Next() >> 8 ) * ( 1.0f / 16777216.0f ); }
        float Range( float a, float b ) { return a + ( b - a ) * Uniform(); }
    };

    float Lattice( int x, int y, uint32_t seed ) { return ( Hash( x * 73856093u ^ y * 19349663u ^ seed ) >> 8 ) * ( 1.0f / 16777216.0f ); }

    float ValueNoise( float x, float y, uint32_t seed )
    {
        int xi = static_cast< int >( std::floor( x ) ), yi = static_cast< int >( std::floor( y ) );
        float fx = x - xi, fy = y - yi;
        fx = fx * fx * ( 3 - 2 * fx );
        fy = fy * fy * ( 3 - 2 * fy );
        float a = Lattice( xi, yi, seed ), b = Lattice( xi + 1, yi, seed );
        float c = Lattice( xi, yi + 1, seed ), d = Lattice( xi + 1, yi + 1, seed );
        return ( a + ( b - a ) * fx ) + ( ( c + ( d - c ) * fx ) - ( a + ( b - a ) * fx ) ) * fy;
    }

    float Fbm( float x, float y, uint32_t seed )
    {
        float s = 0, amp = 0.5f;
        for ( int o = 0; o < 4; ++o )
        {
            s += amp * ValueNoise( x, y, seed + o * 101u );
            x *= 2;
            y *= 2;
            amp *= 0.5f;
        }
        return s / 0.9375f;
    }

    // ---------------------------------------------------------------- procedural diffuse textures
    enum Pattern { Bricks, Tiles, Stripes, Marble, Speckle };

    void MakeTexture( std::vector< uint8_t >& out, int size, Pattern pattern, vec3 base, vec3 accent, uint32_t seed )
    {
        out.resize( static_cast< size_t >( size ) * size * 4 );
        for ( int y = 0; y < size; ++y )
        {
            for ( int x = 0; x < size; ++x )
            {
                float u = ( x + 0.5f ) / size, v = ( y + 0.5f ) / size;
                float k = 0;
                switch ( pattern )
                {
                    case Bricks:
                    {
                        float row = std::floor( v * 8 );
                        float uu  = u * 4 + ( static_cast< int >( row ) & 1 ? 0.5f : 0.0f );
                        float fu = uu - std::floor( uu ), fv = v * 8 - row;
                        bool mortar = fu < 0.04f || fv < 0.08f;
                        k = mortar ? 1.0f : 0.35f * Lattice( static_cast< int >( std::floor( uu ) ), static_cast< int >( row ), seed );
                        break;
                    }
                    case Tiles:
                    {
                        int cx = static_cast< int >( u * 8 ), cy = static_cast< int >( v * 8 );
                        k = ( ( cx + cy ) & 1 ) ? 0.8f : 0.1f;
                        k += 0.15f * Fbm( u * 32, v * 32, seed );
                        break;
                    }
                    case Stripes:
                        k = 0.5f + 0.5f * std::sin( u * 6.2831853f * 12 ) * ( 0.7f + 0.3f * Fbm( u * 4, v * 16, seed ) );
                        break;
                    case Marble:
                        k = 0.5f + 0.5f * std::sin( ( u + 2 * Fbm( u * 4, v * 4, seed ) ) * 6.2831853f * 3 );
                        break;
                    case Speckle:
                    default:
                        k = Fbm( u * 64, v * 64, seed );
                        break;
                }
                k          = std::min( 1.0f, std::max( 0.0f, k ) );
                vec3 c     = base + ( accent - base ) * k;
                uint8_t* p = &out[4 * ( static_cast< size_t >( y ) * size + x )];
                p[0]       = static_cast< uint8_t >( 255.0f * std::min( 1.0f, std::max( 0.0f, c.x ) ) );
                p[1]       = static_cast< uint8_t >( 255.0f * std::min( 1.0f, std::max( 0.0f, c.y ) ) );
                p[2]       = static_cast< uint8_t >( 255.0f * std::min( 1.0f, std::max( 0.0f, c.z ) ) );
                p[3]       = 255;
            }
        }
    }

    // ---------------------------------------------------------------- parametric surface patches
    using Surface = std::function< vec3( float, float ) >; // (u, v) in [0,1]^2 -> position

    struct PatchSpec
    {
        Surface surface;
        float nu, nv;   // base tessellation (scaled by the detail factor)
        bool closedU;   // u = 0 and u = 1 describe the same curve: share those vertices
        int material;
        float uvScaleU, uvScaleV;
    };

    int Segments( float base, float detail ) { return std::max( 1, static_cast< int >( base * detail + 0.5f ) ); }

    long long PatchTriangles( const PatchSpec& p, float detail )
    {
        return 2ll * Segments( p.nu, detail ) * Segments( p.nv, detail );
    }

    // Appends the tessellated patch to `mesh` with smooth normals of its own (patches do not share vertices).
    void AppendPatch( LocalMesh& mesh, const PatchSpec& p, float detail )
    {
        const int nu = Segments( p.nu, detail ), nv = Segments( p.nv, detail );
        const int cols = p.closedU ? nu : nu + 1;
        LocalMesh part;
        part.vertices.reserve( static_cast< size_t >( cols ) * ( nv + 1 ) );
        for ( int j = 0; j <= nv; ++j )
        {
            for ( int i = 0; i < cols; ++i )
            {
                float u = static_cast< float >( i ) / nu, v = static_cast< float >( j ) / nv;
                part.vertices.push_back( p.surface( u, v ) );
                part.uvs.push_back( vec2{ u * p.uvScaleU, v * p.uvScaleV } );
            }
        }
        for ( int j = 0; j < nv; ++j )
        {
            for ( int i = 0; i < nu; ++i )
            {
                uint32_t i1 = ( i + 1 ) % cols;
                uint32_t a = j * cols + i, b = j * cols + i1, c = ( j + 1 ) * cols + i1, d = ( j + 1 ) * cols + i;
                part.indices.insert( part.indices.end(), { a, b, c, a, c, d } );
            }
        }
        RecalculateNormals( part );
        for ( auto& n : part.normals ) // degenerate poles can produce NaN normals: replace them
            if ( !( n.x == n.x ) ) n = vec3( 0, 1, 0 );

        const uint32_t base = static_cast< uint32_t >( mesh.vertices.size() );
        mesh.vertices.insert( mesh.vertices.end(), part.vertices.begin(), part.vertices.end() );
        mesh.normals.insert( mesh.normals.end(), part.normals.begin(), part.normals.end() );
        mesh.uvs.insert( mesh.uvs.end(), part.uvs.begin(), part.uvs.end() );
        for ( uint32_t idx : part.indices ) mesh.indices.push_back( base + idx );
    }

    // Rectangle origin + u * eu + v * ev whose face normal (counter-clockwise winding) points along `facing`;
    // the u axis is reversed if the given axes would face the other way.
    Surface Plane( vec3 origin, vec3 eu, vec3 ev, vec3 facing )
    {
        if ( dot( cross( eu, ev ), facing ) < 0 )
        {
            origin = origin + eu;
            eu     = -eu;
        }
        return [=]( float u, float v ) { return origin + eu * u + ev * v; };
    }

    Surface Cylinder( vec3 base, float radius, float height, float bulge )
    {
        return [=]( float u, float v ) {
            float a = -u * 6.2831853f; // clockwise seen from above: the face normals point outwards
            float r = radius * ( 1.0f + bulge * std::sin( v * 3.1415927f ) );
            return base + vec3( r * std::cos( a ), height * v, r * std::sin( a ) );
        };
    }

    // Half-ring arch in the plane spanned by `along` (unit) and +y, extruded by `depth` along `across`.
    Surface Arch( vec3 centre, vec3 along, vec3 across, float radius, float depth )
    {
        return [=]( float u, float v ) {
            float a = u * 3.1415927f;
            return centre + along * ( radius * std::cos( a ) ) + vec3( 0, radius * std::sin( a ), 0 ) + across * ( depth * ( v - 0.5f ) );
        };
    }

    Surface Curtain( vec3 topLeft, vec3 along, float width, float drop, float wave, float phase )
    {
        return [=]( float u, float v ) {
            vec3 side = cross( along, vec3( 0, 1, 0 ) );
            float w   = wave * std::sin( u * 6.2831853f * 5 + phase ) * ( 0.3f + 0.7f * v );
            return topLeft + along * ( width * u ) + vec3( 0, -drop * v, 0 ) + side * w;
        };
    }

    // Surface of revolution around +y with a profile radius(v).
    Surface Lathe( vec3 base, float height, std::function< float( float ) > radius )
    {
        return [=]( float u, float v ) {
            float a = -u * 6.2831853f, r = radius( v ); // outward normals, as for Cylinder
            return base + vec3( r * std::cos( a ), height * v, r * std::sin( a ) );
        };
    }

    // A tangent that came out parallel to its normal normalises to NaN: replace it by any perpendicular.
    void SanitizeTangents( LocalMesh& mesh )
    {
        for ( size_t i = 0; i < mesh.tangents.size(); ++i )
        {
            vec3& t = mesh.tangents[i];
            if ( t.x == t.x && t.y == t.y && t.z == t.z ) continue;
            vec3 n    = mesh.normals[i];
            vec3 axis = std::fabs( n.x ) < 0.9f ? vec3( 1, 0, 0 ) : vec3( 0, 1, 0 );
            t         = normalize( axis - n * dot( n, axis ) );
        }
    }

    void SetCamera( HostScene& s, vec3 position, vec3 rotationDeg, float vfovDeg, float aspect, float exposure, float gamma )
    {
        s.camera.position[0] = position.x;
        s.camera.position[1] = position.y;
        s.camera.position[2] = position.z;
        s.cameraRotation      = radians( rotationDeg );
        s.camera.vfov         = radians( vfovDeg );
        s.camera.aspect_ratio = aspect;
        s.camera.exposure     = exposure;
        s.camera.gamma        = gamma;
        s.UpdateCameraOrientation();
    }

    void AddDirectionalLight( HostScene& s, vec3 direction, vec3 Lemit )
    {
        PtLight l{};
        l.type      = PT_LIGHT_DIRECTIONAL;
        l.n_samples = 1;
        l.prim      = -1;
        vec3 d      = normalize( direction );
        l.direction[0] = d.x; l.direction[1] = d.y; l.direction[2] = d.z;
        l.Lemit[0] = Lemit.x; l.Lemit[1] = Lemit.y; l.Lemit[2] = Lemit.z;
        s.AddLight( l );
    }

} // namespace

void GenerateAtrium( HostScene& scene, uint32_t seed, int targetTriangles, int textureSize )
{
    if ( targetTriangles < 2000 ) throw std::runtime_error( "GenerateAtrium: need at least 2000 triangles" );
    Rng rng( seed );

    // ---- 24 textured materials, all Kd 0.4704 like resources/sponza/sponza.mtl
    enum { FLOOR, BRICKS, ARCH, COLUMN_A, COLUMN_B, COLUMN_C, CEILING, ROOF, DETAILS, FABRIC_R, FABRIC_G, FABRIC_B,
           CURTAIN_R, CURTAIN_G, CURTAIN_B, VASE, VASE_ROUND, VASE_HANGING, CHAIN, FLAGPOLE, LION, BACKGROUND, THORN, PLANT, NUM_MATERIALS };
    const Pattern patterns[NUM_MATERIALS] = { Tiles, Bricks, Marble, Marble, Marble, Speckle, Tiles, Bricks, Speckle, Stripes, Stripes, Stripes,
                                              Stripes, Stripes, Stripes, Marble, Marble, Speckle, Speckle, Speckle, Marble, Bricks, Speckle, Speckle };
    const vec3 bases[NUM_MATERIALS] = {
        { .75f, .72f, .65f }, { .72f, .60f, .50f }, { .80f, .78f, .72f }, { .82f, .80f, .75f }, { .78f, .74f, .70f }, { .70f, .70f, .68f },
        { .85f, .83f, .78f }, { .65f, .40f, .30f }, { .60f, .58f, .55f }, { .80f, .15f, .12f }, { .15f, .60f, .20f }, { .15f, .25f, .75f },
        { .75f, .10f, .10f }, { .10f, .55f, .15f }, { .12f, .20f, .70f }, { .65f, .55f, .40f }, { .55f, .45f, .35f }, { .50f, .50f, .45f },
        { .35f, .35f, .38f }, { .45f, .40f, .35f }, { .75f, .72f, .65f }, { .70f, .62f, .55f }, { .25f, .45f, .15f }, { .20f, .55f, .18f } };
    std::vector< int > materialIds( NUM_MATERIALS );
    std::vector< uint8_t > pixels;
    for ( int m = 0; m < NUM_MATERIALS; ++m )
    {
        vec3 accent = bases[m] * 0.45f;
        MakeTexture( pixels, textureSize, patterns[m], bases[m], accent, Hash( seed * 31u + m ) );
        std::string name = "atrium_" + std::to_string( m );
        int tex          = scene.AddTexture( name, textureSize, textureSize, pixels.data() );
        PtMaterial pm{};
        pm.albedo[0] = pm.albedo[1] = pm.albedo[2] = 0.4704f;
        pm.albedo_texture = tex;
        materialIds[m]    = scene.AddMaterial( name, pm );
    }

    // ---- geometry, in centimetres (the instance is scaled by 0.01 like resources/scenes/sponza.json:23)
    std::vector< PatchSpec > patches;
    auto add = [&]( Surface s, float nu, float nv, bool closedU, int mat, float su = 1, float sv = 1 ) {
        patches.push_back( PatchSpec{ std::move( s ), nu, nv, closedU, mat, su, sv } );
    };
    const float X = 1800, Z = 1000, courtZ = 400, H1 = 450, H2 = 950, HR = 1000;

    add( Plane( { -X, 0, Z }, { 2 * X, 0, 0 }, { 0, 0, -2 * Z }, { 0, 1, 0 } ), 72, 40, false, FLOOR, 18, 10 );                       // ground floor
    for ( int side = -1; side <= 1; side += 2 )
    {
        float z0 = side * courtZ, z1 = side * Z;
        add( Plane( { -X, H1, z0 }, { 2 * X, 0, 0 }, { 0, 0, z1 - z0 }, { 0, 1, 0 } ), 72, 12, false, CEILING, 18, 3 );              // gallery floor (top)
        add( Plane( { -X, H1 - 40, z1 }, { 2 * X, 0, 0 }, { 0, 0, z0 - z1 }, { 0, -1, 0 } ), 72, 12, false, CEILING, 18, 3 );         // gallery floor (underside)
        add( Plane( { -X, H1 - 40, z0 }, { 2 * X, 0, 0 }, { 0, 40, 0 }, { 0, 0, -1.0f * side } ), 72, 2, false, DETAILS, 18, 1 );               // slab edge
        add( Plane( { -X, H2, z1 }, { 2 * X, 0, 0 }, { 0, 0, z0 - z1 }, { 0, -1, 0 } ), 72, 12, false, ROOF, 18, 3 );                 // aisle roof underside
        add( Plane( { -X, HR, z0 }, { 2 * X, 0, 0 }, { 0, 0, z1 - z0 }, { 0, 1, 0 } ), 36, 6, false, ROOF, 18, 3 );                  // aisle roof top
        add( Plane( { -X, H2, z0 }, { 2 * X, 0, 0 }, { 0, HR - H2, 0 }, { 0, 0, -1.0f * side } ), 72, 2, false, DETAILS, 18, 1 );               // roof edge
        add( Plane( { side * X, 0, -side * Z }, { 0, 0, side * 2 * Z }, { 0, HR, 0 }, { -1.0f * side, 0, 0 } ), 40, 24, false, BACKGROUND, 10, 5 ); // end wall (x = +-X)
        add( Plane( { -X, 0, z1 }, { 2 * X, 0, 0 }, { 0, HR, 0 }, { 0, 0, -1.0f * side } ), 72, 24, false, BRICKS, 18, 5 ); // long wall
        for ( int i = 0; i <= 10; ++i )
        {
            float x = -1500.0f + 300.0f * i;
            add( Cylinder( { x, 0, z0 }, 28, H1 - 40, 0.08f ), 24, 20, true, i % 3 == 0 ? COLUMN_A : ( i % 3 == 1 ? COLUMN_B : COLUMN_C ), 2, 6 );
            add( Cylinder( { x, H1, z0 }, 22, H2 - H1, 0.06f ), 20, 18, true, i % 2 ? COLUMN_A : COLUMN_C, 2, 6 );
            add( Lathe( { x, 0, z0 }, 40, []( float v ) { return 45.0f - 12.0f * v; } ), 16, 3, true, DETAILS, 2, 1 );            // base
            add( Lathe( { x, H1 - 80, z0 }, 40, []( float v ) { return 30.0f + 16.0f * v; } ), 16, 3, true, DETAILS, 2, 1 );     // capital
            if ( i < 10 )
            {
                add( Arch( { x + 150, H1 - 190, z0 }, { 1, 0, 0 }, { 0, 0, 1 }, 122, 60 ), 24, 3, false, ARCH, 4, 1 );            // lower arch
                add( Arch( { x + 150, H2 - 150, z0 }, { 1, 0, 0 }, { 0, 0, 1 }, 128, 45 ), 20, 3, false, ARCH, 4, 1 );            // upper arch
                int fabric = ( i % 3 == 0 ) ? CURTAIN_R : ( i % 3 == 1 ? CURTAIN_G : CURTAIN_B );
                if ( i % 2 == 0 )
                    add( Curtain( { x + 30, H2 - 160, z0 + side * 12.0f }, { 1, 0, 0 }, 240, 280, 14, rng.Range( 0, 6.28f ) ), 40, 24, false, fabric, 3, 3 );
                else
                    add( Curtain( { x + 30, H1 - 200, z0 - side * 12.0f }, { 1, 0, 0 }, 240, 200, 10, rng.Range( 0, 6.28f ) ), 36, 18, false, fabric - 3, 3, 3 );
            }
        }
    }
    for ( int end = -1; end <= 1; end += 2 ) // end galleries close the roof ring; the courtyard stays open to the sky
    {
        float x0 = end * 1200.0f, x1 = end * X;
        add( Plane( { x0, H2, -courtZ }, { x1 - x0, 0, 0 }, { 0, 0, 2 * courtZ }, { 0, -1, 0 } ), 12, 16, false, ROOF, 3, 4 );
        add( Plane( { x0, HR, courtZ }, { x1 - x0, 0, 0 }, { 0, 0, -2 * courtZ }, { 0, 1, 0 } ), 8, 8, false, ROOF, 3, 4 );
        add( Plane( { x0, H2, -courtZ }, { 0, 0, 2 * courtZ }, { 0, HR - H2, 0 }, { -1.0f * end, 0, 0 } ), 16, 2, false, DETAILS, 4, 1 );
        add( Plane( { x0, H1, -courtZ }, { x1 - x0, 0, 0 }, { 0, 0, 2 * courtZ }, { 0, 1, 0 } ), 12, 16, false, CEILING, 3, 4 );
        add( Plane( { x0, H1 - 40, courtZ }, { x1 - x0, 0, 0 }, { 0, 0, -2 * courtZ }, { 0, -1, 0 } ), 12, 16, false, CEILING, 3, 4 );
        // lion-head stand-in: bumpy sphere on the end wall
        vec3 c( end * ( X - 60.0f ), 260, 0 );
        uint32_t ns = Hash( seed + 77u * ( end + 2 ) );
        add( [=]( float u, float v ) {
            float a = u * 6.2831853f, b = v * 3.1415927f;
            float r = 70.0f * ( 1.0f + 0.25f * Fbm( u * 8, v * 8, ns ) );
            return c + vec3( r * std::sin( b ) * std::cos( a ), r * std::cos( b ), r * std::sin( b ) * std::sin( a ) );
        }, 96, 64, true, LION, 2, 2 );
        // flagpoles reaching into the courtyard
        for ( int k = -1; k <= 1; ++k )
            add( Cylinder( { end * 900.0f, H1 + 80, k * 250.0f }, 4, 300, 0 ), 10, 8, true, FLAGPOLE, 1, 4 );
    }
    // vases on the courtyard floor and hanging from the gallery
    for ( int i = 0; i < 8; ++i )
    {
        float x = -1050.0f + 300.0f * i, z = ( i & 1 ) ? 260.0f : -260.0f;
        add( Lathe( { x, 0, z }, 120, []( float v ) { return 18.0f + 34.0f * std::sin( 3.1415927f * std::min( 1.0f, v * 1.15f ) ); } ),
             40, 28, true, ( i & 1 ) ? VASE : VASE_ROUND, 2, 2 );
        add( Lathe( { x + 150, H1 - 170, -z * 1.45f }, 70, []( float v ) { return 6.0f + 22.0f * std::sin( 3.1415927f * v ); } ),
             28, 18, true, VASE_HANGING, 2, 2 );
        add( Cylinder( { x + 150, H1 - 100, -z * 1.45f }, 1.5f, 60, 0 ), 6, 10, true, CHAIN, 1, 8 );
    }

    // ---- choose the detail factor that fits the triangle budget, keep 1.5 % for the plant leaves
    const long long budget = targetTriangles - std::max( 64, targetTriangles / 64 );
    auto total = [&]( float detail ) {
        long long t = 0;
        for ( const auto& p : patches ) t += PatchTriangles( p, detail );
        return t;
    };
    float lo = 0.01f, hi = 64.0f;
    for ( int it = 0; it < 60; ++it )
    {
        float mid = 0.5f * ( lo + hi );
        ( total( mid ) <= budget ? lo : hi ) = mid;
    }
    const float detail = lo;

    LocalModel model;
    model.name = "atrium";
    model.meshes.resize( NUM_MATERIALS );
    for ( int m = 0; m < NUM_MATERIALS; ++m )
    {
        model.meshes[m].name     = "atrium_mesh_" + std::to_string( m );
        model.meshes[m].material = materialIds[m];
    }
    long long emitted = 0;
    for ( const auto& p : patches )
    {
        AppendPatch( model.meshes[p.material], p, detail );
        emitted += PatchTriangles( p, detail );
    }

    // ---- plant leaves around the vases take the remaining triangles exactly
    LocalMesh& plant = model.meshes[PLANT];
    for ( long long k = emitted; k < targetTriangles; ++k )
    {
        int vase   = static_cast< int >( rng.Next() % 8 );
        float x    = -1050.0f + 300.0f * vase, z = ( vase & 1 ) ? 260.0f : -260.0f;
        vec3 root( x + rng.Range( -25, 25 ), 120 + rng.Range( 0, 60 ), z + rng.Range( -25, 25 ) );
        vec3 dir = normalize( vec3( rng.Range( -1, 1 ), rng.Range( 0.2f, 1 ), rng.Range( -1, 1 ) ) );
        vec3 side = normalize( cross( dir, vec3( 0.3f, 1, 0.2f ) ) );
        float len = rng.Range( 20, 60 ), wid = rng.Range( 4, 10 );
        uint32_t base = static_cast< uint32_t >( plant.vertices.size() );
        vec3 a = root - side * wid, b = root + side * wid, c = root + dir * len;
        vec3 n = normalize( cross( b - a, c - a ) );
        plant.vertices.insert( plant.vertices.end(), { a, b, c } );
        plant.normals.insert( plant.normals.end(), { n, n, n } );
        plant.uvs.insert( plant.uvs.end(), { vec2{ 0, 0 }, vec2{ 1, 0 }, vec2{ 0.5f, 1 } } );
        plant.indices.insert( plant.indices.end(), { base, base + 1, base + 2 } );
    }

    mat4 instance = makeTransform( vec3( 0 ), vec3( 0 ), vec3( 0.01f ) );
    for ( auto& mesh : model.meshes )
    {
        if ( mesh.indices.empty() ) continue;
        ComputeUvTangents( mesh );
        SanitizeTangents( mesh );
        scene.AddMeshInstance( mesh, instance, -1 );
    }
    scene.models.push_back( std::move( model ) );

    // resources/scenes/sponza.json:2-18
    SetCamera( scene, vec3( -9, 2, 0 ), vec3( 0, -90, 0 ), 45, 1, 1.0f, 2.2f );
    scene.splitMethod            = SplitMethod::SAH;
    scene.numSamplesPerAreaLight = 1;
    scene.numSamplesPerPixel     = { 16 };
    scene.maxDepth               = 5;
    scene.imageWidth = scene.imageHeight = 500;
    scene.backgroundRadiance     = vec3( 0 );
    scene.outputImageFilename    = "sponza.png";
    AddDirectionalLight( scene, vec3( 0, -6, 0 ), vec3( 12, 10, 8 ) );
    scene.BuildAccel();
}

void GenerateKnot( HostScene& scene, uint32_t seed, int targetTriangles )
{
    if ( targetTriangles < 64 ) throw std::runtime_error( "GenerateKnot: need at least 64 triangles" );
    PtMaterial green{};
    green.albedo[0] = 0.1f; green.albedo[1] = 0.8f; green.albedo[2] = 0.2f;
    green.albedo_texture = -1;
    PtMaterial floor{};
    floor.albedo[0] = floor.albedo[1] = floor.albedo[2] = 0.6f;
    floor.albedo_texture = -1;
    int greenId = scene.AddMaterial( "green", green );
    int floorId = scene.AddMaterial( "floor", floor );

    // tube around a (2,3) torus knot, radius modulated by noise; nu : nv = 16 : 1
    const int budget = targetTriangles - 2;
    int nv = std::max( 3, static_cast< int >( std::sqrt( budget / 32.0 ) ) );
    int nu = std::max( 3, budget / ( 2 * nv ) );
    const uint32_t ns = Hash( seed );
    auto centre = []( float t ) {
        float a = t * 6.2831853f;
        float r = 2.0f + std::cos( 3 * a );
        return vec3( r * std::cos( 2 * a ), r * std::sin( 2 * a ), -std::sin( 3 * a ) ) * 0.3f;
    };
    Surface knot = [=]( float u, float v ) {
        vec3 c = centre( u ), ahead = centre( u + 1e-3f );
        vec3 tangent = normalize( ahead - c );
        vec3 n1 = normalize( cross( tangent, vec3( 0.13f, 0.31f, 0.94f ) ) );
        vec3 n2 = cross( tangent, n1 );
        float b = -v * 6.2831853f; // this sense of rotation makes the tube's face normals point outwards
        float r = 0.11f * ( 1.0f + 0.35f * Fbm( u * 96, v * 6, ns ) );
        return c + n1 * ( r * std::cos( b ) ) + n2 * ( r * std::sin( b ) );
    };
    LocalMesh mesh;
    mesh.name     = "knot";
    mesh.material = greenId;
    {
        // closed in both directions: build the u-closed patch, the v seam keeps duplicate vertices
        PatchSpec spec{ knot, static_cast< float >( nu ), static_cast< float >( nv ), true, 0, 8, 1 };
        AppendPatch( mesh, spec, 1.0f );
    }
    long long made = 2ll * nu * nv;
    for ( long long k = made; k < budget; ++k ) // pad to the exact count with small scales on the surface
    {
        float u = ( Hash( seed + 3u * static_cast< uint32_t >( k ) ) >> 8 ) / 16777216.0f;
        float v = ( Hash( seed + 3u * static_cast< uint32_t >( k ) + 1u ) >> 8 ) / 16777216.0f;
        vec3 a = knot( u, v ), b = knot( u + 0.0005f, v ), c = knot( u, v + 0.01f );
        vec3 out = normalize( cross( b - a, c - a ) );
        a += out * 0.004f; b += out * 0.004f; c += out * 0.004f;
        uint32_t base = static_cast< uint32_t >( mesh.vertices.size() );
        mesh.vertices.insert( mesh.vertices.end(), { a, b, c } );
        mesh.normals.insert( mesh.normals.end(), { out, out, out } );
        mesh.uvs.insert( mesh.uvs.end(), { vec2{ 0, 0 }, vec2{ 1, 0 }, vec2{ 0, 1 } } );
        mesh.indices.insert( mesh.indices.end(), { base, base + 1, base + 2 } );
    }
    ComputeUvTangents( mesh );
    SanitizeTangents( mesh );
    scene.AddMeshInstance( mesh, makeTransform( vec3( 0, 0, -3 ), vec3( 0 ), vec3( 1 ) ), -1 );

    LocalMesh plane; // resources/models/plane.obj scaled up, two triangles
    plane.name     = "plane";
    plane.material = floorId;
    plane.vertices = { { -1, 0, 1 }, { 1, 0, 1 }, { 1, 0, -1 }, { -1, 0, -1 } };
    plane.normals  = { { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 0 } };
    plane.uvs      = { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 } };
    plane.indices  = { 0, 1, 2, 0, 2, 3 };
    ComputeUvTangents( plane );
    scene.AddMeshInstance( plane, makeTransform( vec3( 0, -1.1f, -3 ), vec3( 0 ), vec3( 10 ) ), -1 );

    SetCamera( scene, vec3( 0, 0, 0 ), vec3( 0 ), 45, 16.0f / 9.0f, 1.0f, 2.2f );
    scene.splitMethod        = SplitMethod::SAH;
    scene.numSamplesPerPixel = { 1 };
    scene.maxDepth           = 1;
    scene.imageWidth         = 3840;
    scene.imageHeight        = 2160;
    scene.backgroundRadiance = vec3( 0.2f );
    scene.outputImageFilename = "knot.png";
    AddDirectionalLight( scene, vec3( 0, -4, -1 ), vec3( 3, 3, 3 ) );
    scene.BuildAccel();
}

void GenerateSoup( HostScene& scene, uint32_t seed, uint64_t numTriangles, float edge, int skyboxSize )
{
    if ( numTriangles == 0 ) throw std::runtime_error( "GenerateSoup: need at least one triangle" );
    Rng rng( seed );
    std::vector< int > mats( 64 );
    for ( int m = 0; m < 64; ++m )
    {
        PtMaterial pm{};
        pm.albedo[0] = rng.Range( 0.2f, 0.8f );
        pm.albedo[1] = rng.Range( 0.2f, 0.8f );
        pm.albedo[2] = rng.Range( 0.2f, 0.8f );
        pm.albedo_texture = -1;
        mats[m]           = scene.AddMaterial( "soup_" + std::to_string( m ), pm );
    }
    const uint64_t perMesh = 4096;
    const mat4 identity( 1.0f );
    for ( uint64_t start = 0, meshIdx = 0; start < numTriangles; start += perMesh, ++meshIdx )
    {
        uint64_t count = std::min( perMesh, numTriangles - start );
        LocalMesh mesh;
        mesh.name     = "soup";
        mesh.material = mats[meshIdx % 64];
        mesh.vertices.reserve( 3 * count );
        for ( uint64_t t = 0; t < count; ++t )
        {
            vec3 c( rng.Range( -1, 1 ), rng.Range( -1, 1 ), rng.Range( -1, 1 ) );
            vec3 e1( rng.Range( -edge, edge ), rng.Range( -edge, edge ), rng.Range( -edge, edge ) );
            vec3 e2( rng.Range( -edge, edge ), rng.Range( -edge, edge ), rng.Range( -edge, edge ) );
            vec3 a = c, b = c + e1, d = c + e2;
            vec3 n = cross( e1, e2 );
            n      = length( n ) > 0 ? normalize( n ) : vec3( 0, 1, 0 );
            uint32_t base = static_cast< uint32_t >( mesh.vertices.size() );
            mesh.vertices.insert( mesh.vertices.end(), { a, b, d } );
            mesh.normals.insert( mesh.normals.end(), { n, n, n } );
            mesh.uvs.insert( mesh.uvs.end(), { vec2{}, vec2{}, vec2{} } );
            mesh.indices.insert( mesh.indices.end(), { base, base + 1, base + 2 } );
        }
        GenerateFallbackTangents( mesh );
        for ( auto& t : mesh.tangents )
            if ( !( t.x == t.x ) ) t = vec3( 1, 0, 0 );
        scene.AddMeshInstance( mesh, identity, -1 );
    }

    // LDR cubemap: sky gradient with a sun patch and clouds, values in [0,1] like an 8-bit image / 255
    scene.skyboxWidth = scene.skyboxHeight = skyboxSize;
    scene.skybox.resize( 6ull * skyboxSize * skyboxSize * 4 );
    for ( int f = 0; f < 6; ++f )
    {
        for ( int y = 0; y < skyboxSize; ++y )
        {
            for ( int x = 0; x < skyboxSize; ++x )
            {
                float u = ( x + 0.5f ) / skyboxSize, v = ( y + 0.5f ) / skyboxSize;
                float up = f == 2 ? 1.0f : ( f == 3 ? 0.0f : v );
                float cloud = Fbm( u * 6 + f * 3.7f, v * 6, seed + 11u );
                vec3 sky = vec3( 0.35f, 0.55f, 0.9f ) * ( 0.4f + 0.6f * up ) + vec3( 0.5f ) * cloud * up;
                if ( f == 2 && ( u - 0.5f ) * ( u - 0.5f ) + ( v - 0.5f ) * ( v - 0.5f ) < 0.01f ) sky = vec3( 1.0f );
                float* p = &scene.skybox[4 * ( ( static_cast< size_t >( f ) * skyboxSize + y ) * skyboxSize + x )];
                for ( int c = 0; c < 3; ++c )
                    p[c] = static_cast< uint8_t >( 255.0f * std::min( 1.0f, std::max( 0.0f, sky[c] ) ) ) / 255.0f;
                p[3] = 1.0f;
            }
        }
    }

    SetCamera( scene, vec3( 0, 0, 3.2f ), vec3( 0 ), 45, 16.0f / 9.0f, 1.0f, 2.2f );
    scene.splitMethod        = SplitMethod::SAH;
    scene.numSamplesPerPixel = { 4096 };
    scene.maxDepth           = 5;
    scene.imageWidth         = 3840;
    scene.imageHeight        = 2160;
    scene.outputImageFilename = "soup.png";
    scene.BuildAccel();
}

} // namespace pth
