#include "json_reader.hpp"
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace pth
{

const JsonValue* JsonValue::Find( const std::string& key ) const
{
    for ( const auto& m : members )
        if ( m.first == key ) return &m.second;
    return nullptr;
}

namespace
{

    struct Parser
    {
        const std::string& s;
        size_t pos = 0;
        std::string err;

        bool Fail( const std::string& msg )
        {
            if ( err.empty() )
            {
                int line = 1;
                for ( size_t i = 0; i < pos && i < s.size(); ++i )
                    if ( s[i] == '\n' ) ++line;
                err = "line " + std::to_string( line ) + ": " + msg;
            }
            return false;
        }

        void SkipSpace()
        {
            while ( pos < s.size() )
            {
                char c = s[pos];
                if ( c == ' ' || c == '\t' || c == '\n' || c == '\r' )
                    ++pos;
                else if ( c == '/' && pos + 1 < s.size() && s[pos + 1] == '/' ) // tolerate line comments
                    while ( pos < s.size() && s[pos] != '\n' ) ++pos;
                else
                    break;
            }
        }

        bool ParseString( std::string& out )
        {
            if ( s[pos] != '"' ) return Fail( "expected string" );
            ++pos;
            out.clear();
            while ( pos < s.size() && s[pos] != '"' )
            {
                char c = s[pos++];
                if ( c == '\\' )
                {
                    if ( pos >= s.size() ) return Fail( "unterminated escape" );
                    char e = s[pos++];
                    switch ( e )
                    {
                        case 'n': out += '\n'; break;
                        case 't': out += '\t'; break;
                        case 'r': out += '\r'; break;
                        case 'b': out += '\b'; break;
                        case 'f': out += '\f'; break;
                        case 'u':
                        {
                            if ( pos + 4 > s.size() ) return Fail( "bad \\u escape" );
                            unsigned code = std::strtoul( s.substr( pos, 4 ).c_str(), nullptr, 16 );
                            pos += 4;
                            out += code < 0x80 ? static_cast< char >( code ) : '?';
                            break;
                        }
                        default: out += e; break;
                    }
                }
                else
                    out += c;
            }
            if ( pos >= s.size() ) return Fail( "unterminated string" );
            ++pos;
            return true;
        }

        bool ParseValue( JsonValue& v, int depth )
        {
            if ( depth > 64 ) return Fail( "nesting too deep" );
            SkipSpace();
            if ( pos >= s.size() ) return Fail( "unexpected end of input" );
            char c = s[pos];
            if ( c == '{' )
            {
                v.kind = JsonValue::Object;
                ++pos;
                SkipSpace();
                if ( pos < s.size() && s[pos] == '}' ) { ++pos; return true; }
                while ( true )
                {
                    SkipSpace();
                    if ( pos >= s.size() ) return Fail( "unterminated object" );
                    std::string key;
                    if ( !ParseString( key ) ) return false;
                    SkipSpace();
                    if ( pos >= s.size() || s[pos] != ':' ) return Fail( "expected ':'" );
                    ++pos;
                    JsonValue member;
                    if ( !ParseValue( member, depth + 1 ) ) return false;
                    v.members.emplace_back( std::move( key ), std::move( member ) );
                    SkipSpace();
                    if ( pos < s.size() && s[pos] == ',' ) { ++pos; continue; }
                    if ( pos < s.size() && s[pos] == '}' ) { ++pos; return true; }
                    return Fail( "expected ',' or '}'" );
                }
            }
            if ( c == '[' )
            {
                v.kind = JsonValue::Array;
                ++pos;
                SkipSpace();
                if ( pos < s.size() && s[pos] == ']' ) { ++pos; return true; }
                while ( true )
                {
                    JsonValue e;
                    if ( !ParseValue( e, depth + 1 ) ) return false;
                    v.array.push_back( std::move( e ) );
                    SkipSpace();
                    if ( pos < s.size() && s[pos] == ',' ) { ++pos; continue; }
                    if ( pos < s.size() && s[pos] == ']' ) { ++pos; return true; }
                    return Fail( "expected ',' or ']'" );
                }
            }
            if ( c == '"' )
            {
                v.kind = JsonValue::String;
                return ParseString( v.string );
            }
            if ( s.compare( pos, 4, "true" ) == 0 ) { v.kind = JsonValue::Bool; v.boolean = true; pos += 4; return true; }
            if ( s.compare( pos, 5, "false" ) == 0 ) { v.kind = JsonValue::Bool; v.boolean = false; pos += 5; return true; }
            if ( s.compare( pos, 4, "null" ) == 0 ) { v.kind = JsonValue::Null; pos += 4; return true; }

            const char* begin = s.c_str() + pos;
            char* end         = nullptr;
            double d          = std::strtod( begin, &end );
            if ( end == begin ) return Fail( std::string( "unexpected character '" ) + c + "'" );
            v.kind   = JsonValue::Number;
            v.number = d;
            pos += static_cast< size_t >( end - begin );
            return true;
        }
    };

} // namespace

bool ParseJson( const std::string& text, JsonValue& out, std::string& error )
{
    Parser p{ text };
    out = JsonValue();
    if ( !p.ParseValue( out, 0 ) )
    {
        error = p.err;
        return false;
    }
    p.SkipSpace();
    if ( p.pos != text.size() )
    {
        p.Fail( "trailing characters after the document" );
        error = p.err;
        return false;
    }
    return true;
}

bool ParseJsonFile( const std::string& filename, JsonValue& out, std::string& error )
{
    std::ifstream in( filename, std::ios::binary );
    if ( !in )
    {
        error = "could not open '" + filename + "'";
        return false;
    }
    std::stringstream ss;
    ss << in.rdbuf();
    return ParseJson( ss.str(), out, error );
}

} // namespace pth
