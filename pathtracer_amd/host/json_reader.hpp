// Minimal JSON reader for scene files.
//
// Scene files repeat top-level keys ("Material", "Model", "ModelInstance" appear several times in one
// object) and the reference dispatches members in document order (src/utils/json_parsing.hpp:47-61), so
// an object is kept as an ordered list of (key, value) pairs, duplicates included.
#pragma once

#include <string>
#include <utility>
#include <vector>

namespace pth
{

struct JsonValue
{
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool boolean  = false;
    double number = 0;
    std::string string;
    std::vector< JsonValue > array;
    std::vector< std::pair< std::string, JsonValue > > members;

    bool IsNumber() const { return kind == Number; }
    bool IsString() const { return kind == String; }
    bool IsArray() const { return kind == Array; }
    bool IsObject() const { return kind == Object; }
    const JsonValue* Find( const std::string& key ) const; // first member with that key
};

// Parses `text`; on failure returns false and sets `error` to "line N: message".
bool ParseJson( const std::string& text, JsonValue& out, std::string& error );

bool ParseJsonFile( const std::string& filename, JsonValue& out, std::string& error );

} // namespace pth
