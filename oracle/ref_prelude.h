/* Force-included ahead of every reference translation unit by Makefile.ref (and nothing else).
 * The reference defines M_PI as a float literal only when <cmath> has not defined it (src/math.hpp:9-11),
 * which is the case with the compiler its pictures were made with; glibc defines a double M_PI, with which
 * src/resource/material.cpp:17 (vec3 / double) does not compile.  Including <cmath> first and then
 * redefining M_PI gives every reference file the float constant its own fallback would have given it. */
#include <cmath>
#undef M_PI
#define M_PI 3.14159265358979323846f
