// TEST DOUBLE, not glm: the four vector types include/gscuda_shim.hpp aliases when <glm/glm.hpp> exists, with glm's
// default (tightly packed) layout. tests/test_capi_cpu.py puts this directory on the include path to compile the
// shim's glm branch in an image that has no glm; nothing else may include it.
#pragma once
#include <cstdint>
namespace glm {
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct uvec2 { uint32_t x, y; };
}  // namespace glm
