// TEST-ONLY shim (tests/test_cpp_api.py): just enough of <glm/vec3.hpp> for include/chad/tsdf.hpp's glm overloads (the reference's
// tsdf.hpp:68-90) to go through a compiler once -- glm itself is not in this image.  A 12-byte vec3 with x, y, z members, as glm's
// packed default; `packed_vec3_pad` below is a 16-byte variant for the overload's other branch.  Nothing outside tests/ sees this.
#pragma once
namespace glm {
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
}  // namespace glm
