// Small fp32 vector/matrix types for the host side of the render step.
// The reference uses GLM (an external dependency that is not part of this
// build); these types offer the subset its scene and camera code needs, with
// the scalar evaluation order GLM documents, so that values computed on the
// host (triangle normals, camera matrices) carry the reference's rounding.
#pragma once

#include <cmath>

namespace ptm {

struct vec2 {
    float x = 0, y = 0;
    vec2() = default;
    template <class A, class B>
    vec2(A x_, B y_) : x(static_cast<float>(x_)), y(static_cast<float>(y_)) {}
};
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    template <class A, class B, class C>
    vec3(A x_, B y_, C z_)
        : x(static_cast<float>(x_)), y(static_cast<float>(y_)), z(static_cast<float>(z_)) {}
    vec3 &operator+=(vec3 o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3 &operator*=(vec3 o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
    template <class S>
    vec3 &operator*=(S s) { float f = static_cast<float>(s); x *= f; y *= f; z *= f; return *this; }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

// (a.x*b.x + a.y*b.y) + a.z*b.z
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3 cross(vec3 p, vec3 q)
{
    return {p.y * q.z - q.y * p.z, p.z * q.x - q.z * p.x, p.x * q.y - q.x * p.y};
}
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float clamp(float v, float lo, float hi) { return std::fmin(std::fmax(v, lo), hi); }

// Column-major 3x3: col[c] is column c, col[c][r] the element in row r.
struct mat3 {
    vec3 col[3];
    mat3() : col{vec3(1, 0, 0), vec3(0, 1, 0), vec3(0, 0, 1)} {}
    mat3(vec3 c0, vec3 c1, vec3 c2) : col{c0, c1, c2} {}
    // nine scalars fill column 0 first
    template <class... S>
    explicit mat3(S... s)
    {
        static_assert(sizeof...(S) == 9, "mat3 takes three columns or nine scalars");
        const float f[9] = {static_cast<float>(s)...};
        col[0] = vec3(f[0], f[1], f[2]);
        col[1] = vec3(f[3], f[4], f[5]);
        col[2] = vec3(f[6], f[7], f[8]);
    }
    const vec3 &operator[](int c) const { return col[c]; }
};

// row r = (m[0][r]*v.x + m[1][r]*v.y) + m[2][r]*v.z
inline vec3 operator*(const mat3 &m, vec3 v)
{
    return {(m.col[0].x * v.x + m.col[1].x * v.y) + m.col[2].x * v.z,
            (m.col[0].y * v.x + m.col[1].y * v.y) + m.col[2].y * v.z,
            (m.col[0].z * v.x + m.col[1].z * v.y) + m.col[2].z * v.z};
}
// column j = a[0]*b[j].x + a[1]*b[j].y + a[2]*b[j].z
inline mat3 operator*(const mat3 &a, const mat3 &b)
{
    mat3 r;
    for (int j = 0; j < 3; j++)
        r.col[j] = (a.col[0] * b.col[j].x + a.col[1] * b.col[j].y) + a.col[2] * b.col[j].z;
    return r;
}
inline float determinant(const mat3 &m)
{
    return + m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
           - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
           + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}

}  // namespace ptm
