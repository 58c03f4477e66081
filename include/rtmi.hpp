// rtmi.hpp — C++17 host front end above the C ABI (include/rtmi.h).
//
// Mirrors the reference's scene API by name so that its example programs and tests transliterate
// 1:1 (the reference is Rust; no Rust toolchain exists in this build environment, so the host side
// is C++):
//   tuple.rs   point / vector / Point / Vector            color.rs   color / Color
//   matrix.rs  Matrix, translation, scaling, rotation_*, shearing, view_transform
//   pattern.rs stripe_pattern, gradient_pattern, ring_pattern, checkers_pattern
//   materials.rs Phong (builder methods)                  lights.rs  PointLight, AmbientLight, ...
//   shapes/mod.rs Shape, Group, BoundingGroup, CsgPair, SceneItem, build_bounding_tree
//   world.rs   World (add_light, add_item, finalize_scene) camera.rs  Camera (render)
//   obj_loader.rs ObjParser                               canvas.rs  Canvas (write_ppm)
// Everything here is host-side scene preparation in f64; the per-ray
// work happens behind rtmi_render() in hand-written HIP.  Camera::render is the drop-in for
// rust/src/camera.rs:155-161.
#ifndef RTMI_HPP
#define RTMI_HPP

#include "rtmi.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

namespace rtmi {

constexpr double EPSILON = RTMI_EPSILON; // approx_eq.rs:13
constexpr double PI = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------
// tuple.rs
// ---------------------------------------------------------------------------------------------
struct Vector {
    double x = 0, y = 0, z = 0;
    double dot(const Vector &o) const { return x * o.x + y * o.y + z * o.z; }
    Vector cross(const Vector &o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double square_len() const { return dot(*this); }
    double len() const { return std::sqrt(square_len()); }
    Vector normalized() const { double s = 1.0 / len(); return {x * s, y * s, z * s}; }
    Vector reflect(const Vector &n) const { double k = 2.0 * dot(n); return {x - n.x * k, y - n.y * k, z - n.z * k}; }
    double get(int axis) const { return axis == 0 ? x : axis == 1 ? y : z; }
};
struct Point {
    double x = 0, y = 0, z = 0;
    double get(int axis) const { return axis == 0 ? x : axis == 1 ? y : z; }
};
inline Point point(double x, double y, double z) { return {x, y, z}; }
inline Vector vector(double x, double y, double z) { return {x, y, z}; }
inline Vector operator-(const Point &a, const Point &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Point operator+(const Point &a, const Vector &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Point operator-(const Point &a, const Vector &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector operator+(const Vector &a, const Vector &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector operator-(const Vector &a, const Vector &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector operator-(const Vector &a) { return {-a.x, -a.y, -a.z}; }
inline Vector operator*(const Vector &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector operator*(double s, const Vector &a) { return a * s; }
inline Vector operator/(const Vector &a, double s) { return {a.x / s, a.y / s, a.z / s}; }

// ---------------------------------------------------------------------------------------------
// color.rs
// ---------------------------------------------------------------------------------------------
struct Color {
    double red = 0, green = 0, blue = 0;
    Color clip(double lo, double hi) const { // color.rs:62-68
        return {std::fmin(std::fmax(red, lo), hi), std::fmin(std::fmax(green, lo), hi), std::fmin(std::fmax(blue, lo), hi)};
    }
    void to_u8(uint8_t out[3]) const { // color.rs:53-60 (f64::round: half away from zero)
        Color c = clip(0.0, 1.0);
        out[0] = (uint8_t)std::round(c.red * 255.0);
        out[1] = (uint8_t)std::round(c.green * 255.0);
        out[2] = (uint8_t)std::round(c.blue * 255.0);
    }
    double sum() const { return red + green + blue; }
    static Color from_hsv(double hue, double saturation, double value) { // color.rs:23-39
        hue = std::fmod(hue, 360.0);
        int h = (int)std::floor(hue / 60.0);
        double f = hue / 60.0 - h;
        double p = value * (1.0 - saturation);
        double q = value * (1.0 - saturation * f);
        double t = value * (1.0 - saturation * (1.0 - f));
        switch (h) {
        case 1: return {q, value, p};
        case 2: return {p, value, t};
        case 3: return {p, q, value};
        case 4: return {t, p, value};
        case 5: return {value, p, q};
        default: return {value, t, p};
        }
    }
};
inline Color color(double r, double g, double b) { return {r, g, b}; }
inline Color operator+(const Color &a, const Color &b) { return {a.red + b.red, a.green + b.green, a.blue + b.blue}; }
inline Color operator-(const Color &a, const Color &b) { return {a.red - b.red, a.green - b.green, a.blue - b.blue}; }
inline Color operator*(const Color &a, const Color &b) { return {a.red * b.red, a.green * b.green, a.blue * b.blue}; }
inline Color operator*(const Color &a, double s) { return {a.red * s, a.green * s, a.blue * s}; }
inline Color operator*(double s, const Color &a) { return a * s; }
inline Color operator/(const Color &a, double s) { return a * (1.0 / s); } // color.rs:145-150
const Color BLACK{0, 0, 0};
const Color WHITE{1, 1, 1};

// ---------------------------------------------------------------------------------------------
// matrix.rs — the reference's lazy enum (Identity/Full/Translate/Scale/Rotate) collapses to a
// dense 4x4 here; only the products, inverse and into_flat reach the device.
// ---------------------------------------------------------------------------------------------
struct Matrix {
    double m[4][4];
    Matrix() { *this = identity(); }
    explicit Matrix(const double (&v)[16]) { std::memcpy(m, v, sizeof m); }
    static Matrix identity() {
        Matrix r(0);
        for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
        return r;
    }
    static Matrix from_rows(std::initializer_list<double> v) {
        if (v.size() != 16) throw std::invalid_argument("Matrix::from_rows needs 16 values");
        Matrix r(0);
        int k = 0;
        for (double x : v) { r.m[k / 4][k % 4] = x; ++k; }
        return r;
    }
    double operator()(int i, int j) const { return m[i][j]; }
    Matrix operator*(const Matrix &b) const { // matrix.rs:252-309
        Matrix r(0);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += m[i][k] * b.m[k][j];
                r.m[i][j] = s;
            }
        return r;
    }
    Point operator*(const Point &p) const { // matrix.rs:311-327 (w = 1)
        return {m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3], m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
                m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]};
    }
    Vector operator*(const Vector &v) const { // matrix.rs:329-345 (w = 0)
        return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
    }
    Matrix transpose() const { // matrix.rs:109-123
        Matrix r(0);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[i][j] = m[j][i];
        return r;
    }
    double determinant() const {
        double det = 0;
        for (int j = 0; j < 4; ++j) det += m[0][j] * cofactor(0, j);
        return det;
    }
    bool is_invertible() const { return determinant() != 0.0; } // matrix.rs:144-153
    Matrix inverse() const { // matrix.rs:155-164 (vecmath::mat4_inv: adjugate / determinant)
        double det = determinant();
        Matrix r(0);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[j][i] = cofactor(i, j) / det;
        return r;
    }
    void into_flat(double out[16]) const { std::memcpy(out, m, sizeof m); } // matrix.rs:166-187
    bool is_identity() const { return approx_eq(identity()); }
    bool approx_eq(const Matrix &o) const {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (std::fabs(m[i][j] - o.m[i][j]) >= EPSILON) return false;
        return true;
    }
    // chaining builders (matrix.rs:84-94): M.translate(..) == translation(..) * M
    Matrix translate(double dx, double dy, double dz) const;
    Matrix scale(double sx, double sy, double sz) const;
    Matrix rotate(double phi, const Vector &axis) const;
    static Matrix view(const Point &from, const Point &to, const Vector &up); // matrix.rs:96-107

  private:
    explicit Matrix(int) { std::memset(m, 0, sizeof m); }
    double minor3(int r, int c) const {
        double s[3][3];
        int ii = 0;
        for (int i = 0; i < 4; ++i) {
            if (i == r) continue;
            int jj = 0;
            for (int j = 0; j < 4; ++j) {
                if (j == c) continue;
                s[ii][jj++] = m[i][j];
            }
            ++ii;
        }
        return s[0][0] * (s[1][1] * s[2][2] - s[1][2] * s[2][1]) - s[0][1] * (s[1][0] * s[2][2] - s[1][2] * s[2][0]) +
               s[0][2] * (s[1][0] * s[2][1] - s[1][1] * s[2][0]);
    }
    double cofactor(int r, int c) const { return ((r + c) & 1) ? -minor3(r, c) : minor3(r, c); }
};

inline Matrix translation(double dx, double dy, double dz) {
    Matrix r;
    r.m[0][3] = dx; r.m[1][3] = dy; r.m[2][3] = dz;
    return r;
}
inline Matrix scaling(double sx, double sy, double sz) {
    Matrix r;
    r.m[0][0] = sx; r.m[1][1] = sy; r.m[2][2] = sz;
    return r;
}
// matrix.rs:45-47,92-94,190-230: quaternion::axis_angle(axis, phi) -> rotation matrix
inline Matrix rotation(double phi, const Vector &axis) {
    double r = std::cos(phi / 2.0), s = std::sin(phi / 2.0);
    double i = axis.x * s, j = axis.y * s, k = axis.z * s;
    Matrix q;
    q.m[0][0] = 1.0 - 2.0 * (j * j + k * k); q.m[0][1] = 2.0 * (i * j - k * r); q.m[0][2] = 2.0 * (i * k + j * r);
    q.m[1][0] = 2.0 * (i * j + k * r); q.m[1][1] = 1.0 - 2.0 * (i * i + k * k); q.m[1][2] = 2.0 * (j * k - i * r);
    q.m[2][0] = 2.0 * (i * k - j * r); q.m[2][1] = 2.0 * (j * k + i * r); q.m[2][2] = 1.0 - 2.0 * (i * i + j * j);
    return q;
}
inline Matrix rotation_x(double phi) { return rotation(phi, {1, 0, 0}); }
inline Matrix rotation_y(double phi) { return rotation(phi, {0, 1, 0}); }
inline Matrix rotation_z(double phi) { return rotation(phi, {0, 0, 1}); }
inline Matrix shearing(double xy, double xz, double yx, double yz, double zx, double zy) { // matrix.rs:53-67
    return Matrix::from_rows({1, xy, xz, 0, yx, 1, yz, 0, zx, zy, 1, 0, 0, 0, 0, 1});
}
inline Matrix Matrix::translate(double dx, double dy, double dz) const { return translation(dx, dy, dz) * *this; }
inline Matrix Matrix::scale(double sx, double sy, double sz) const { return scaling(sx, sy, sz) * *this; }
inline Matrix Matrix::rotate(double phi, const Vector &axis) const { return rotation(phi, axis) * *this; }
inline Matrix Matrix::view(const Point &from, const Point &to, const Vector &up) {
    Vector forward = (to - from).normalized();
    Vector left = forward.cross(up.normalized());
    Vector true_up = left.cross(forward);
    Matrix orientation = Matrix::from_rows({left.x, left.y, left.z, 0, true_up.x, true_up.y, true_up.z, 0, -forward.x, -forward.y,
                                            -forward.z, 0, 0, 0, 0, 1});
    return orientation * translation(-from.x, -from.y, -from.z);
}
inline Matrix view_transform(const Point &from, const Point &to, const Vector &up) { return Matrix::view(from, to, up); }

// ---------------------------------------------------------------------------------------------
// aabb.rs (host part: construction, merge, transform; the slab test runs on the device)
// ---------------------------------------------------------------------------------------------
struct Aabb {
    Point min_p{INFINITY, INFINITY, INFINITY}, max_p{-INFINITY, -INFINITY, -INFINITY}; // Aabb::empty, aabb.rs:26-31
    static Aabb make(double xmin, double xmax, double ymin, double ymax, double zmin, double zmax) {
        Aabb a; a.min_p = {xmin, ymin, zmin}; a.max_p = {xmax, ymax, zmax}; return a;
    }
    static Aabb empty() { return Aabb(); }
    static Aabb empty_at(const Point &p) { Aabb a; a.min_p = p; a.max_p = p; return a; }
    Point center() const { return min_p + (max_p - min_p) / 2.0; } // aabb.rs:37-39
    Aabb merge(const Aabb &o) const {                              // aabb.rs:72-81 (f64::min/max ignore NaN)
        return make(std::fmin(min_p.x, o.min_p.x), std::fmax(max_p.x, o.max_p.x), std::fmin(min_p.y, o.min_p.y),
                    std::fmax(max_p.y, o.max_p.y), std::fmin(min_p.z, o.min_p.z), std::fmax(max_p.z, o.max_p.z));
    }
    Aabb extend(const Point &p) const { // aabb.rs:83-92
        return make(std::fmin(min_p.x, p.x), std::fmax(max_p.x, p.x), std::fmin(min_p.y, p.y), std::fmax(max_p.y, p.y),
                    std::fmin(min_p.z, p.z), std::fmax(max_p.z, p.z));
    }
    Aabb transform(const Matrix &t) const { // aabb.rs:94-103
        return empty_at(t * min_p)
            .extend(t * Point{min_p.x, min_p.y, max_p.z})
            .extend(t * Point{min_p.x, max_p.y, min_p.z})
            .extend(t * Point{min_p.x, max_p.y, max_p.z})
            .extend(t * Point{max_p.x, min_p.y, min_p.z})
            .extend(t * Point{max_p.x, min_p.y, max_p.z})
            .extend(t * Point{max_p.x, max_p.y, min_p.z})
            .extend(t * max_p);
    }
};

// ---------------------------------------------------------------------------------------------
// pattern.rs — closures become an enum: only the four built-in patterns (plus the tests'
// xyz pattern) can cross a C ABI.
// ---------------------------------------------------------------------------------------------
struct Pattern {
    int kind = RTMI_PATTERN_STRIPE;
    Color a, b;
    Matrix inv_transform_;
    void set_transform(const Matrix &t) { inv_transform_ = t.inverse(); } // pattern.rs:55-57
    Pattern with_transform(const Matrix &t) const { Pattern p = *this; p.set_transform(t); return p; }
    const Matrix &inv_transform() const { return inv_transform_; }
};
inline Pattern make_pattern(int kind, const Color &a, const Color &b) { Pattern p; p.kind = kind; p.a = a; p.b = b; return p; }
inline Pattern stripe_pattern(const Color &a, const Color &b) { return make_pattern(RTMI_PATTERN_STRIPE, a, b); }
inline Pattern gradient_pattern(const Color &a, const Color &b) { return make_pattern(RTMI_PATTERN_GRADIENT, a, b); }
inline Pattern ring_pattern(const Color &a, const Color &b) { return make_pattern(RTMI_PATTERN_RING, a, b); }
inline Pattern checkers_pattern(const Color &a, const Color &b) { return make_pattern(RTMI_PATTERN_CHECKERS, a, b); }
inline Pattern test_pattern() { return make_pattern(RTMI_PATTERN_XYZ, BLACK, BLACK); } // pattern.rs:90-92

// ---------------------------------------------------------------------------------------------
// materials.rs — Phong (53-237)
// ---------------------------------------------------------------------------------------------
struct Phong {
    bool has_pattern = false;
    Color color_{1, 1, 1};
    Pattern pattern_;
    double emissive_ = 0.0, diffuse_ = 0.9, specular_ = 0.9, shininess_ = 200.0, reflective_ = 0.0, transparency_ = 0.0,
           refractive_index_ = 1.0; // materials.rs:65-69

    Phong() = default;
    Phong(const Color &c, double emissive, double diffuse, double specular, double shininess, double reflective,
          double transparency, double refractive_index) // Phong::new, materials.rs:72-92
        : color_(c), emissive_(emissive), diffuse_(diffuse), specular_(specular), shininess_(shininess),
          reflective_(reflective), transparency_(transparency), refractive_index_(refractive_index) {}
    static Phong new_pattern(const Pattern &p, double emissive, double diffuse, double specular, double shininess,
                             double reflective, double transparency, double refractive_index) {
        Phong m(BLACK, emissive, diffuse, specular, shininess, reflective, transparency, refractive_index);
        m.has_pattern = true; m.pattern_ = p;
        return m;
    }
    Phong with_color(const Color &c) const { Phong m = *this; m.has_pattern = false; m.color_ = c; return m; }
    Phong with_rgb(double r, double g, double b) const { return with_color({r, g, b}); }
    Phong with_hsv(double h, double s, double v) const { return with_color(Color::from_hsv(h, s, v)); }
    Phong with_pattern(const Pattern &p) const { Phong m = *this; m.has_pattern = true; m.pattern_ = p; return m; }
    Phong with_emissive(double v) const { Phong m = *this; m.emissive_ = v; return m; }
    Phong with_diffuse(double v) const { Phong m = *this; m.diffuse_ = v; return m; }
    Phong with_specular(double v) const { Phong m = *this; m.specular_ = v; return m; }
    Phong with_shininess(double v) const { Phong m = *this; m.shininess_ = v; return m; }
    Phong with_reflective(double v) const { Phong m = *this; m.reflective_ = v; return m; }
    Phong with_transparency(double v) const { Phong m = *this; m.transparency_ = v; return m; }
    Phong with_refractive_index(double v) const { Phong m = *this; m.refractive_index_ = v; return m; }
    void set_color(const Color &c) { has_pattern = false; color_ = c; }
    void set_pattern(const Pattern &p) { has_pattern = true; pattern_ = p; }
    void set_emissive(double v) { emissive_ = v; }
    void set_diffuse(double v) { diffuse_ = v; }
    void set_specular(double v) { specular_ = v; }
    void set_shininess(double v) { shininess_ = v; }
    void set_reflective(double v) { reflective_ = v; }
    void set_transparency(double v) { transparency_ = v; }
    void set_refractive_index(double v) { refractive_index_ = v; }
    double emissive() const { return emissive_; }
    double diffuse() const { return diffuse_; }
    double specular() const { return specular_; }
    double shininess() const { return shininess_; }
    double reflective() const { return reflective_; }
    double transparency() const { return transparency_; }
    double refractive_index() const { return refractive_index_; }
    const Color &color() const { return color_; }
};

// ---------------------------------------------------------------------------------------------
// lights.rs — constructors only; incoming_at runs on the device
// ---------------------------------------------------------------------------------------------
struct Light {
    rtmi_light l;
    Light() { std::memset(&l, 0, sizeof l); }
};
inline void set3(double *d, double x, double y, double z) { d[0] = x; d[1] = y; d[2] = z; }
inline Light PointLight(const Point &position, const Color &intensity) { // lights.rs:52-58
    Light L; L.l.kind = RTMI_LIGHT_POINT;
    set3(L.l.position, position.x, position.y, position.z); set3(L.l.intensity, intensity.red, intensity.green, intensity.blue);
    return L;
}
inline Light RealisticPointLight(const Point &position, const Color &intensity) { // lights.rs:120-126
    Light L = PointLight(position, intensity); L.l.kind = RTMI_LIGHT_REALISTIC_POINT; return L;
}
inline Light AmbientLight(const Color &intensity) { // lights.rs:269-272
    Light L; L.l.kind = RTMI_LIGHT_AMBIENT; set3(L.l.intensity, intensity.red, intensity.green, intensity.blue); return L;
}
inline Light SphereLight(const Point &position, double radius, const Color &intensity) { // lights.rs:314-321
    Light L = PointLight(position, intensity); L.l.kind = RTMI_LIGHT_SPHERE; L.l.radius = radius; return L;
}
inline Light DiscLight(const Point &position, const Vector &normal, double radius, const Color &intensity) { // lights.rs:393-409
    Light L = PointLight(position, intensity); L.l.kind = RTMI_LIGHT_DISC; L.l.radius = radius;
    Vector helper{1, 0, 0};
    if (std::fabs(helper.dot(normal) - 1.0) < EPSILON) helper = {0, 1, 0};
    Vector width = normal.cross(helper);
    Vector height = width.cross(normal);
    width = width * radius; height = height * radius;
    set3(L.l.normal, normal.x, normal.y, normal.z);
    set3(L.l.width, width.x, width.y, width.z);
    set3(L.l.height, height.x, height.y, height.z);
    return L;
}
inline Light Beam(const Point &position, const Vector &up, const Vector &right, const Color &intensity) { // lights.rs:188-201
    if (std::fabs(up.dot(right)) >= EPSILON) throw std::invalid_argument("*up* and *right* must be orthogonal");
    Light L = PointLight(position, intensity); L.l.kind = RTMI_LIGHT_BEAM;
    Vector direction = right.cross(up).normalized();
    set3(L.l.up, up.x, up.y, up.z); set3(L.l.right, right.x, right.y, right.z);
    set3(L.l.direction, direction.x, direction.y, direction.z);
    return L;
}

// ---------------------------------------------------------------------------------------------
// shapes/mod.rs — Shape, Group, BoundingGroup, CsgPair, SceneItem
// ---------------------------------------------------------------------------------------------
struct Shape {
    int kind = RTMI_SPHERE;
    double minimum = -INFINITY, maximum = INFINITY; // cylinder.rs:21-25
    bool closed = false;
    Point p1, p2, p3;
    Vector n1, n2, n3;
    Phong material_;
    Matrix transform_, cumulative_transform_, inv_cumulative_transform_;
    bool cast_shadow_ = true;

    Shape with_material(const Phong &m) const { Shape s = *this; s.material_ = m; return s; }
    Shape with_transform(const Matrix &t) const { Shape s = *this; s.set_transform(t); return s; } // mod.rs:278-285
    Shape with_cast_shadow(bool b) const { Shape s = *this; s.cast_shadow_ = b; return s; }
    void set_material(const Phong &m) { material_ = m; }
    void set_transform(const Matrix &t) { transform_ = t; cumulative_transform_ = t; inv_cumulative_transform_ = t.inverse(); }
    void set_cast_shadow(bool b) { cast_shadow_ = b; }
    void update_transform(const Matrix &t) { // mod.rs:309-313
        cumulative_transform_ = t * transform_;
        inv_cumulative_transform_ = cumulative_transform_.inverse();
    }
    const Phong &material() const { return material_; }
    Phong &material_mut() { return material_; }
    const Matrix &inv_transform() const { return inv_cumulative_transform_; }
    bool cast_shadow() const { return cast_shadow_; }
    Point world_to_object(const Point &p) const { return inv_cumulative_transform_ * p; } // mod.rs:315-317
    Vector normal_to_world(const Vector &n) const { return (inv_cumulative_transform_.transpose() * n).normalized(); } // mod.rs:319-322
    Aabb local_aabb() const { // Geometry::aabb per kind
        switch (kind) {
        case RTMI_SPHERE: case RTMI_CUBE: return Aabb::make(-1, 1, -1, 1, -1, 1);                       // sphere.rs:69, cube.rs:69
        case RTMI_PLANE: return Aabb::make(-INFINITY, INFINITY, -EPSILON, EPSILON, -INFINITY, INFINITY); // plane.rs:50-52
        case RTMI_CYLINDER: return Aabb::make(-1, 1, minimum, maximum, -1, 1);                           // cylinder.rs:125-127
        case RTMI_CONE: { double r = std::fmax(std::fabs(minimum), std::fabs(maximum)); return Aabb::make(-r, r, minimum, maximum, -r, r); } // cone.rs:144-147
        default: // triangle.rs:73-82
            return Aabb::make(std::fmin(std::fmin(p1.x, p2.x), p3.x), std::fmax(std::fmax(p1.x, p2.x), p3.x),
                              std::fmin(std::fmin(p1.y, p2.y), p3.y), std::fmax(std::fmax(p1.y, p2.y), p3.y),
                              std::fmin(std::fmin(p1.z, p2.z), p3.z), std::fmax(std::fmax(p1.z, p2.z), p3.z));
        }
    }
    Aabb aabb() const { return local_aabb().transform(cumulative_transform_); } // mod.rs:345-347
};

inline Shape sphere() { Shape s; s.kind = RTMI_SPHERE; return s; }
inline Shape glass_sphere() { // sphere.rs:13-23
    return sphere().with_material(Phong().with_rgb(1, 1, 1).with_diffuse(0.1).with_specular(0.9).with_shininess(250.0).with_transparency(1.0).with_refractive_index(1.5));
}
inline Shape plane() { Shape s; s.kind = RTMI_PLANE; return s; }
inline Shape cube() { Shape s; s.kind = RTMI_CUBE; return s; }
inline Shape cylinder(double minimum = -INFINITY, double maximum = INFINITY, bool closed = false) {
    Shape s; s.kind = RTMI_CYLINDER; s.minimum = minimum; s.maximum = maximum; s.closed = closed; return s;
}
inline Shape cone(double minimum = -INFINITY, double maximum = INFINITY, bool closed = false) {
    Shape s; s.kind = RTMI_CONE; s.minimum = minimum; s.maximum = maximum; s.closed = closed; return s;
}
inline Shape triangle(const Point &p1, const Point &p2, const Point &p3) {
    Shape s; s.kind = RTMI_TRIANGLE; s.p1 = p1; s.p2 = p2; s.p3 = p3; return s;
}
inline Shape smooth_triangle(const Point &p1, const Point &p2, const Point &p3, const Vector &n1, const Vector &n2, const Vector &n3) {
    Shape s; s.kind = RTMI_SMOOTH_TRIANGLE; s.p1 = p1; s.p2 = p2; s.p3 = p3; s.n1 = n1; s.n2 = n2; s.n3 = n3; return s;
}

struct SceneItem;

struct Group { // mod.rs:350-441
    Matrix transform_, cumulative_transform_, inv_cumulative_transform_;
    bool cast_shadow = true;
    std::vector<SceneItem> items;
    Group with_child(const SceneItem &c) const;
    void add_child(const SceneItem &c);
    const SceneItem &get_child(size_t i) const;
    bool aint_empty() const { return !items.empty(); }
    Group with_transform(const Matrix &t) const { Group g = *this; g.set_transform(t); return g; }
    void set_transform(const Matrix &t) { transform_ = t; cumulative_transform_ = t; inv_cumulative_transform_ = t.inverse(); }
    void update_transform(const Matrix &t);
    Aabb update_aabb();
    Aabb aabb() const;
};

struct BoundingGroup { // mod.rs:443-487
    Group group;
    Aabb aabb_;
    void add_child(const SceneItem &c);
    BoundingGroup with_transform(const Matrix &t) const { BoundingGroup b = *this; b.group = group.with_transform(t); return b; }
    const Aabb &update_aabb();
};

enum class CsgOp { Union, Intersection, Difference };

struct CsgPair { // csg.rs:35-129
    Matrix transform_, cumulative_transform_, inv_cumulative_transform_;
    bool cast_shadow_ = true;
    CsgOp op = CsgOp::Union;
    std::vector<SceneItem> items; // exactly two: (left, right)
    CsgPair with_cast_shadow(bool b) const { CsgPair c = *this; c.cast_shadow_ = b; return c; }
    CsgPair with_transform(const Matrix &t) const { CsgPair c = *this; c.set_transform(t); return c; }
    void set_transform(const Matrix &t) { transform_ = t; cumulative_transform_ = t; inv_cumulative_transform_ = t.inverse(); }
    void update_transform(const Matrix &t);
    Aabb update_aabb();
    Aabb aabb() const;
    bool cast_shadow() const { return cast_shadow_; }
    const SceneItem &left() const;
    const SceneItem &right() const;
};

struct SceneItem { // mod.rs:54-180
    std::variant<Shape, Group, BoundingGroup, CsgPair> v;
    SceneItem() = default;
    SceneItem(const Shape &s) : v(s) {}
    SceneItem(const Group &g) : v(g) {}
    SceneItem(const BoundingGroup &g) : v(g) {}
    SceneItem(const CsgPair &c) : v(c) {}
    const Shape *as_shape() const { return std::get_if<Shape>(&v); }
    Shape *as_shape_mut() { return std::get_if<Shape>(&v); }
    const Group *as_group() const { return std::get_if<Group>(&v); }
    const BoundingGroup *as_bounded() const { return std::get_if<BoundingGroup>(&v); }
    const CsgPair *as_csg() const { return std::get_if<CsgPair>(&v); }
    SceneItem with_transform(const Matrix &t) const { // mod.rs:98-106
        return std::visit([&](const auto &x) { return SceneItem(x.with_transform(t)); }, v);
    }
    bool cast_shadow() const { // mod.rs:117-124
        if (auto s = as_shape()) return s->cast_shadow();
        if (auto g = as_group()) return g->cast_shadow;
        if (auto b = as_bounded()) return b->group.cast_shadow;
        return as_csg()->cast_shadow();
    }
    void update_transform(const Matrix &t) { // mod.rs:126-133
        if (auto s = std::get_if<Shape>(&v)) s->update_transform(t);
        else if (auto g = std::get_if<Group>(&v)) g->update_transform(t);
        else if (auto b = std::get_if<BoundingGroup>(&v)) b->group.update_transform(t);
        else std::get<CsgPair>(v).update_transform(t);
    }
    Aabb update_aabb() { // mod.rs:153-160
        if (auto s = std::get_if<Shape>(&v)) return s->aabb();
        if (auto g = std::get_if<Group>(&v)) return g->update_aabb();
        if (auto b = std::get_if<BoundingGroup>(&v)) return b->update_aabb();
        return std::get<CsgPair>(v).update_aabb();
    }
    Aabb aabb() const { // mod.rs:162-169
        if (auto s = as_shape()) return s->aabb();
        if (auto g = as_group()) return g->aabb();
        if (auto b = as_bounded()) return b->aabb_;
        return as_csg()->aabb();
    }
};

inline Group Group::with_child(const SceneItem &c) const { Group g = *this; g.items.push_back(c); return g; }
inline void Group::add_child(const SceneItem &c) { items.push_back(c); }
inline const SceneItem &Group::get_child(size_t i) const { return items.at(i); }
inline void Group::update_transform(const Matrix &t) { // mod.rs:418-424
    cumulative_transform_ = t * transform_;
    inv_cumulative_transform_ = cumulative_transform_.inverse();
    for (auto &c : items) c.update_transform(cumulative_transform_);
}
inline Aabb Group::update_aabb() { Aabb a; for (auto &c : items) a = a.merge(c.update_aabb()); return a; } // mod.rs:426-432
inline Aabb Group::aabb() const { Aabb a; for (auto &c : items) a = a.merge(c.aabb()); return a; }         // mod.rs:434-440
inline void BoundingGroup::add_child(const SceneItem &c) { group.add_child(c); }
inline const Aabb &BoundingGroup::update_aabb() { // mod.rs:461-469
    Aabb a;
    for (auto &c : group.items) a = a.merge(c.update_aabb());
    aabb_ = a;
    return aabb_;
}
inline void CsgPair::update_transform(const Matrix &t) { // csg.rs:75-80
    cumulative_transform_ = t * transform_;
    inv_cumulative_transform_ = cumulative_transform_.inverse();
    items[0].update_transform(cumulative_transform_);
    items[1].update_transform(cumulative_transform_);
}
inline Aabb CsgPair::update_aabb() { return items[0].update_aabb().merge(items[1].update_aabb()); } // csg.rs:82-87
inline Aabb CsgPair::aabb() const { return items[0].aabb().merge(items[1].aabb()); }               // csg.rs:89-91
inline const SceneItem &CsgPair::left() const { return items[0]; }
inline const SceneItem &CsgPair::right() const { return items[1]; }

inline Group group() { return Group(); }

// TriangleMesh (triangle.rs:164-231): in the reference ONE Shape whose geometry is a list of flat faces - intersect() runs
// intersect_triangle over every face and tags each root with the face's index (Intersection::i), normal_at() asks that
// face.  The C ABI has no leaf kind for it: a mesh FLATTENS into a Group of RTMI_TRIANGLE leaves that share the Shape's
// transform and material (into_group below), and build_bounding_tree may then split it like any Group - the same roots,
// normals, colours and shadows, since a hit is decided triangle by triangle in the reference too.  What differs is object
// IDENTITY, which the reference uses in two places: the n1 / n2 container list of a TRANSPARENT
// mesh (ray.rs:111-140 toggles the one Shape on every face crossed; flattened, each face is a container of its own) and a
// mesh as a CSG operand (csg.rs:102 `contains`: one shape there, one leaf per face here - the filter's in/out state is
// per operand side, so the result is the same).  Opaque meshes, the only kind the reference's examples could build,
// render identically; the `meshes` scene of raytracing_amd/scenes.py is the parity case.
struct TriangleMesh {
    std::vector<Shape> faces;
    static TriangleMesh from_mesh(const std::vector<Point> &vertices, const std::vector<std::vector<size_t>> &triangles) { // triangle.rs:174-180
        TriangleMesh m;
        for (const auto &t : triangles) {
            if (t.size() != 3) throw std::invalid_argument("TriangleMesh::from_mesh: a face is three vertex indices");
            m.faces.push_back(triangle(vertices.at(t[0]), vertices.at(t[1]), vertices.at(t[2])));
        }
        return m;
    }
    bool aint_empty() const { return !faces.empty(); }                 // triangle.rs:182-184
    void add_triangle(const Shape &tri) {                             // triangle.rs:186-188
        if (tri.kind != RTMI_TRIANGLE) throw std::invalid_argument("TriangleMesh::add_triangle: flat triangles only");
        faces.push_back(tri);
    }
    Shape get_triangle(size_t idx) const { return faces.at(idx); }     // triangle.rs:190-192
    // Shape::new(mesh).with_material(m): the faces as a Group, every face with the mesh's material; the mesh's transform is
    // the Group's (with_transform on the result)
    Group into_group(const Phong &material) const {
        Group g;
        for (const Shape &f : faces) g.add_child(SceneItem(f.with_material(material)));
        return g;
    }
};
inline BoundingGroup bounding_group() { return BoundingGroup(); }
inline CsgPair make_csg(CsgOp op, const SceneItem &a, const SceneItem &b) { CsgPair c; c.op = op; c.items = {a, b}; return c; }
inline CsgPair csg_union(const SceneItem &a, const SceneItem &b) { return make_csg(CsgOp::Union, a, b); }
inline CsgPair csg_intersection(const SceneItem &a, const SceneItem &b) { return make_csg(CsgOp::Intersection, a, b); }
inline CsgPair csg_difference(const SceneItem &a, const SceneItem &b) { return make_csg(CsgOp::Difference, a, b); }

// build_bounding_tree + compute_extents (mod.rs:489-563): median split on AABB centres of the
// largest-extent axis; kept split-for-split so that node counts and leaf sizes match the reference.
inline BoundingGroup build_bounding_tree(Group grp, size_t max_leaf_size) {
    size_t n = grp.items.size();
    if (n <= max_leaf_size) { BoundingGroup b; b.group = std::move(grp); return b; }
    grp.update_transform(Matrix::identity());
    std::vector<Aabb> aabbs;
    aabbs.reserve(n);
    for (auto &it : grp.items) aabbs.push_back(it.update_aabb());
    std::vector<double> cx(n), cy(n), cz(n);
    for (size_t i = 0; i < n; ++i) { Point c = aabbs[i].center(); cx[i] = c.x; cy[i] = c.y; cz[i] = c.z; }
    auto sx = cx, sy = cy, sz = cz;
    std::sort(sx.begin(), sx.end()); std::sort(sy.begin(), sy.end()); std::sort(sz.begin(), sz.end());
    double ex = sx[n - 1] - sx[0], ey = sy[n - 1] - sy[0], ez = sz[n - 1] - sz[0];
    double mx = sx[n / 2], my = sy[n / 2], mz = sz[n / 2];
    std::vector<bool> criterion(n);
    if (ex >= ey && ex >= ez) { for (size_t i = 0; i < n; ++i) criterion[i] = cx[i] < mx; }
    else if (ey >= ex && ey >= ez) { for (size_t i = 0; i < n; ++i) criterion[i] = cy[i] < my; }
    else { for (size_t i = 0; i < n; ++i) criterion[i] = cz[i] < mz; }
    size_t n1 = 0;
    for (size_t i = 0; i < n; ++i) n1 += criterion[i] ? 1 : 0;
    if (n1 == 0 || n1 == n) { BoundingGroup b; b.group = std::move(grp); return b; }
    Group g1, g2;
    for (size_t i = 0; i < n; ++i) (criterion[i] ? g1 : g2).items.push_back(std::move(grp.items[i]));
    BoundingGroup result = BoundingGroup().with_transform(grp.transform_);
    result.add_child(build_bounding_tree(std::move(g1), max_leaf_size));
    result.add_child(build_bounding_tree(std::move(g2), max_leaf_size));
    return result;
}

// ---------------------------------------------------------------------------------------------
// obj_loader.rs
// ---------------------------------------------------------------------------------------------
struct ObjParser {
    std::vector<std::string> ignored_;
    std::vector<Point> vertices;
    std::vector<Vector> normals;
    std::map<std::string, Group> groups; // BTreeMap: ordered by name
    std::string current_group = "default_group";
    // What the reference does with a `vt` (or `vp`) line is push it as a vertex - obj_loader.rs:35-36 matches any line
    // that starts with "v" - which shifts every later vertex index; harmless for data/teapot.obj, whose `vt` block comes
    // after its last `v`.  That behaviour is the default here too (drop-in); skip_texture_lines = true reads such files
    // as OBJ means them.
    bool skip_texture_lines = false;

    static ObjParser parse_str(const std::string &input, bool skip_texture_lines = false) { // obj_loader.rs:28-44
        ObjParser data;
        data.skip_texture_lines = skip_texture_lines;
        data.groups["default_group"] = Group();
        std::istringstream in(input);
        std::string line;
        while (std::getline(in, line)) {
            size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
            line = b == std::string::npos ? "" : line.substr(b, e - b + 1);
            if (line.rfind("vn", 0) == 0) data.parse_normal(line);
            else if (skip_texture_lines && (line.rfind("vt", 0) == 0 || line.rfind("vp", 0) == 0)) data.ignored_.push_back(line);
            else if (line.rfind("v", 0) == 0) data.parse_vertex(line); // NB: also catches "vt" like the reference (obj_loader.rs:36)
            else if (line.rfind("f", 0) == 0) data.parse_face(line);
            else if (line.rfind("g", 0) == 0) data.parse_group(line);
            else data.ignored_.push_back(line);
        }
        return data;
    }
    const std::vector<std::string> &ignored() const { return ignored_; }
    const Group &get_group(const std::string &name) const { return groups.at(name); }
    Group into_group() const { // obj_loader.rs:144-161
        if (groups.size() == 1) return groups.begin()->second;
        Group g;
        for (auto &kv : groups)
            if (kv.second.aint_empty()) g.add_child(kv.second);
        return g;
    }

  private:
    static std::vector<std::string> words(const std::string &line) {
        std::istringstream in(line);
        std::vector<std::string> w;
        std::string s;
        while (in >> s) w.push_back(s);
        return w;
    }
    static std::vector<double> floats(const std::string &line) { // skip(1).map(parse).filter_map(ok)
        std::vector<double> v;
        auto w = words(line);
        for (size_t i = 1; i < w.size(); ++i) {
            try { size_t pos = 0; double d = std::stod(w[i], &pos); if (pos == w[i].size()) v.push_back(d); } catch (...) {}
        }
        return v;
    }
    void parse_vertex(const std::string &line) {
        auto v = floats(line);
        if (v.size() < 3) throw std::runtime_error("obj: vertex needs 3 numbers: " + line); // reference panics on v[2]
        vertices.push_back({v[0], v[1], v[2]});
    }
    void parse_normal(const std::string &line) {
        auto v = floats(line);
        if (v.size() < 3) throw std::runtime_error("obj: normal needs 3 numbers: " + line);
        normals.push_back({v[0], v[1], v[2]});
    }
    void parse_group(const std::string &line) { // obj_loader.rs:100-110
        auto w = words(line);
        current_group = w.size() > 1 ? w[1] : "default_group";
        groups.emplace(current_group, Group());
    }
    // An OBJ index: 1-based, or negative = counted back from the element defined last (-1 is the latest).  The reference
    // parses into usize and knows only the first form (a negative index is dropped from a plain face, obj_loader.rs:113,
    // and panics in a v/t/n one, :82); files that use the second form load here.
    static bool parse_index(const std::string &s, size_t count, size_t &out) {
        if (s.empty()) return false;
        try {
            size_t pos = 0;
            const long long k = std::stoll(s, &pos);
            if (pos != s.size()) return false;
            if (k > 0) { out = (size_t)k - 1; return true; }
            if (k < 0 && (size_t)(-k) <= count) { out = count - (size_t)(-k); return true; }
        } catch (...) {}
        return false;
    }
    void parse_face(const std::string &line) { // obj_loader.rs:74-98
        auto el = words(line);
        el.erase(el.begin());
        if (el.empty()) throw std::runtime_error("obj: empty face");
        if (el[0].find('/') != std::string::npos) {
            std::vector<std::pair<size_t, size_t>> idx; // (vertex, normal); normal = npos for v/t elements
            bool with_normals = true;
            for (auto &s : el) {
                const size_t a = s.find('/'), b = a == std::string::npos ? a : s.find('/', a + 1);
                if (a == std::string::npos) throw std::runtime_error("obj: face element needs v/t/n: " + s);
                size_t v = 0, n = std::string::npos;
                if (!parse_index(s.substr(0, a), vertices.size(), v)) throw std::runtime_error("obj: bad vertex index in face element: " + s);
                if (b == std::string::npos || b + 1 >= s.size()) with_normals = false; // "v/t": no normal (the reference panics here)
                else if (!parse_index(s.substr(b + 1), normals.size(), n)) throw std::runtime_error("obj: bad normal index in face element: " + s);
                idx.push_back({v, n});
            }
            for (size_t i = 1; i + 1 < idx.size(); ++i) { // triangulate_smooth_fan, obj_loader.rs:126-141
                if (with_normals)
                    groups[current_group].add_child(smooth_triangle(vertices.at(idx[0].first), vertices.at(idx[i].first), vertices.at(idx[i + 1].first),
                                                                    normals.at(idx[0].second), normals.at(idx[i].second), normals.at(idx[i + 1].second)));
                else
                    groups[current_group].add_child(triangle(vertices.at(idx[0].first), vertices.at(idx[i].first), vertices.at(idx[i + 1].first)));
            }
        } else {
            std::vector<size_t> idx;
            for (auto &s : el) {
                size_t k = 0;
                if (parse_index(s, vertices.size(), k)) idx.push_back(k); // (.filter_map(ok), obj_loader.rs:113)
            }
            for (size_t i = 1; i + 1 < idx.size(); ++i) // triangulate_fan, obj_loader.rs:112-124
                groups[current_group].add_child(triangle(vertices.at(idx[0]), vertices.at(idx[i]), vertices.at(idx[i + 1])));
        }
    }
};

// ---------------------------------------------------------------------------------------------
// canvas.rs
// ---------------------------------------------------------------------------------------------
struct Canvas {
    uint32_t width_ = 0, height_ = 0;
    std::vector<Color> data;
    Canvas() = default;
    Canvas(uint32_t w, uint32_t h) : width_(w), height_(h), data((size_t)w * h) {}
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    void clear(const Color &c) { std::fill(data.begin(), data.end(), c); }
    void set_pixel(uint32_t x, uint32_t y, const Color &c) { data.at((size_t)y * width_ + x) = c; }
    void add_to_pixel(uint32_t x, uint32_t y, const Color &c) { auto &p = data.at((size_t)y * width_ + x); p = p + c; }
    Color get_pixel(uint32_t x, uint32_t y) const { return data.at((size_t)y * width_ + x); }
    Color average_brightness() const { Color s; for (auto &c : data) s = s + c; return s / (double)((size_t)width_ * height_); }
    std::vector<uint8_t> to_rgb8() const { // canvas.rs:101-107
        std::vector<uint8_t> out(data.size() * 3);
        for (size_t i = 0; i < data.size(); ++i) data[i].to_u8(&out[3 * i]);
        return out;
    }
    // write_png (canvas.rs:95-110): 8-bit RGB, every pixel through Color::to_u8.  The reference hands the bytes to the png
    // crate; here the file is written directly: signature, IHDR, one IDAT whose zlib stream holds the scanlines (filter
    // type 0 in front of each) in stored deflate blocks - valid PNG, no compressor needed - and IEND.
    std::vector<uint8_t> to_png() const {
        const std::vector<uint8_t> rgb = to_rgb8();
        std::vector<uint8_t> raw;
        raw.reserve(((size_t)width_ * 3 + 1) * height_);
        for (uint32_t y = 0; y < height_; ++y) {
            raw.push_back(0);
            raw.insert(raw.end(), rgb.begin() + (size_t)y * width_ * 3, rgb.begin() + (size_t)(y + 1) * width_ * 3);
        }
        std::vector<uint8_t> z = {0x78, 0x01}; // zlib header: deflate, 32 K window, no preset dictionary
        uint32_t a = 1, b = 0;                  // Adler-32 of the raw bytes
        size_t pos = 0;
        do {
            const size_t n = std::min<size_t>(65535, raw.size() - pos);
            z.push_back(pos + n == raw.size() ? 1 : 0); // BFINAL, BTYPE = 00 (stored)
            z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
            z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
            for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
            z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
            pos += n;
        } while (pos < raw.size());
        const uint32_t adler = (b << 16) | a;
        for (int k = 3; k >= 0; --k) z.push_back((uint8_t)(adler >> (8 * k)));
        std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
        auto be32 = [](std::vector<uint8_t> &v, uint32_t x) { for (int k = 3; k >= 0; --k) v.push_back((uint8_t)(x >> (8 * k))); };
        auto chunk = [&](const char *type, const std::vector<uint8_t> &body) {
            be32(out, (uint32_t)body.size());
            std::vector<uint8_t> td(type, type + 4);
            td.insert(td.end(), body.begin(), body.end());
            uint32_t crc = 0xFFFFFFFFu; // CRC-32 (IEEE 802.3) of type + data
            for (uint8_t byte : td) {
                crc ^= byte;
                for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
            }
            out.insert(out.end(), td.begin(), td.end());
            be32(out, ~crc);
        };
        std::vector<uint8_t> ihdr;
        be32(ihdr, width_); be32(ihdr, height_);
        ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0}); // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
        chunk("IHDR", ihdr);
        chunk("IDAT", z);
        chunk("IEND", {});
        return out;
    }
    bool write_png(const std::string &path) const {
        const std::vector<uint8_t> bytes = to_png();
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
        return std::fclose(f) == 0 && ok;
    }
    // write_ppm (canvas.rs:69-93) with MaxWidthWriter's 70-column wrap (113-181): a line is broken at the
    // last space at or before column 70.
    std::string to_ppm() const {
        std::string out = "P3\n" + std::to_string(width_) + " " + std::to_string(height_) + "\n255\n";
        for (uint32_t y = 0; y < height_; ++y) {
            std::string line;
            for (uint32_t x = 0; x < width_; ++x) {
                uint8_t c[3];
                data[(size_t)y * width_ + x].to_u8(c);
                for (int k = 0; k < 3; ++k) {
                    if (x > 0 || k > 0) line += ' ';
                    line += std::to_string((int)c[k]);
                }
            }
            while (line.size() > 70) {
                size_t i = 70;
                while (i > 0 && line[i] != ' ') --i;
                if (line[i] != ' ') { i = line.find(' ', 70); if (i == std::string::npos) break; }
                out += line.substr(0, i) + "\n";
                line = line.substr(i + 1);
            }
            out += line + "\n";
        }
        return out;
    }
};

// ---------------------------------------------------------------------------------------------
// world.rs — World, and the flattener that feeds rtmi_scene_create
// ---------------------------------------------------------------------------------------------
struct FlatScene {
    std::vector<rtmi_leaf> leaves;
    std::vector<rtmi_node> nodes;
    std::vector<int32_t> children, top_nodes, top_cast_shadow;
    std::vector<rtmi_material> materials;
    std::vector<rtmi_pattern> patterns;
    std::vector<rtmi_light> lights;
    int32_t max_depth = 10;
    std::vector<rtmi_photon> photons; // World::photon_map, if the gather is to run
    int32_t n_nearest = 0;
    double max_search_radius = 0.0;
    bool direct_illumination = true;
    rtmi_scene_desc desc_;

    const rtmi_scene_desc *desc() {
        std::memset(&desc_, 0, sizeof desc_);
        desc_.abi_version = RTMI_ABI_VERSION;
        desc_.max_depth = max_depth;
        desc_.n_leaves = (int32_t)leaves.size(); desc_.leaves = leaves.data();
        desc_.n_nodes = (int32_t)nodes.size(); desc_.nodes = nodes.data();
        desc_.n_children = (int32_t)children.size(); desc_.children = children.data();
        desc_.n_top = (int32_t)top_nodes.size(); desc_.top_nodes = top_nodes.data(); desc_.top_cast_shadow = top_cast_shadow.data();
        desc_.n_materials = (int32_t)materials.size(); desc_.materials = materials.data();
        desc_.n_patterns = (int32_t)patterns.size(); desc_.patterns = patterns.data();
        desc_.n_lights = (int32_t)lights.size(); desc_.lights = lights.data();
        desc_.n_photons = (int64_t)photons.size(); desc_.photons = photons.empty() ? nullptr : photons.data();
        desc_.n_nearest = n_nearest; desc_.max_search_radius = max_search_radius;
        desc_.no_direct_illumination = direct_illumination ? 0 : 1;
        return &desc_;
    }

    int32_t add_pattern(const Pattern &p) {
        rtmi_pattern q;
        std::memset(&q, 0, sizeof q);
        q.kind = p.kind;
        set3(q.a, p.a.red, p.a.green, p.a.blue); set3(q.b, p.b.red, p.b.green, p.b.blue);
        p.inv_transform_.into_flat(q.inv);
        for (size_t i = 0; i < patterns.size(); ++i)
            if (!std::memcmp(&patterns[i], &q, sizeof q)) return (int32_t)i;
        patterns.push_back(q);
        return (int32_t)patterns.size() - 1;
    }
    int32_t add_material(const Phong &m) {
        rtmi_material q;
        std::memset(&q, 0, sizeof q);
        set3(q.color, m.color_.red, m.color_.green, m.color_.blue);
        q.pattern = m.has_pattern ? add_pattern(m.pattern_) : -1;
        q.emissive = m.emissive_; q.diffuse = m.diffuse_; q.specular = m.specular_; q.shininess = m.shininess_;
        q.reflective = m.reflective_; q.transparency = m.transparency_; q.refractive_index = m.refractive_index_;
        // most recently added first: meshes repeat one material thousands of times
        for (size_t i = materials.size(); i-- > 0;)
            if (!std::memcmp(&materials[i], &q, sizeof q)) return (int32_t)i;
        materials.push_back(q);
        return (int32_t)materials.size() - 1;
    }
    static void put_aabb(rtmi_node &n, const Aabb &a) {
        set3(n.aabb_min, a.min_p.x, a.min_p.y, a.min_p.z);
        set3(n.aabb_max, a.max_p.x, a.max_p.y, a.max_p.z);
    }
    int32_t add_item(const SceneItem &it) {
        rtmi_node n;
        std::memset(&n, 0, sizeof n);
        if (auto s = it.as_shape()) {
            rtmi_leaf l;
            std::memset(&l, 0, sizeof l);
            l.kind = s->kind; l.material = add_material(s->material_); l.closed = s->closed ? 1 : 0;
            s->inv_cumulative_transform_.into_flat(l.inv);
            l.minimum = s->minimum; l.maximum = s->maximum;
            set3(l.p1, s->p1.x, s->p1.y, s->p1.z); set3(l.p2, s->p2.x, s->p2.y, s->p2.z); set3(l.p3, s->p3.x, s->p3.y, s->p3.z);
            set3(l.n1, s->n1.x, s->n1.y, s->n1.z); set3(l.n2, s->n2.x, s->n2.y, s->n2.z); set3(l.n3, s->n3.x, s->n3.y, s->n3.z);
            leaves.push_back(l);
            n.kind = RTMI_NODE_LEAF; n.first = (int32_t)leaves.size() - 1; n.count = 1;
            put_aabb(n, s->aabb());
        } else if (auto c = it.as_csg()) {
            int32_t l = add_item(c->items[0]), r = add_item(c->items[1]);
            n.kind = c->op == CsgOp::Union ? RTMI_NODE_CSG_UNION : c->op == CsgOp::Intersection ? RTMI_NODE_CSG_INTERSECTION : RTMI_NODE_CSG_DIFFERENCE;
            n.first = l; n.count = r;
            put_aabb(n, c->aabb());
        } else {
            const Group *g = it.as_group();
            const BoundingGroup *b = it.as_bounded();
            if (b) g = &b->group;
            std::vector<int32_t> kids;
            kids.reserve(g->items.size());
            for (auto &c : g->items) kids.push_back(add_item(c));
            n.kind = b ? RTMI_NODE_BOUNDED : RTMI_NODE_GROUP;
            n.first = (int32_t)children.size(); n.count = (int32_t)kids.size();
            children.insert(children.end(), kids.begin(), kids.end());
            put_aabb(n, b ? b->aabb_ : g->aabb());
        }
        nodes.push_back(n);
        return (int32_t)nodes.size() - 1;
    }
};

// photon_map.rs:15-72 — a photon as the map stores it
struct StoredPhoton {
    rtmi_photon p;
    StoredPhoton(const Point &position, const Vector &direction, const Color &power) { // StoredPhoton::new + compress_color (photon_map.rs:23-39)
        std::memset(&p, 0, sizeof p);
        p.position[0] = (float)position.x; p.position[1] = (float)position.y; p.position[2] = (float)position.z;
        p.direction[0] = (float)direction.x; p.direction[1] = (float)direction.y; p.direction[2] = (float)direction.z;
        const float x = (float)std::fmax(std::fmax(power.red, power.green), power.blue);
        const double c[3] = {power.red, power.green, power.blue};
        for (int k = 0; k < 3; ++k) {
            float v = 255.0f * (float)c[k] / x;
            v = std::fmin(std::fmax(v, 0.0f), 255.0f); // (fmax / fmin drop a NaN, as f32::max / min do)
            p.power_rgb[k] = (uint8_t)v;
        }
        p.power_scale = x;
    }
    StoredPhoton scale_power(double s) const { StoredPhoton q = *this; q.p.power_scale *= (float)s; return q; } // photon_map.rs:61-64
    Color power() const { // decompress_color, photon_map.rs:41-47
        return color((double)p.power_rgb[0] * (double)p.power_scale / 255.0, (double)p.power_rgb[1] * (double)p.power_scale / 255.0,
                     (double)p.power_rgb[2] * (double)p.power_scale / 255.0);
    }
};

struct World {
    std::vector<Light> lights;
    std::vector<SceneItem> objects;
    uint32_t max_reflection_depth = 10; // world.rs:61
    // world.rs:18-22.  The map is either GIVEN (set_photon_map) or traced on the device (compute_photon_map, below); its
    // gather in Phong::shade_hit (materials.rs:376-384) runs in the render kernels.
    std::vector<rtmi_photon> photon_map;
    size_t n_nearest = 0;
    double max_search_radius = 0.0; // (<= 0: unlimited)
    bool direct_illumination_enabled = true, direct_photon_map_enabled = false, diffuse_photon_map_enabled = true, caustic_photon_map_enabled = true;
    void enable_direct_illumination(bool f) { direct_illumination_enabled = f; }
    void enable_direct_photon_map(bool f) { direct_photon_map_enabled = f; }
    void enable_diffuse_photon_map(bool f) { diffuse_photon_map_enabled = f; }
    void enable_caustic_photon_map(bool f) { caustic_photon_map_enabled = f; }
    bool photon_map_enabled() const { return direct_photon_map_enabled || caustic_photon_map_enabled || diffuse_photon_map_enabled; } // world.rs:124-128
    void drop_photon_map() { photon_map.clear(); } // world.rs:182-184
    // what compute_photon_map leaves in the World (world.rs:203-206): the photons, each already scaled by 1 / n_photons
    void set_photon_map(const std::vector<StoredPhoton> &photons, size_t nearest, double max_radius) {
        photon_map.clear();
        for (const auto &q : photons) photon_map.push_back(q.p);
        n_nearest = nearest; max_search_radius = max_radius;
    }

    // World::compute_photon_map (world.rs:186-209): rtmi_trace_photons on a device copy of this world, then the map is
    // kept for the renders that follow.  `seed` has no counterpart (the reference draws from thread_rng).
    inline rtmi_photon_result compute_photon_map(size_t n_photons, size_t nearest, double max_radius, uint64_t seed = 0x5EED, int precision = RTMI_PRECISION_F32);

    static World empty() { return World(); }
    static World default_world() { // world.rs:25-53
        World w;
        w.add_light(PointLight(point(-10, 10, -10), color(1, 1, 1)));
        w.add_item(sphere().with_material(Phong().with_color(color(0.8, 1.0, 0.6)).with_emissive(0.1).with_diffuse(0.7).with_specular(0.2)));
        w.add_item(sphere().with_material(Phong().with_emissive(0.1)).with_transform(scaling(0.5, 0.5, 0.5)));
        return w;
    }
    void add_light(const Light &l) { lights.push_back(l); }
    void add_item(const SceneItem &it) { objects.push_back(it); }
    void finalize_scene() { // world.rs:98-105
        for (auto &o : objects) o.update_transform(Matrix::identity());
        for (auto &o : objects) o.update_aabb();
    }
    // extension: the reference hard-codes 10 (world.rs:61); the Scheme twin and BASELINE config 2 use 5
    void set_max_reflection_depth(uint32_t d) { max_reflection_depth = d; }

    std::unique_ptr<FlatScene> flatten() const {
        auto fs = std::make_unique<FlatScene>();
        fs->max_depth = (int32_t)max_reflection_depth;
        if (photon_map_enabled()) { fs->photons = photon_map; fs->n_nearest = (int32_t)n_nearest; fs->max_search_radius = max_search_radius; }
        fs->direct_illumination = direct_illumination_enabled;
        for (auto &l : lights) fs->lights.push_back(l.l);
        for (auto &o : objects) {
            fs->top_nodes.push_back(fs->add_item(o));
            fs->top_cast_shadow.push_back(o.cast_shadow() ? 1 : 0);
        }
        return fs;
    }
};

// RAII handle on a device-resident scene (rtmi_scene_create / rtmi_scene_destroy)
struct DeviceScene {
    rtmi_scene *handle = nullptr;
    explicit DeviceScene(const World &w) {
        auto fs = w.flatten();
        if (rtmi_scene_create(fs->desc(), &handle) != RTMI_OK) throw std::runtime_error(std::string("rtmi_scene_create: ") + rtmi_last_error());
    }
    DeviceScene(const DeviceScene &) = delete;
    DeviceScene &operator=(const DeviceScene &) = delete;
    ~DeviceScene() { if (handle) rtmi_scene_destroy(handle); }
};

inline rtmi_photon_result World::compute_photon_map(size_t n_photons, size_t nearest, double max_radius, uint64_t seed, int precision) {
    drop_photon_map(); // "Save some memory", world.rs:192 (and the device copy below carries no map)
    DeviceScene scene(*this);
    rtmi_photon_opts o;
    std::memset(&o, 0, sizeof o);
    o.seed = seed; o.n_emit = (int64_t)n_photons; o.precision = precision;
    o.enable_direct = direct_photon_map_enabled; o.enable_diffuse = diffuse_photon_map_enabled; o.enable_caustic = caustic_photon_map_enabled;
    rtmi_photon_result res;
    if (rtmi_trace_photons(scene.handle, &o, nullptr, 0, &res) != RTMI_OK) throw std::runtime_error(std::string("rtmi_trace_photons: ") + rtmi_last_error());
    std::vector<rtmi_photon> found((size_t)res.n_stored);
    if (res.n_stored > 0 && rtmi_trace_photons(scene.handle, &o, found.data(), res.n_stored, &res) != RTMI_OK)
        throw std::runtime_error(std::string("rtmi_trace_photons: ") + rtmi_last_error());
    photon_map.swap(found);
    n_nearest = nearest; max_search_radius = max_radius;
    return res;
}

// ---------------------------------------------------------------------------------------------
// camera.rs
// ---------------------------------------------------------------------------------------------
struct Camera {
    uint32_t hsize_, vsize_;
    double field_of_view_;
    Matrix transform_, inv_transform_;
    double pixel_size_, half_width_, half_height_;
    double pixel_allowed_standard_error = 1e-1;
    uint16_t pixel_min_samples = 5;
    double focal_distance = 3e100, aperture_size = 0.0;

    Camera(uint32_t hsize, uint32_t vsize, double field_of_view) : hsize_(hsize), vsize_(vsize), field_of_view_(field_of_view) { // camera.rs:35-64
        double half_view = std::tan(field_of_view / 2.0);
        double aspect = (double)hsize / (double)vsize;
        if (aspect >= 1.0) { half_width_ = half_view; half_height_ = half_view / aspect; }
        else { half_width_ = half_view * aspect; half_height_ = half_view; }
        pixel_size_ = half_width_ * 2.0 / (double)hsize;
    }
    uint32_t hsize() const { return hsize_; }
    uint32_t vsize() const { return vsize_; }
    double field_of_view() const { return field_of_view_; }
    double pixel_size() const { return pixel_size_; }
    const Matrix &transform() const { return transform_; }
    const Matrix &inv_transform() const { return inv_transform_; }
    void set_transform(const Matrix &t) { transform_ = t; inv_transform_ = t.inverse(); }
    Camera with_transform(const Matrix &t) const { Camera c = *this; c.set_transform(t); return c; }
    Camera with_view_transform(const Point &from, const Point &to, const Vector &up) const { return with_transform(Matrix::view(from, to, up)); }
    void set_allowed_standard_error(double se) { pixel_allowed_standard_error = se; }
    void set_min_samples(uint16_t n) { pixel_min_samples = n; }
    void set_aperture_size(double s) { aperture_size = s; }
    void set_focal_distance(double d) { focal_distance = d; }

    rtmi_camera abi() const {
        rtmi_camera c;
        std::memset(&c, 0, sizeof c);
        c.hsize = (int32_t)hsize_; c.vsize = (int32_t)vsize_;
        inv_transform_.into_flat(c.inv);
        c.pixel_size = pixel_size_; c.half_width = half_width_; c.half_height = half_height_;
        c.allowed_standard_error = pixel_allowed_standard_error; c.min_samples = pixel_min_samples;
        c.focal_distance = focal_distance; c.aperture_size = aperture_size;
        return c;
    }

    // Camera::render (camera.rs:155-161) on the GPU.  `opts` selects seed / precision / sample cap.
    Canvas render(const DeviceScene &scene, const rtmi_render_opts *opts = nullptr, rtmi_stats *stats = nullptr) const {
        rtmi_camera c = abi();
        rtmi_render_opts o;
        std::memset(&o, 0, sizeof o);
        if (opts) o = *opts;
        std::vector<float> rgb((size_t)hsize_ * vsize_ * 3);
        if (rtmi_render(scene.handle, &c, &o, rgb.data(), stats) != RTMI_OK) throw std::runtime_error(std::string("rtmi_render: ") + rtmi_last_error());
        Canvas canvas(hsize_, vsize_);
        for (size_t i = 0; i < canvas.data.size(); ++i) canvas.data[i] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        return canvas;
    }
    Canvas render(const World &world, const rtmi_render_opts *opts = nullptr, rtmi_stats *stats = nullptr) const {
        DeviceScene scene(world);
        return render(scene, opts, stats);
    }

    // The preview of render_live / render_interactive (camera.rs:162-248: pixels reach the window while the frame is
    // traced, through trace_pixels' pixel_callback, camera.rs:250-265) - pass by pass instead of pixel by pixel,
    // which is how a GPU delivers a frame.  Pass k renders the whole frame with at most caps[k] samples per pixel and
    // hands it to pass_callback(k, n_passes, canvas); after the capped passes comes the frame as render() gives it
    // (opts->max_samples, 0 = the open-ended loop), which is also what is returned.  Sample s of a pixel does not depend
    // on the pass (counter RNG), so every pass refines the picture before it.  A callback that returns false ends the
    // preview there, as a key event does in render_interactive (camera.rs:220-226); the canvas of that pass is returned.
    template <class PassCallback>
    Canvas render_progressive(const DeviceScene &scene, PassCallback &&pass_callback, const std::vector<int> &caps = {1, 4, 16, 64},
                              const rtmi_render_opts *opts = nullptr, rtmi_stats *stats = nullptr) const {
        rtmi_render_opts o;
        std::memset(&o, 0, sizeof o);
        if (opts) o = *opts;
        const int final_cap = o.max_samples;
        std::vector<int> passes;
        for (int c : caps)
            if (c > 0 && (final_cap <= 0 || c < final_cap) && (passes.empty() || c > passes.back())) passes.push_back(c);
        passes.push_back(final_cap);
        // rtmi_progressive_*: the per-pixel sums stay on the device between the passes, so a pass costs only the samples
        // it adds and the last one leaves exactly the frame Camera::render gives (stats: summed over the passes)
        rtmi_camera c = abi();
        rtmi_progressive *prog = nullptr;
        if (rtmi_progressive_begin(scene.handle, &c, &o, &prog) != RTMI_OK) throw std::runtime_error(std::string("rtmi_progressive_begin: ") + rtmi_last_error());
        Canvas canvas(hsize_, vsize_);
        std::vector<float> rgb((size_t)hsize_ * vsize_ * 3);
        if (stats) std::memset(stats, 0, sizeof *stats);
        for (size_t k = 0; k < passes.size(); ++k) {
            rtmi_stats st;
            if (rtmi_progressive_pass(prog, passes[k], rgb.data(), &st) != RTMI_OK) {
                const std::string msg = rtmi_last_error();
                rtmi_progressive_end(prog);
                throw std::runtime_error("rtmi_progressive_pass: " + msg);
            }
            if (stats) {
                for (int i = 0; i < RTMI_N_STATS; ++i) stats->counter[i] += st.counter[i];
                stats->kernel_ms += st.kernel_ms;
            }
            for (size_t i = 0; i < canvas.data.size(); ++i) canvas.data[i] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
            if (!pass_callback((int)k, (int)passes.size(), static_cast<const Canvas &>(canvas))) break;
        }
        rtmi_progressive_end(prog);
        return canvas;
    }
};

} // namespace rtmi
#endif // RTMI_HPP
