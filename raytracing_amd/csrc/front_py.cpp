// front_py.cpp — pybind11 binding of the C++ host front end (include/rtmi.hpp).
// Plumbing only: every class and method keeps the reference's name so that the Python parity
// tests read like the reference's own `#[test]`s.
#include <pybind11/numpy.h>
#include <pybind11/operators.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "rtmi.hpp"

namespace py = pybind11;
using namespace rtmi;

struct CameraAbi {
    rtmi_camera c;
    uintptr_t address() const { return (uintptr_t)&c; }
};

static py::array_t<float> canvas_to_numpy(const Canvas &cv) {
    py::array_t<float> a({(py::ssize_t)cv.height(), (py::ssize_t)cv.width(), (py::ssize_t)3});
    auto r = a.mutable_unchecked<3>();
    for (uint32_t y = 0; y < cv.height(); ++y)
        for (uint32_t x = 0; x < cv.width(); ++x) {
            Color c = cv.get_pixel(x, y);
            r(y, x, 0) = (float)c.red; r(y, x, 1) = (float)c.green; r(y, x, 2) = (float)c.blue;
        }
    return a;
}

static rtmi_render_opts make_opts(uint64_t seed, int max_samples, int precision) {
    rtmi_render_opts o;
    std::memset(&o, 0, sizeof o);
    o.seed = seed; o.max_samples = max_samples; o.precision = precision;
    return o;
}

PYBIND11_MODULE(_front, m) {
    m.doc() = "C++ host front end of rtmi (mirrors mbillingr/raytracing's Rust scene API)";
    m.attr("EPSILON") = EPSILON;

    py::class_<Point>(m, "Point")
        .def(py::init([](double x, double y, double z) { return Point{x, y, z}; }))
        .def_readwrite("x", &Point::x).def_readwrite("y", &Point::y).def_readwrite("z", &Point::z)
        .def("__sub__", [](const Point &a, const Point &b) { return a - b; })
        .def("__sub__", [](const Point &a, const Vector &b) { return a - b; })
        .def("__add__", [](const Point &a, const Vector &b) { return a + b; })
        .def("tuple", [](const Point &a) { return py::make_tuple(a.x, a.y, a.z); })
        .def("__repr__", [](const Point &a) { return "point(" + std::to_string(a.x) + ", " + std::to_string(a.y) + ", " + std::to_string(a.z) + ")"; });
    py::class_<Vector>(m, "Vector")
        .def(py::init([](double x, double y, double z) { return Vector{x, y, z}; }))
        .def_readwrite("x", &Vector::x).def_readwrite("y", &Vector::y).def_readwrite("z", &Vector::z)
        .def("dot", &Vector::dot).def("cross", &Vector::cross).def("len", &Vector::len).def("square_len", &Vector::square_len)
        .def("normalized", &Vector::normalized).def("reflect", &Vector::reflect)
        .def(py::self + py::self).def(py::self - py::self).def(-py::self)
        .def("__mul__", [](const Vector &a, double s) { return a * s; })
        .def("__rmul__", [](const Vector &a, double s) { return a * s; })
        .def("__truediv__", [](const Vector &a, double s) { return a / s; })
        .def("tuple", [](const Vector &a) { return py::make_tuple(a.x, a.y, a.z); })
        .def("__repr__", [](const Vector &a) { return "vector(" + std::to_string(a.x) + ", " + std::to_string(a.y) + ", " + std::to_string(a.z) + ")"; });
    m.def("point", &point);
    m.def("vector", &vector);

    py::class_<Color>(m, "Color")
        .def(py::init([](double r, double g, double b) { return Color{r, g, b}; }))
        .def_readwrite("red", &Color::red).def_readwrite("green", &Color::green).def_readwrite("blue", &Color::blue)
        .def("clip", &Color::clip)
        .def("to_u8", [](const Color &c) { uint8_t o[3]; c.to_u8(o); return py::make_tuple((int)o[0], (int)o[1], (int)o[2]); })
        .def_static("from_hsv", &Color::from_hsv)
        .def(py::self + py::self).def(py::self - py::self).def(py::self * py::self)
        .def("__mul__", [](const Color &a, double s) { return a * s; })
        .def("__rmul__", [](const Color &a, double s) { return a * s; })
        .def("__truediv__", [](const Color &a, double s) { return a / s; })
        .def("tuple", [](const Color &a) { return py::make_tuple(a.red, a.green, a.blue); });
    m.def("color", &color);

    py::class_<Matrix>(m, "Matrix")
        .def(py::init<>())
        .def_static("identity", &Matrix::identity)
        .def_static("from_rows", [](const std::vector<double> &v) {
            if (v.size() != 16) throw std::invalid_argument("need 16 values");
            double a[16];
            std::copy(v.begin(), v.end(), a);
            return Matrix(a);
        })
        .def_static("view", &Matrix::view)
        .def("__call__", [](const Matrix &a, int i, int j) { return a(i, j); })
        .def("__mul__", [](const Matrix &a, const Matrix &b) { return a * b; })
        .def("__mul__", [](const Matrix &a, const Point &b) { return a * b; })
        .def("__mul__", [](const Matrix &a, const Vector &b) { return a * b; })
        .def("transpose", &Matrix::transpose).def("inverse", &Matrix::inverse).def("determinant", &Matrix::determinant)
        .def("is_invertible", &Matrix::is_invertible).def("is_identity", &Matrix::is_identity).def("approx_eq", &Matrix::approx_eq)
        .def("translate", &Matrix::translate).def("scale", &Matrix::scale).def("rotate", &Matrix::rotate)
        .def("into_flat", [](const Matrix &a) { std::vector<double> v(16); a.into_flat(v.data()); return v; });
    m.def("translation", &translation);
    m.def("scaling", &scaling);
    m.def("rotation_x", &rotation_x);
    m.def("rotation_y", &rotation_y);
    m.def("rotation_z", &rotation_z);
    m.def("rotation", &rotation);
    m.def("shearing", &shearing);
    m.def("view_transform", &view_transform);

    py::class_<Aabb>(m, "Aabb")
        .def(py::init<>())
        .def_static("new", &Aabb::make)
        .def_readwrite("min_p", &Aabb::min_p).def_readwrite("max_p", &Aabb::max_p)
        .def("center", &Aabb::center).def("merge", &Aabb::merge).def("extend", &Aabb::extend).def("transform", &Aabb::transform);

    py::class_<Pattern>(m, "Pattern")
        .def("with_transform", &Pattern::with_transform).def("set_transform", &Pattern::set_transform)
        .def("inv_transform", &Pattern::inv_transform)
        .def_readonly("kind", &Pattern::kind);
    m.def("stripe_pattern", &stripe_pattern);
    m.def("gradient_pattern", &gradient_pattern);
    m.def("ring_pattern", &ring_pattern);
    m.def("checkers_pattern", &checkers_pattern);
    m.def("test_pattern", &test_pattern);

    py::class_<Phong>(m, "Phong")
        .def(py::init<>())
        .def(py::init<const Color &, double, double, double, double, double, double, double>())
        .def_static("default", []() { return Phong(); })
        .def_static("new", [](const Color &c, double e, double d, double s, double sh, double r, double t, double n) { return Phong(c, e, d, s, sh, r, t, n); })
        .def_static("new_pattern", &Phong::new_pattern)
        .def("with_color", &Phong::with_color).def("with_rgb", &Phong::with_rgb).def("with_hsv", &Phong::with_hsv)
        .def("with_pattern", &Phong::with_pattern).def("with_emissive", &Phong::with_emissive).def("with_diffuse", &Phong::with_diffuse)
        .def("with_specular", &Phong::with_specular).def("with_shininess", &Phong::with_shininess)
        .def("with_reflective", &Phong::with_reflective).def("with_transparency", &Phong::with_transparency)
        .def("with_refractive_index", &Phong::with_refractive_index)
        .def("set_color", &Phong::set_color).def("set_pattern", &Phong::set_pattern).def("set_emissive", &Phong::set_emissive)
        .def("set_diffuse", &Phong::set_diffuse).def("set_specular", &Phong::set_specular).def("set_shininess", &Phong::set_shininess)
        .def("set_reflective", &Phong::set_reflective).def("set_transparency", &Phong::set_transparency)
        .def("set_refractive_index", &Phong::set_refractive_index)
        .def("color", &Phong::color).def("emissive", &Phong::emissive).def("diffuse", &Phong::diffuse).def("specular", &Phong::specular)
        .def("shininess", &Phong::shininess).def("reflective", &Phong::reflective).def("transparency", &Phong::transparency)
        .def("refractive_index", &Phong::refractive_index);

    py::class_<Light>(m, "Light")
        .def_property_readonly("kind", [](const Light &l) { return l.l.kind; })
        .def("address", [](const Light &l) { return (uintptr_t)&l.l; });
    m.def("PointLight", &PointLight);
    m.def("RealisticPointLight", &RealisticPointLight);
    m.def("AmbientLight", &AmbientLight);
    m.def("SphereLight", &SphereLight);
    m.def("DiscLight", &DiscLight);
    m.def("Beam", &Beam);

    py::class_<Shape>(m, "Shape")
        .def("with_material", &Shape::with_material).def("with_transform", &Shape::with_transform)
        .def("with_cast_shadow", &Shape::with_cast_shadow).def("set_material", &Shape::set_material)
        .def("set_transform", &Shape::set_transform).def("set_cast_shadow", &Shape::set_cast_shadow)
        .def("update_transform", &Shape::update_transform).def("material", &Shape::material)
        .def("inv_transform", &Shape::inv_transform).def("cast_shadow", &Shape::cast_shadow)
        .def("world_to_object", &Shape::world_to_object).def("normal_to_world", &Shape::normal_to_world)
        .def("aabb", &Shape::aabb)
        .def_readonly("kind", &Shape::kind).def_readonly("p1", &Shape::p1).def_readonly("p2", &Shape::p2).def_readonly("p3", &Shape::p3)
        .def_readonly("n1", &Shape::n1).def_readonly("n2", &Shape::n2).def_readonly("n3", &Shape::n3)
        .def_readonly("minimum", &Shape::minimum).def_readonly("maximum", &Shape::maximum).def_readonly("closed", &Shape::closed);
    m.def("sphere", &sphere);
    m.def("glass_sphere", &glass_sphere);
    m.def("plane", &plane);
    m.def("cube", &cube);
    m.def("cylinder", &cylinder, py::arg("minimum") = -INFINITY, py::arg("maximum") = INFINITY, py::arg("closed") = false);
    m.def("cone", &cone, py::arg("minimum") = -INFINITY, py::arg("maximum") = INFINITY, py::arg("closed") = false);
    m.def("triangle", &triangle);
    m.def("smooth_triangle", &smooth_triangle);

    py::class_<SceneItem>(m, "SceneItem")
        .def(py::init<const Shape &>()).def(py::init<const Group &>()).def(py::init<const BoundingGroup &>()).def(py::init<const CsgPair &>())
        .def("with_transform", &SceneItem::with_transform).def("cast_shadow", &SceneItem::cast_shadow)
        .def("as_shape", [](const SceneItem &s) -> py::object { auto p = s.as_shape(); return p ? py::cast(*p) : py::none(); })
        .def("as_group", [](const SceneItem &s) -> py::object { auto p = s.as_group(); return p ? py::cast(*p) : py::none(); })
        .def("as_bounded", [](const SceneItem &s) -> py::object { auto p = s.as_bounded(); return p ? py::cast(*p) : py::none(); })
        .def("as_csg", [](const SceneItem &s) -> py::object { auto p = s.as_csg(); return p ? py::cast(*p) : py::none(); })
        .def("aabb", &SceneItem::aabb);
    py::implicitly_convertible<Shape, SceneItem>();
    py::implicitly_convertible<Group, SceneItem>();
    py::implicitly_convertible<BoundingGroup, SceneItem>();
    py::implicitly_convertible<CsgPair, SceneItem>();

    py::class_<Group>(m, "Group")
        .def(py::init<>())
        .def("with_child", &Group::with_child).def("add_child", &Group::add_child).def("get_child", &Group::get_child)
        .def("aint_empty", &Group::aint_empty).def("with_transform", &Group::with_transform).def("set_transform", &Group::set_transform)
        .def("update_transform", &Group::update_transform).def("update_aabb", &Group::update_aabb).def("aabb", &Group::aabb)
        .def("__len__", [](const Group &g) { return g.items.size(); });
    py::class_<BoundingGroup>(m, "BoundingGroup")
        .def(py::init<>())
        .def("add_child", &BoundingGroup::add_child).def("with_transform", &BoundingGroup::with_transform)
        .def("update_aabb", [](BoundingGroup &b) { return b.update_aabb(); })
        .def_readonly("group", &BoundingGroup::group).def_readonly("aabb", &BoundingGroup::aabb_);
    py::enum_<CsgOp>(m, "CsgOp").value("Union", CsgOp::Union).value("Intersection", CsgOp::Intersection).value("Difference", CsgOp::Difference);
    py::class_<CsgPair>(m, "CsgPair")
        .def_static("new", &make_csg)
        .def("with_cast_shadow", &CsgPair::with_cast_shadow).def("with_transform", &CsgPair::with_transform)
        .def("set_transform", &CsgPair::set_transform).def("update_transform", &CsgPair::update_transform)
        .def("cast_shadow", &CsgPair::cast_shadow).def("left", &CsgPair::left).def("right", &CsgPair::right)
        .def("aabb", &CsgPair::aabb).def_readonly("op", &CsgPair::op);
    m.def("group", &group);
    py::class_<TriangleMesh>(m, "TriangleMesh")
        .def(py::init<>()).def_static("new", []() { return TriangleMesh(); })
        .def_static("from_mesh", &TriangleMesh::from_mesh)
        .def("aint_empty", &TriangleMesh::aint_empty).def("add_triangle", &TriangleMesh::add_triangle)
        .def("get_triangle", &TriangleMesh::get_triangle).def("into_group", &TriangleMesh::into_group)
        .def("__len__", [](const TriangleMesh &t) { return t.faces.size(); });
    m.def("bounding_group", &bounding_group);
    m.def("csg_union", &csg_union);
    m.def("csg_intersection", &csg_intersection);
    m.def("csg_difference", &csg_difference);
    m.def("build_bounding_tree", [](const Group &g, size_t n) { return build_bounding_tree(g, n); });

    py::class_<ObjParser>(m, "ObjParser")
        .def_static("parse_str", &ObjParser::parse_str, py::arg("input"), py::arg("skip_texture_lines") = false)
        .def("ignored", &ObjParser::ignored).def("get_group", &ObjParser::get_group).def("into_group", &ObjParser::into_group)
        .def_readonly("vertices", &ObjParser::vertices).def_readonly("normals", &ObjParser::normals);

    py::class_<Canvas>(m, "Canvas")
        .def(py::init<uint32_t, uint32_t>())
        .def("width", &Canvas::width).def("height", &Canvas::height).def("clear", &Canvas::clear)
        .def("set_pixel", &Canvas::set_pixel).def("add_to_pixel", &Canvas::add_to_pixel).def("get_pixel", &Canvas::get_pixel)
        .def("average_brightness", &Canvas::average_brightness)
        .def("to_ppm", &Canvas::to_ppm)
        .def("to_png", [](const Canvas &c) { auto v = c.to_png(); return py::bytes(reinterpret_cast<const char *>(v.data()), v.size()); })
        .def("write_png", &Canvas::write_png)
        .def("to_rgb8", [](const Canvas &c) {
            auto v = c.to_rgb8();
            py::array_t<uint8_t> a({(py::ssize_t)c.height(), (py::ssize_t)c.width(), (py::ssize_t)3});
            std::memcpy(a.mutable_data(), v.data(), v.size());
            return a;
        })
        .def("numpy", &canvas_to_numpy);
    m.def("canvas", [](uint32_t w, uint32_t h) { return Canvas(w, h); });

    py::class_<FlatScene>(m, "FlatScene")
        .def("desc_address", [](FlatScene &f) { return (uintptr_t)f.desc(); })
        .def("leaf_address", [](FlatScene &f, size_t i) { return (uintptr_t)&f.leaves.at(i); })
        .def("material_address", [](FlatScene &f, size_t i) { return (uintptr_t)&f.materials.at(i); })
        .def("pattern_address", [](FlatScene &f, size_t i) { return (uintptr_t)&f.patterns.at(i); })
        .def("light_address", [](FlatScene &f, size_t i) { return (uintptr_t)&f.lights.at(i); })
        .def("leaf_material", [](FlatScene &f, size_t i) { return f.leaves.at(i).material; })
        .def("leaf_kind", [](FlatScene &f, size_t i) { return f.leaves.at(i).kind; })
        .def("leaf_inv", [](FlatScene &f, size_t i) { return std::vector<double>(f.leaves.at(i).inv, f.leaves.at(i).inv + 16); })
        .def("node", [](FlatScene &f, size_t i) {
            const rtmi_node &n = f.nodes.at(i);
            return py::make_tuple(n.kind, n.first, n.count, py::make_tuple(n.aabb_min[0], n.aabb_min[1], n.aabb_min[2]),
                                  py::make_tuple(n.aabb_max[0], n.aabb_max[1], n.aabb_max[2]));
        })
        .def_readonly("children", &FlatScene::children).def_readonly("top_nodes", &FlatScene::top_nodes)
        .def_readonly("top_cast_shadow", &FlatScene::top_cast_shadow)
        .def_property_readonly("n_leaves", [](const FlatScene &f) { return f.leaves.size(); })
        .def_property_readonly("n_nodes", [](const FlatScene &f) { return f.nodes.size(); })
        .def_property_readonly("n_materials", [](const FlatScene &f) { return f.materials.size(); })
        .def_property_readonly("n_patterns", [](const FlatScene &f) { return f.patterns.size(); })
        .def_property_readonly("n_lights", [](const FlatScene &f) { return f.lights.size(); })
        .def_readwrite("max_depth", &FlatScene::max_depth);

    py::class_<World>(m, "World")
        .def(py::init<>())
        .def_static("empty", &World::empty)
        .def_static("default", &World::default_world)
        .def("add_light", &World::add_light).def("add_item", &World::add_item).def("finalize_scene", &World::finalize_scene)
        .def("set_max_reflection_depth", &World::set_max_reflection_depth)
        .def_readwrite("max_reflection_depth", &World::max_reflection_depth)
        .def_readwrite("objects", &World::objects).def_readwrite("lights", &World::lights)
        .def("set_object", [](World &w, size_t i, const SceneItem &it) { w.objects.at(i) = it; })
        .def("enable_direct_illumination", &World::enable_direct_illumination).def("enable_direct_photon_map", &World::enable_direct_photon_map)
        .def("enable_diffuse_photon_map", &World::enable_diffuse_photon_map).def("enable_caustic_photon_map", &World::enable_caustic_photon_map)
        .def("photon_map_enabled", &World::photon_map_enabled).def("drop_photon_map", &World::drop_photon_map)
        .def("set_photon_map", &World::set_photon_map, py::arg("photons"), py::arg("n_nearest"), py::arg("max_search_radius") = 0.0)
        .def("n_photons", [](const World &w) { return w.photon_map.size(); })
        .def("compute_photon_map", [](World &w, size_t n_photons, size_t nearest, double max_radius, uint64_t seed, int precision) {
            const rtmi_photon_result r = w.compute_photon_map(n_photons, nearest, max_radius, seed, precision);
            py::dict d;
            d["n_stored"] = r.n_stored; d["n_rays"] = r.n_rays; d["n_truncated"] = r.n_truncated; d["kernel_ms"] = r.kernel_ms;
            return d;
        }, py::arg("n_photons"), py::arg("n_nearest"), py::arg("max_search_radius"), py::arg("seed") = 0x5EED, py::arg("precision") = 0)
        // (the stored photons as one array of 32-byte records, rtmi_photon)
        .def("photon_bytes", [](const World &w) { return py::bytes((const char *)w.photon_map.data(), w.photon_map.size() * sizeof(rtmi_photon)); })
        // (positions float32[n,3], directions float32[n,3], power_scale float32[n], power_rgb uint8[n,3]: the stored photons as arrays)
        .def("set_photon_arrays", [](World &w, py::array_t<float, py::array::c_style | py::array::forcecast> pos,
                                     py::array_t<float, py::array::c_style | py::array::forcecast> dir,
                                     py::array_t<float, py::array::c_style | py::array::forcecast> scale,
                                     py::array_t<uint8_t, py::array::c_style | py::array::forcecast> rgb, size_t nearest, double max_radius) {
            const size_t n = (size_t)scale.size();
            if ((size_t)pos.size() != 3 * n || (size_t)dir.size() != 3 * n || (size_t)rgb.size() != 3 * n) throw std::invalid_argument("photon arrays disagree in length");
            w.photon_map.assign(n, rtmi_photon{});
            for (size_t i = 0; i < n; ++i) {
                rtmi_photon &q = w.photon_map[i];
                for (int k = 0; k < 3; ++k) { q.position[k] = pos.data()[3 * i + k]; q.direction[k] = dir.data()[3 * i + k]; q.power_rgb[k] = rgb.data()[3 * i + k]; }
                q.power_scale = scale.data()[i];
            }
            w.n_nearest = nearest; w.max_search_radius = max_radius;
        }, py::arg("positions"), py::arg("directions"), py::arg("power_scale"), py::arg("power_rgb"), py::arg("n_nearest"), py::arg("max_search_radius") = 0.0)
        .def("flatten", [](const World &w) { return w.flatten(); });

    py::class_<StoredPhoton>(m, "StoredPhoton")
        .def(py::init<const Point &, const Vector &, const Color &>())
        .def("scale_power", &StoredPhoton::scale_power).def("power", &StoredPhoton::power)
        .def("power_scale", [](const StoredPhoton &q) { return q.p.power_scale; })
        .def("power_rgb", [](const StoredPhoton &q) { return std::vector<int>{q.p.power_rgb[0], q.p.power_rgb[1], q.p.power_rgb[2]}; })
        .def("position", [](const StoredPhoton &q) { return std::vector<float>{q.p.position[0], q.p.position[1], q.p.position[2]}; });

    py::class_<DeviceScene>(m, "DeviceScene")
        .def(py::init<const World &>())
        .def("handle", [](const DeviceScene &d) { return (uintptr_t)d.handle; });

    py::class_<CameraAbi>(m, "CameraAbi").def("address", &CameraAbi::address);

    py::class_<Camera>(m, "Camera")
        .def(py::init<uint32_t, uint32_t, double>())
        .def_static("new", [](uint32_t h, uint32_t v, double fov) { return Camera(h, v, fov); })
        .def("hsize", &Camera::hsize).def("vsize", &Camera::vsize).def("field_of_view", &Camera::field_of_view)
        .def("pixel_size", &Camera::pixel_size).def("transform", &Camera::transform).def("inv_transform", &Camera::inv_transform)
        .def("set_transform", &Camera::set_transform).def("with_transform", &Camera::with_transform)
        .def("with_view_transform", &Camera::with_view_transform)
        .def("set_allowed_standard_error", &Camera::set_allowed_standard_error).def("set_min_samples", &Camera::set_min_samples)
        .def("set_aperture_size", &Camera::set_aperture_size).def("set_focal_distance", &Camera::set_focal_distance)
        .def("abi", [](const Camera &c) { CameraAbi a; a.c = c.abi(); return a; })
        .def("render",
             [](const Camera &c, const DeviceScene &s, uint64_t seed, int max_samples, int precision) {
                 rtmi_render_opts o = make_opts(seed, max_samples, precision);
                 Canvas cv;
                 { py::gil_scoped_release rel; cv = c.render(s, &o, nullptr); }
                 return cv;
             },
             py::arg("scene"), py::arg("seed") = 0x5EED, py::arg("max_samples") = 0, py::arg("precision") = 0)
        .def("render",
             [](const Camera &c, const World &w, uint64_t seed, int max_samples, int precision) {
                 rtmi_render_opts o = make_opts(seed, max_samples, precision);
                 Canvas cv;
                 { py::gil_scoped_release rel; cv = c.render(w, &o, nullptr); }
                 return cv;
             },
             py::arg("world"), py::arg("seed") = 0x5EED, py::arg("max_samples") = 0, py::arg("precision") = 0)
        .def("render_progressive",
             [](const Camera &c, const DeviceScene &s, py::function callback, std::vector<int> caps, uint64_t seed, int max_samples, int precision) {
                 rtmi_render_opts o = make_opts(seed, max_samples, precision);
                 return c.render_progressive(s, [&](int k, int n, const Canvas &cv) { return py::cast<bool>(callback(k, n, cv)); }, caps, &o, nullptr);
             },
             py::arg("scene"), py::arg("callback"), py::arg("caps") = std::vector<int>{1, 4, 16, 64}, py::arg("seed") = 0x5EED,
             py::arg("max_samples") = 0, py::arg("precision") = 0);
}
