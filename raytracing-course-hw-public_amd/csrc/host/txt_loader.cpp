// txt_loader.cpp — the scene-txt front end (BASELINE configs 1-2), host C++.
//
// Status against the reference: HEAD has NO parser for these files (main unconditionally calls parse_gltf_scene,
// src/main.cpp:27) and renders triangles only (geometry.h:505). The files under sample_data/ are leftovers of earlier
// homework stages; what remains of their parser in HEAD is typed_read / ensure_variant (scene.h:28-40) and operator>>
// for vectors and quaternions (geometry.h:154-156: a quaternion is read as x y z w). The grammar below is therefore
// inferred from the data files (e.g. sample_data/scene-002.txt:1-73, homebrew_primitives/practice3_5.txt:1-52,
// practice5_2.txt:16-21) and the SEMANTICS are this project's own:
//
//   DIMENSIONS w h | RAY_DEPTH n | SAMPLES n          image size / Scene::ray_depth / samples recorded with the scene (the
//                                                     CLI's <width> <height> <samples> win, as in main.cpp:23-25,32-34)
//   BG_COLOR r g b                                    Scene::bg_color: the uniform environment radiance (scene.h:75,83-89)
//   CAMERA_POSITION|RIGHT|UP|FORWARD x y z, CAMERA_FOV_X rad      Camera (scene.h:60-72), taken verbatim
//   AMBIENT_LIGHT r g b, NEW_LIGHT, LIGHT_DIRECTION|POSITION|ATTENUATION|INTENSITY ...
//                                                     lights of the Whitted-style homework stages: HEAD's path tracer has no
//                                                     such lights (only emissive surfaces); parsed, counted, IGNORED
//   NEW_PRIMITIVE, then one of
//     TRIANGLE ax ay az bx by bz cx cy cz             -> one triangle
//     BOX sx sy sz            (half extents)          -> 12 triangles (outward winding)
//     ELLIPSOID rx ry rz | PLANE nx ny nz             -> analytic primitive (rt_primitive_desc; include/rt_primspec.h)
//     POSITION x y z, ROTATION x y z w                p_world = rotate(q, p_local) + position   (rt_quat_rotate)
//     COLOR r g b (default 0 0 0), EMISSION r g b     material::color (alpha 1) / material::emission
//     METALLIC | DIELECTRIC, IOR x                    mapped onto HEAD's metallic-roughness material (geometry.h:604-613):
//                                                     default: metallic 0, roughness 1; METALLIC: metallic 1, roughness 0
//                                                     (the BRDF clamps to MIN_ROUGHNESS, config.h:20); DIELECTRIC: metallic 0,
//                                                     roughness 0, ior as given — a glossy Fresnel coat over the base colour,
//                                                     NO refraction (HEAD has no transmission)
//
// TRIANGLE and BOX become ordinary triangles (vertices transformed here in binary32, -0.0 canonicalised to +0.0 as a glTF
// round trip through an identity node would), so they take the triangle path whose parity IS pinned: the same arrays
// exported as glTF render byte-identically in the unmodified reference (tests/test_scene_txt.py). Per-vertex normals are
// the face normal, texcoords 0, tangents (1,0,0): what parse_gltf_scene stores for a primitive without NORMAL /
// TEXCOORD_0 / tangent (scene.h:392-407, 423-430). ELLIPSOID / PLANE are "parity unpinned" (oracle == GPU only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "../../../include/rt_primspec.h"
#include "../rt_error.h"
#include "loaded_scene.h"

namespace {

struct TxtError {
    std::string msg;
};

struct Prim {
    int kind = 0; // 0 none, 1 ellipsoid, 2 plane, 3 box, 4 triangle
    float param[9] = {0};
    float pos[3] = {0, 0, 0};
    float rot[4] = {0, 0, 0, 1};
    float color[3] = {0, 0, 0};
    float emission[3] = {0, 0, 0};
    int surface = 0; // 0 diffuse, 1 metallic, 2 dielectric
    float ior = 1.0f;
    bool open = false;
};

void read_floats(std::istringstream &in, float *dst, int n, const std::string &cmd, int line) {
    for (int i = 0; i < n; ++i) {
        std::string tok;
        if (!(in >> tok))
            throw TxtError{"line " + std::to_string(line) + ": " + cmd + " expects " + std::to_string(n) + " numbers"};
        char *end = nullptr;
        dst[i] = std::strtof(tok.c_str(), &end);
        if (end == tok.c_str() || *end != '\0')
            throw TxtError{"line " + std::to_string(line) + ": " + cmd + ": '" + tok + "' is not a number"};
    }
}

void emit_triangle(rt_loaded_scene &res, const Prim &p, const float v[3][3], uint32_t material) {
    float w[3][3];
    for (int k = 0; k < 3; ++k) {
        float r[3];
        rt_quat_rotate(p.rot[0], p.rot[1], p.rot[2], p.rot[3], v[k], r);
        for (int c = 0; c < 3; ++c)
            w[k][c] = (r[c] + p.pos[c]) + 0.0f; // + 0.0f: -0.0 -> +0.0
    }
    // triangle::normal (geometry.h:477-479): norm(crs(b - a, c - a)), replicated on the three vertices (scene.h:427-430)
    const float vx = w[1][0] - w[0][0], vy = w[1][1] - w[0][1], vz = w[1][2] - w[0][2];
    const float ux = w[2][0] - w[0][0], uy = w[2][1] - w[0][1], uz = w[2][2] - w[0][2];
    const float cx = vy * uz - vz * uy, cy = vz * ux - vx * uz, cz = vx * uy - vy * ux;
    const float l = std::sqrt(cx * cx + cy * cy + cz * cz);
    const float n[3] = {cx / l, cy / l, cz / l};
    for (int k = 0; k < 3; ++k) {
        res.positions.insert(res.positions.end(), {w[k][0], w[k][1], w[k][2]});
        res.normals.insert(res.normals.end(), {n[0], n[1], n[2]});
        res.tangents.insert(res.tangents.end(), {1.0f, 0.0f, 0.0f});
        res.texcoords.insert(res.texcoords.end(), {0.0f, 0.0f});
    }
    res.material_ids.push_back(material);
}

void close_primitive(rt_loaded_scene &res, Prim &p, int line) {
    if (!p.open)
        return;
    if (p.kind == 0)
        throw TxtError{"line " + std::to_string(line) + ": NEW_PRIMITIVE without ELLIPSOID / PLANE / BOX / TRIANGLE"};
    rt_material_desc m{};
    m.color[0] = p.color[0], m.color[1] = p.color[1], m.color[2] = p.color[2], m.color[3] = 1.0f;
    m.emission[0] = p.emission[0], m.emission[1] = p.emission[1], m.emission[2] = p.emission[2];
    m.metallic = p.surface == 1 ? 1.0f : 0.0f;
    m.roughness = p.surface == 0 ? 1.0f : 0.0f;
    m.ior = p.surface == 2 ? p.ior : 1.5f; // glTF default ior 1.5 (scene.h:262)
    m.color_tex = m.emissive_tex = m.metallic_roughness_tex = m.normal_tex = RT_TEX_NONE;
    const uint32_t mat = (uint32_t)res.materials.size();
    res.materials.push_back(m);
    if (p.kind == 1 || p.kind == 2) {
        rt_primitive_desc d{};
        d.kind = p.kind == 1 ? RT_PRIM_ELLIPSOID : RT_PRIM_PLANE;
        d.material_id = mat;
        std::memcpy(d.param, p.param, 12);
        std::memcpy(d.position, p.pos, 12);
        std::memcpy(d.rotation, p.rot, 16);
        if (d.kind == RT_PRIM_PLANE) { // the plane's normal turns with the primitive; its test needs no frame change afterwards
            float n[3];
            rt_quat_rotate(p.rot[0], p.rot[1], p.rot[2], p.rot[3], p.param, n);
            std::memcpy(d.param, n, 12);
            d.rotation[0] = d.rotation[1] = d.rotation[2] = 0.0f;
            d.rotation[3] = 1.0f;
        }
        res.primitives.push_back(d);
    } else if (p.kind == 4) {
        const float v[3][3] = {{p.param[0], p.param[1], p.param[2]}, {p.param[3], p.param[4], p.param[5]}, {p.param[6], p.param[7], p.param[8]}};
        emit_triangle(res, p, v, mat);
    } else { // BOX: corners (+-sx, +-sy, +-sz); two triangles per face, counter-clockwise seen from outside
        const float s[3] = {p.param[0], p.param[1], p.param[2]};
        float c[8][3];
        for (int i = 0; i < 8; ++i)
            for (int a = 0; a < 3; ++a)
                c[i][a] = (i >> a) & 1 ? s[a] : -s[a];
        static const int quads[6][4] = {{0, 2, 3, 1}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 4, 6, 2}, {1, 3, 7, 5}};
        for (const auto &q : quads) {
            const float t0[3][3] = {{c[q[0]][0], c[q[0]][1], c[q[0]][2]}, {c[q[1]][0], c[q[1]][1], c[q[1]][2]}, {c[q[2]][0], c[q[2]][1], c[q[2]][2]}};
            const float t1[3][3] = {{c[q[0]][0], c[q[0]][1], c[q[0]][2]}, {c[q[2]][0], c[q[2]][1], c[q[2]][2]}, {c[q[3]][0], c[q[3]][1], c[q[3]][2]}};
            emit_triangle(res, p, t0, mat);
            emit_triangle(res, p, t1, mat);
        }
    }
    p = Prim();
}

void load_txt(const std::string &path, rt_loaded_scene &res) {
    std::ifstream in(path);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    rt_camera cam{};
    cam.right[0] = 1, cam.up[1] = 1, cam.forward[2] = -1;
    cam.fov_x = 1.5707963f;
    float bg[3] = {0, 0, 0};
    uint32_t ray_depth = 8; // DEFAULT_RAY_DEPTH config.h:17
    Prim cur;
    std::string text;
    int line = 0;
    bool any = false;
    while (std::getline(in, text)) {
        ++line;
        std::istringstream ls(text);
        std::string cmd;
        if (!(ls >> cmd))
            continue;
        any = true;
        float f[9];
        auto prim_cmd = [&]() -> Prim & {
            if (!cur.open)
                throw TxtError{"line " + std::to_string(line) + ": " + cmd + " outside NEW_PRIMITIVE"};
            return cur;
        };
        // DIMENSIONS / SAMPLES / RAY_DEPTH are counts: finite, non-negative, integral and small enough to be exact in a float
        auto as_count = [&](float v, float hi) -> uint32_t {
            if (!(v >= 0.0f && v <= hi) || v != std::floor(v))
                throw TxtError{"line " + std::to_string(line) + ": " + cmd + " out of range (an integer in 0.." + std::to_string((long long)hi) + ")"};
            return (uint32_t)v;
        };
        if (cmd == "DIMENSIONS") {
            read_floats(ls, f, 2, cmd, line);
            res.file_width = as_count(f[0], 16777216.0f), res.file_height = as_count(f[1], 16777216.0f);
        } else if (cmd == "RAY_DEPTH") {
            read_floats(ls, f, 1, cmd, line);
            ray_depth = as_count(f[0], 32.0f);
        } else if (cmd == "SAMPLES") {
            read_floats(ls, f, 1, cmd, line);
            res.file_samples = as_count(f[0], 16777216.0f);
        } else if (cmd == "BG_COLOR") {
            read_floats(ls, bg, 3, cmd, line);
        } else if (cmd == "CAMERA_POSITION") {
            read_floats(ls, cam.position, 3, cmd, line);
        } else if (cmd == "CAMERA_RIGHT") {
            read_floats(ls, cam.right, 3, cmd, line);
        } else if (cmd == "CAMERA_UP") {
            read_floats(ls, cam.up, 3, cmd, line);
        } else if (cmd == "CAMERA_FORWARD") {
            read_floats(ls, cam.forward, 3, cmd, line);
        } else if (cmd == "CAMERA_FOV_X") {
            read_floats(ls, &cam.fov_x, 1, cmd, line);
        } else if (cmd == "AMBIENT_LIGHT" || cmd == "LIGHT_DIRECTION" || cmd == "LIGHT_POSITION" || cmd == "LIGHT_ATTENUATION" || cmd == "LIGHT_INTENSITY") {
            read_floats(ls, f, 3, cmd, line); // syntax checked, value unused (see the header of this file)
            res.ignored_light_commands++;
        } else if (cmd == "NEW_LIGHT") {
            close_primitive(res, cur, line);
            res.ignored_lights++;
        } else if (cmd == "NEW_PRIMITIVE") {
            close_primitive(res, cur, line);
            cur.open = true;
        } else if (cmd == "ELLIPSOID" || cmd == "PLANE" || cmd == "BOX") {
            Prim &p = prim_cmd();
            read_floats(ls, p.param, 3, cmd, line);
            p.kind = cmd == "ELLIPSOID" ? 1 : cmd == "PLANE" ? 2 : 3;
        } else if (cmd == "TRIANGLE") {
            Prim &p = prim_cmd();
            read_floats(ls, p.param, 9, cmd, line);
            p.kind = 4;
        } else if (cmd == "POSITION") {
            read_floats(ls, prim_cmd().pos, 3, cmd, line);
        } else if (cmd == "ROTATION") {
            read_floats(ls, prim_cmd().rot, 4, cmd, line);
        } else if (cmd == "COLOR") {
            read_floats(ls, prim_cmd().color, 3, cmd, line);
        } else if (cmd == "EMISSION") {
            read_floats(ls, prim_cmd().emission, 3, cmd, line);
        } else if (cmd == "METALLIC") {
            prim_cmd().surface = 1;
        } else if (cmd == "DIELECTRIC") {
            prim_cmd().surface = 2;
        } else if (cmd == "IOR") {
            read_floats(ls, &prim_cmd().ior, 1, cmd, line);
        } else {
            throw TxtError{"line " + std::to_string(line) + ": unknown command '" + cmd + "'"};
        }
    }
    close_primitive(res, cur, line);
    if (!any)
        throw TxtError{"empty scene file"};
    if (res.primitives.size() > RT_MAX_PRIMITIVES)
        throw TxtError{"more than " + std::to_string(RT_MAX_PRIMITIVES) + " analytic primitives"};
    rt_scene_desc &d = res.desc;
    d.abi_version = RT_ABI_VERSION;
    d.n_triangles = (uint32_t)res.material_ids.size();
    d.positions = res.positions.data();
    d.normals = res.normals.data();
    d.texcoords = res.texcoords.data();
    d.tangents = res.tangents.data();
    d.material_ids = res.material_ids.data();
    d.n_materials = (uint32_t)res.materials.size();
    d.materials = res.materials.data();
    d.n_textures = 0;
    d.textures = nullptr;
    d.camera = cam;
    std::memcpy(d.bg_color, bg, 12);
    d.bg_texture = RT_TEX_NONE;
    d.ray_depth = ray_depth;
    d.n_primitives = (uint32_t)res.primitives.size();
    d.primitives = res.primitives.data();
}

bool ends_with(const std::string &s, const char *suffix) {
    const size_t n = std::strlen(suffix);
    if (s.size() < n)
        return false;
    for (size_t i = 0; i < n; ++i)
        if (std::tolower((unsigned char)s[s.size() - n + i]) != suffix[i])
            return false;
    return true;
}

} // namespace

extern "C" int rt_txt_load(const char *path, rt_loaded_scene **out) {
    if (!path || !out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_txt_load: null argument");
    auto *s = new rt_loaded_scene();
    try {
        load_txt(path, *s);
    } catch (const TxtError &e) {
        delete s;
        return rt::fail(RT_ERR_FORMAT, std::string(path) + ": " + e.msg);
    } catch (const std::exception &e) {
        delete s;
        return rt::fail(RT_ERR_IO, e.what());
    }
    *out = s;
    return RT_OK;
}

extern "C" int rt_scene_load(const char *path, float aspect, rt_loaded_scene **out) {
    if (!path || !out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_scene_load: null argument");
    return ends_with(path, ".txt") ? rt_txt_load(path, out) : rt_gltf_load(path, aspect, out);
}

extern "C" int rt_loaded_info(const rt_loaded_scene *s, uint32_t *width, uint32_t *height, uint32_t *samples, uint32_t *ignored_lights) {
    if (!s)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_loaded_info: null argument");
    if (width)
        *width = s->file_width;
    if (height)
        *height = s->file_height;
    if (samples)
        *samples = s->file_samples;
    if (ignored_lights)
        *ignored_lights = s->ignored_lights;
    return RT_OK;
}
