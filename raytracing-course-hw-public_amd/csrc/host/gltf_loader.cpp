// gltf_loader.cpp — host glTF 2.0 front end producing the POD scene of include/rt_abi.h.
//
// Restates reference src/scene.h:183-501 (parse_gltf_scene) and the matrix/quaternion helpers it uses
// (src/geometry.h:158-265 matrix4, :267-353 matrix3). One-off I/O: stays plain C++ on the host.
// Float arithmetic follows the reference expression by expression (left-to-right sums, no FMA) so that the
// triangles handed to the device are bit-identical to scene.objects of the reference.
//
// Loader quirks kept on purpose:
//   * attribute accessors ignore accessor.byteOffset / byteStride / componentType (scene.h:120-133);
//   * tangents are looked up under the lowercase key "tangent" and read as tightly packed vec3 (scene.h:336);
//   * a node's "matrix" and its TRS are BOTH applied: parent * matrix * (T*R*S) (scene.h:228-230);
//   * the last camera node visited wins; aspectRatio from the file overrides the CLI aspect (scene.h:239-242);
//   * only modes 4 (triangles) and 5 (strips) emit geometry (scene.h:444-458).
// Differences, all in territory where the reference has undefined behaviour or std::terminate: missing
// "indices"/"material", out-of-range indices or texture ids return RT_ERR_FORMAT instead of crashing.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <future>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "../rt_error.h"
#include "loaded_scene.h"
#include "mini_json.h"

namespace {

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};
inline float dot4(const V4 &a, const V4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float dot3(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 norm3(const V3 &v) {
    float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x / l, v.y / l, v.z / l};
}
inline V4 norm4(const V4 &v) {
    float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    return {v.x / l, v.y / l, v.z / l, v.w / l};
}

struct M4 { // row-major rows, geometry.h:158-214
    float m[4][4];
    static M4 id() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
    static M4 translation(const V3 &p) { return {{{1, 0, 0, p.x}, {0, 1, 0, p.y}, {0, 0, 1, p.z}, {0, 0, 0, 1}}}; }
    static M4 scale(const V3 &s) { return {{{s.x, 0, 0, 0}, {0, s.y, 0, 0}, {0, 0, s.z, 0}, {0, 0, 0, 1}}}; }
    static M4 rotation(float x, float y, float z, float w) { // geometry.h:179-196
        return {{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0},
                 {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0},
                 {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0},
                 {0, 0, 0, 1}}};
    }
    V4 mul(const V4 &v) const { // geometry.h:230-238
        V4 r;
        float *o = &r.x;
        for (int i = 0; i < 4; ++i)
            o[i] = dot4({m[i][0], m[i][1], m[i][2], m[i][3]}, v);
        return r;
    }
    V3 apply(const V3 &v) const { // geometry.h:259-261
        V4 r = mul({v.x, v.y, v.z, 1});
        return {r.x, r.y, r.z};
    }
};
inline M4 operator*(const M4 &a, const M4 &b) { // geometry.h:216-228
    M4 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k)
                r.m[i][k] += a.m[i][j] * b.m[j][k];
    return r;
}

struct M3 { // geometry.h:267-313
    float m[3][3];
    float len2row(int r) const { return m[r][0] * m[r][0] + m[r][1] * m[r][1] + m[r][2] * m[r][2]; }
    float maj(int r, int c) const {
        int r1 = (r + 1) % 3, r2 = (r + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        return m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1];
    }
    M3 rs_fast_inv_t() const {
        M3 r;
        float d2 = len2row(0) * len2row(1) * len2row(2);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                r.m[i][j] = maj(i, j) / d2;
        return r;
    }
    V3 apply(const V3 &v) const { // geometry.h:329-337
        return {dot3({m[0][0], m[0][1], m[0][2]}, v), dot3({m[1][0], m[1][1], m[1][2]}, v), dot3({m[2][0], m[2][1], m[2][2]}, v)};
    }
};

std::string read_text(const std::filesystem::path &p) {
    std::ifstream in(p, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + p.string());
    std::ostringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

struct FormatError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

} // namespace

namespace {

void load_impl(const std::filesystem::path &gltf_path, float ar, rt_loaded_scene &res) {
    using mjson::Value;
    const Value root = mjson::parse(read_text(gltf_path));
    int scene_idx = root.contains("scene") ? (int)root["scene"].as_int() : 0;
    const Value &scenes = root["scenes"];
    static const Value null_value;
    const Value &scene_info = (scenes.is_array() && (size_t)scene_idx < scenes.size()) ? scenes[(size_t)scene_idx] : null_value;

    // scene.h:193-202
    std::vector<std::vector<uint8_t>> buffers;
    for (const Value &buf_info : root["buffers"].arr) {
        std::string uri = buf_info["uri"].as_string();
        std::ifstream in(gltf_path.parent_path() / uri, std::ios::binary);
        buffers.emplace_back();
        auto &buf = buffers.back();
        buf.resize((size_t)buf_info["byteLength"].as_int());
        in.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)buf.size());
    }
    // scene.h:204-209. The reference decodes the images one after the other; here every file is decoded on a thread of its
    // own while this thread goes on with the scene graph (SURVEY 8f-2 "async decode"): textures are the bulk of a real
    // asset's load time. Results are collected in file order, so texture indices are unchanged.
    struct Decoded {
        uint32_t w = 0, h = 0;
        uint8_t *px = nullptr;
        int rc = RT_OK;
        std::string err;
    };
    std::vector<std::future<Decoded>> decoding;
    for (const Value &texture_info : root["textures"].arr) {
        int img = (int)texture_info["source"].as_int();
        std::string uri = root["images"][(size_t)img]["uri"].as_string();
        std::string path = (gltf_path.parent_path() / uri).string();
        decoding.push_back(std::async(std::launch::async, [path] {
            Decoded d;
            d.rc = rt_image_decode_file(path.c_str(), &d.w, &d.h, &d.px);
            if (d.rc != RT_OK)
                d.err = rt::last_error(); // thread-local: carried back to the loading thread
            return d;
        }));
    }
    auto collect_textures = [&]() {
        std::string first_error;
        for (auto &f : decoding) {
            Decoded d = f.get();
            if (d.rc != RT_OK && first_error.empty())
                first_error = d.err;
            res.texels.push_back(d.px); // freed by ~rt_loaded_scene, also on the error path
            res.textures.push_back({d.w, d.h, d.px});
        }
        decoding.clear();
        if (!first_error.empty())
            throw std::runtime_error(first_error);
    };
    const size_t n_textures_declared = decoding.size();
    struct JoinDecoders { // an exception on the way (malformed scene graph) must not abandon running decoders or their buffers
        std::vector<std::future<Decoded>> &pending;
        rt_loaded_scene &res;
        ~JoinDecoders() {
            for (auto &f : pending)
                if (f.valid())
                    res.texels.push_back(f.get().px);
        }
    } join_decoders{decoding, res};
    auto tex_index = [&](const Value &v) -> int32_t {
        int64_t idx = v["index"].as_int();
        if (idx < 0 || (size_t)idx >= n_textures_declared)
            throw FormatError("texture index out of range");
        return (int32_t)idx;
    };

    std::vector<int32_t> material_slot(root["materials"].size(), -1);
    auto get_material = [&](int material_idx) -> uint32_t { // scene.h:261-316
        if (material_idx < 0 || (size_t)material_idx >= material_slot.size())
            throw FormatError("material index out of range");
        if (material_slot[(size_t)material_idx] >= 0)
            return (uint32_t)material_slot[(size_t)material_idx];
        const Value &material = root["materials"][(size_t)material_idx];
        rt_material_desc mat{};
        mat.color[0] = mat.color[1] = mat.color[2] = mat.color[3] = 1;
        mat.roughness = 1.0f;
        mat.metallic = 1.0f;
        mat.ior = 1.5f;
        mat.color_tex = mat.emissive_tex = mat.metallic_roughness_tex = mat.normal_tex = RT_TEX_NONE;
        float emission[3] = {0, 0, 0};
        if (material.contains("emissiveFactor"))
            for (int k = 0; k < 3; ++k)
                emission[k] = material["emissiveFactor"][(size_t)k].as_float();
        const Value &strength = material["extensions"]["KHR_materials_emissive_strength"]["emissiveStrength"];
        if (!strength.is_null()) {
            float s = strength.as_float();
            for (float &e : emission)
                e *= s;
        }
        if (material.contains("emissiveTexture"))
            mat.emissive_tex = tex_index(material["emissiveTexture"]);
        for (int k = 0; k < 3; ++k)
            mat.emission[k] = emission[k];
        if (material.contains("pbrMetallicRoughness")) {
            const Value &pbr = material["pbrMetallicRoughness"];
            if (pbr.contains("baseColorFactor"))
                for (int k = 0; k < 4; ++k)
                    mat.color[k] = pbr["baseColorFactor"][(size_t)k].as_float();
            if (pbr.contains("baseColorTexture"))
                mat.color_tex = tex_index(pbr["baseColorTexture"]);
            if (pbr.contains("metallicRoughnessTexture"))
                mat.metallic_roughness_tex = tex_index(pbr["metallicRoughnessTexture"]);
            mat.roughness = pbr.contains("roughnessFactor") ? pbr["roughnessFactor"].as_float() : 1.0f;
            mat.metallic = pbr.contains("metallicFactor") ? pbr["metallicFactor"].as_float() : 1.0f;
        }
        if (material.contains("normalTexture"))
            mat.normal_tex = tex_index(material["normalTexture"]);
        res.materials.push_back(mat);
        material_slot[(size_t)material_idx] = (int32_t)res.materials.size() - 1;
        return (uint32_t)res.materials.size() - 1;
    };

    struct View {
        const uint8_t *ptr = nullptr;
        size_t count = 0;
        size_t avail = 0; // bytes available from ptr to the end of the buffer
    };
    auto attribute_view = [&](int64_t accessor_idx) -> View { // interpret_accessor scene.h:118-133
        const Value &accessor = root["accessors"][(size_t)accessor_idx];
        const Value &bv = root["bufferViews"][(size_t)accessor["bufferView"].as_int()];
        const auto &buffer = buffers.at((size_t)bv["buffer"].as_int());
        size_t offset = bv.contains("byteOffset") ? (size_t)bv["byteOffset"].as_int() : 0;
        if (offset > buffer.size())
            throw FormatError("bufferView outside buffer");
        return {buffer.data() + offset, (size_t)accessor["count"].as_int(), buffer.size() - offset};
    };

    rt_camera cam{};
    std::function<void(int, const M4 &)> handle_node = [&](int node_idx, const M4 &parent) {
        const Value &node = root["nodes"][(size_t)node_idx];
        float qx = 0, qy = 0, qz = 0, qw = 1; // geometry::quaternion() = (a=1, 0,0,0)
        if (node.contains("rotation")) {
            const Value &r = node["rotation"];
            qx = r[0].as_float();
            qy = r[1].as_float();
            qz = r[2].as_float();
            qw = r[3].as_float();
        }
        V3 tr{0, 0, 0}, sc{1, 1, 1};
        if (node.contains("translation"))
            tr = {node["translation"][0].as_float(), node["translation"][1].as_float(), node["translation"][2].as_float()};
        if (node.contains("scale"))
            sc = {node["scale"][0].as_float(), node["scale"][1].as_float(), node["scale"][2].as_float()};
        M4 trs = M4::id();
        if (node.contains("matrix")) { // parse_mat4 scene.h:101-108 (column-major source)
            const Value &s = node["matrix"];
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c)
                    trs.m[r][c] = s[(size_t)(c * 4 + r)].as_float();
        }
        // matrix4::transform = translation * rotation * scale (geometry.h:252-257); scene.h:228-230
        M4 local = (M4::translation(tr) * M4::rotation(qx, qy, qz, qw)) * M4::scale(sc);
        M4 transform = (parent * trs) * local;
        M3 lin;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                lin.m[r][c] = transform.m[r][c];
        M3 normal_transform = lin.rs_fast_inv_t();

        if (node.contains("camera")) { // scene.h:234-255
            const Value &camera = root["cameras"][(size_t)node["camera"].as_int()];
            const Value &persp = camera["perspective"];
            float fov_y = persp["yfov"].as_float();
            float aspect_ratio = persp.contains("aspectRatio") ? persp["aspectRatio"].as_float() : ar;
            V4 p = transform.mul({0, 0, 0, 1});
            V4 f = norm4(transform.mul({0, 0, -1, 0}));
            V4 u = norm4(transform.mul({0, 1, 0, 0}));
            V4 r = norm4(transform.mul({1, 0, 0, 0}));
            cam.position[0] = p.x, cam.position[1] = p.y, cam.position[2] = p.z;
            cam.forward[0] = f.x, cam.forward[1] = f.y, cam.forward[2] = f.z;
            cam.up[0] = u.x, cam.up[1] = u.y, cam.up[2] = u.z;
            cam.right[0] = r.x, cam.right[1] = r.y, cam.right[2] = r.z;
            cam.fov_x = std::atan(std::tan(fov_y / 2) * aspect_ratio) * 2;
        }
        if (node.contains("mesh")) {
            const Value &mesh = root["meshes"][(size_t)node["mesh"].as_int()];
            for (const Value &prim : mesh["primitives"].arr) {
                if (!prim.contains("material"))
                    throw FormatError("primitive without material (the reference aborts here)");
                uint32_t mat_id = get_material((int)prim["material"].as_int());
                const Value &attrs = prim["attributes"];
                View coords = attribute_view(attrs["POSITION"].as_int());
                View normals = attrs.contains("NORMAL") ? attribute_view(attrs["NORMAL"].as_int()) : View{};
                View tangents = attrs.contains("tangent") ? attribute_view(attrs["tangent"].as_int()) : View{};
                View texcoords = attrs.contains("TEXCOORD_0") ? attribute_view(attrs["TEXCOORD_0"].as_int()) : View{};
                if (!prim.contains("indices"))
                    throw FormatError("primitive without indices (the reference aborts here)");
                // load_indices scene.h:138-181
                const Value &iacc = root["accessors"][(size_t)prim["indices"].as_int()];
                const Value &ibv = root["bufferViews"][(size_t)iacc["bufferView"].as_int()];
                const auto &ibuf = buffers.at((size_t)ibv["buffer"].as_int());
                size_t ioff = (ibv.contains("byteOffset") ? (size_t)ibv["byteOffset"].as_int() : 0) +
                              (iacc.contains("byteOffset") ? (size_t)iacc["byteOffset"].as_int() : 0);
                size_t cnt = (size_t)iacc["count"].as_int();
                int ctype = (int)iacc["componentType"].as_int();
                size_t isz = ctype == 5121 ? 1 : ctype == 5123 ? 2 : ctype == 5125 ? 4 : 0;
                if (!isz)
                    throw std::runtime_error("illegal scalar type"); // scene.h:179
                if (ioff + cnt * isz > ibuf.size())
                    throw FormatError("index accessor outside buffer");
                auto get_index = [&](size_t i) -> size_t {
                    const uint8_t *p = ibuf.data() + ioff + i * isz;
                    if (isz == 1)
                        return *p;
                    if (isz == 2) {
                        uint16_t v;
                        std::memcpy(&v, p, 2);
                        return v;
                    }
                    uint32_t v;
                    std::memcpy(&v, p, 4);
                    return v;
                };
                auto vec3_at = [&](const View &v, size_t i) -> V3 {
                    if ((i + 1) * 12 > v.avail)
                        throw FormatError("vertex index outside buffer");
                    V3 r;
                    std::memcpy(&r, v.ptr + i * 12, 12);
                    return r;
                };
                auto push_obj = [&](size_t i1, size_t i2, size_t i3) { // scene.h:409-442
                    size_t idx[3] = {i1, i2, i3};
                    V3 p[3];
                    for (int k = 0; k < 3; ++k)
                        p[k] = transform.apply(vec3_at(coords, idx[k]));
                    V3 n[3];
                    if (normals.count) {
                        for (int k = 0; k < 3; ++k)
                            n[k] = norm3(normal_transform.apply(vec3_at(normals, idx[k])));
                    } else { // triangle::normal geometry.h:477-479
                        V3 v{p[1].x - p[0].x, p[1].y - p[0].y, p[1].z - p[0].z}, u{p[2].x - p[0].x, p[2].y - p[0].y, p[2].z - p[0].z};
                        V3 c{v.y * u.z - v.z * u.y, v.z * u.x - v.x * u.z, v.x * u.y - v.y * u.x};
                        n[0] = n[1] = n[2] = norm3(c);
                    }
                    for (int k = 0; k < 3; ++k) {
                        res.positions.insert(res.positions.end(), {p[k].x, p[k].y, p[k].z});
                        res.normals.insert(res.normals.end(), {n[k].x, n[k].y, n[k].z});
                        float uv[2] = {0, 0};
                        if (texcoords.count) {
                            if ((idx[k] + 1) * 8 > texcoords.avail)
                                throw FormatError("texcoord index outside buffer");
                            std::memcpy(uv, texcoords.ptr + idx[k] * 8, 8);
                        }
                        res.texcoords.insert(res.texcoords.end(), {uv[0], uv[1]});
                        V3 t = tangents.count ? vec3_at(tangents, idx[k]) : V3{1, 0, 0};
                        res.tangents.insert(res.tangents.end(), {t.x, t.y, t.z});
                    }
                    res.material_ids.push_back(mat_id);
                };
                int mode = prim.contains("mode") ? (int)prim["mode"].as_int() : 4;
                if (mode == 4) {
                    for (size_t i = 0; i + 2 < cnt; i += 3)
                        push_obj(get_index(i), get_index(i + 1), get_index(i + 2));
                } else if (mode == 5) {
                    for (size_t i = 2; i < cnt; ++i) {
                        size_t off = i & 1;
                        push_obj(get_index(i - 2), get_index(i - 1 + off), get_index(i - off));
                    }
                }
            }
        }
        if (node.contains("children"))
            for (const Value &child : node["children"].arr)
                handle_node((int)child.as_int(), transform);
    };

    if (scene_info.is_null()) { // scene.h:468-477
        for (int i = 0, n = (int)root["nodes"].size(); i < n; ++i)
            handle_node(i, M4::id());
    } else {
        for (const Value &n : scene_info["nodes"].arr)
            handle_node((int)n.as_int(), M4::id());
    }

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
    collect_textures();
    d.n_textures = (uint32_t)res.textures.size();
    d.textures = res.textures.data();
    d.camera = cam;
    d.bg_color[0] = d.bg_color[1] = d.bg_color[2] = 1.0f; // ENV_MAP_INTENSITY config.h:36, main.cpp:28
    d.bg_texture = RT_TEX_NONE;                           // USE_ENV_MAP = false config.h:37 (rt_loaded_set_env_map changes it)
    d.ray_depth = 8;                                      // DEFAULT_RAY_DEPTH config.h:17, scene.h:186
}

} // namespace

extern "C" int rt_gltf_load(const char *path, float aspect, rt_loaded_scene **out) {
    if (!path || !out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_gltf_load: null argument");
    auto *s = new rt_loaded_scene();
    try {
        load_impl(std::filesystem::path(path), aspect, *s);
    } catch (const FormatError &e) {
        delete s;
        return rt::fail(RT_ERR_FORMAT, e.what());
    } catch (const std::exception &e) {
        delete s;
        return rt::fail(RT_ERR_IO, e.what());
    }
    *out = s;
    return RT_OK;
}

// main.cpp:28-31 for USE_ENV_MAP = true: scene.bg_color = intensity, scene.bg = Texture::load_img(path)
extern "C" int rt_loaded_set_env_map(rt_loaded_scene *s, const char *image_path, float intensity) {
    if (!s || !image_path)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_loaded_set_env_map: null argument");
    uint32_t w = 0, h = 0;
    uint8_t *px = nullptr;
    if (int rc = rt_image_decode_file(image_path, &w, &h, &px); rc != RT_OK)
        return rc;
    s->texels.push_back(px);
    rt_texture_desc t{};
    t.width = w;
    t.height = h;
    t.rgba8 = px;
    s->textures.push_back(t);
    s->desc.textures = s->textures.data(); // the vector may have moved
    s->desc.n_textures = (uint32_t)s->textures.size();
    s->desc.bg_texture = (int32_t)s->textures.size() - 1;
    s->desc.bg_color[0] = s->desc.bg_color[1] = s->desc.bg_color[2] = intensity;
    return RT_OK;
}

// USE_TEXTURES = false (config.h:31-32, geometry.h:547-574): Texture::sample returns the texture's FIRST texel, unfiltered and without gamma,
// whatever the coordinates. That is exactly what it does for a 1x1 texture, so every texture is cut down to its first texel.
extern "C" int rt_loaded_disable_textures(rt_loaded_scene *s) {
    if (!s)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_loaded_disable_textures: null argument");
    for (rt_texture_desc &t : s->textures)
        t.width = t.height = 1; // rgba8 keeps pointing at the picture: its first four bytes are texel 0
    return RT_OK;
}

// scene.h:479-498 for ADD_LIGHT_TRIANGLE = true (config.h:41-47): one more object, an emissive triangle given in the camera's frame
// (vertex = position + r.x * right + r.y * up + r.z * forward), geometric normal, zero texture coordinates, tangent (1, 0, 0), a default
// material (geometry.h:604-613) whose emission is the intensity. The reference fixes position and intensity at compile time.
extern "C" int rt_loaded_add_light_triangle(rt_loaded_scene *s, const float rel[9], float intensity) {
    if (!s || !rel)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_loaded_add_light_triangle: null argument");
    const rt_camera &cam = s->desc.camera;
    V3 p[3];
    for (int k = 0; k < 3; ++k) { // w + transform3(r, x, y, z), geometry.h:355-359: r.x * x + r.y * y + r.z * z, then the sum with w
        const float rx = rel[3 * k], ry = rel[3 * k + 1], rz = rel[3 * k + 2];
        float t[3];
        for (int c = 0; c < 3; ++c)
            t[c] = cam.position[c] + ((rx * cam.right[c] + ry * cam.up[c]) + rz * cam.forward[c]);
        p[k] = {t[0], t[1], t[2]};
    }
    const V3 v{p[1].x - p[0].x, p[1].y - p[0].y, p[1].z - p[0].z}, u{p[2].x - p[0].x, p[2].y - p[0].y, p[2].z - p[0].z};
    const V3 n = norm3({v.y * u.z - v.z * u.y, v.z * u.x - v.x * u.z, v.x * u.y - v.y * u.x}); // triangle::normal geometry.h:477-479
    for (int k = 0; k < 3; ++k) {
        s->positions.insert(s->positions.end(), {p[k].x, p[k].y, p[k].z});
        s->normals.insert(s->normals.end(), {n.x, n.y, n.z});
        s->texcoords.insert(s->texcoords.end(), {0.0f, 0.0f});
        s->tangents.insert(s->tangents.end(), {1.0f, 0.0f, 0.0f});
    }
    rt_material_desc m{};
    m.color[0] = m.color[1] = m.color[2] = m.color[3] = 1.0f;
    m.emission[0] = m.emission[1] = m.emission[2] = intensity;
    m.roughness = 1.0f;
    m.metallic = 1.0f;
    m.ior = 1.5f;
    m.color_tex = m.emissive_tex = m.metallic_roughness_tex = m.normal_tex = RT_TEX_NONE;
    s->materials.push_back(m);
    s->material_ids.push_back((uint32_t)s->materials.size() - 1u);
    rt_scene_desc &d = s->desc; // the vectors may have moved
    d.n_triangles = (uint32_t)s->material_ids.size();
    d.positions = s->positions.data();
    d.normals = s->normals.data();
    d.texcoords = s->texcoords.data();
    d.tangents = s->tangents.data();
    d.material_ids = s->material_ids.data();
    d.n_materials = (uint32_t)s->materials.size();
    d.materials = s->materials.data();
    return RT_OK;
}

extern "C" const rt_scene_desc *rt_loaded_desc(const rt_loaded_scene *s) { return s ? &s->desc : nullptr; }
extern "C" void rt_loaded_free(rt_loaded_scene *s) { delete s; }
