// ref_probe.cpp — golden-vector harness. TEST INFRASTRUCTURE, built only in this container into oracle/_ref/.
//
// It #includes the reference's own headers from /root/reference/src (nothing is copied into the repo) and dumps
// what the parity tests need below the PPM level (SURVEY 8c "golden vectors to manufacture" (2)):
//
//   ref_probe bvh      <gltf> <W> <H> <out.bin>            both BVHs: nodes (10 words each) + object order
//   ref_probe cast     <gltf> <W> <H> <rays.bin> <out.bin> closest hits for explicit rays (prim, b, c, t)
//   ref_probe primary  <gltf> <W> <H> <out.bin>            closest hits of pixel-centre rays gen_ray(camera,x,y)
//   ref_probe lightpdf <gltf> <W> <H> <rays.bin> <out.bin> bvh_mix_dist::pdf for explicit (x, dir) pairs
//   ref_probe scene    <gltf> <W> <H> <out.bin>            flattened scene.objects (positions/normals/uv/tangents)
//   ref_probe bgat     <gltf> <W> <H> <image> <dirs.bin> <out.bin>  Scene::bg_at (scene.h:83-89) with scene.bg = load_img(image) as main.cpp:29-31
//                                                          does under USE_ENV_MAP, for explicit directions (3 floats each) -> rgb
//   ref_probe envrender <gltf> <W> <H> <image> <spp> <out.ppm>  main.cpp:27-43 with the environment map loaded: the reference's own render
//   ref_probe lighttri <gltf> <W> <H> <out.bin>            the extra light source of scene.h:479-498 (compile-time off: ADD_LIGHT_TRIANGLE), put
//                                                          together here from the reference's own constants and helpers (config.h:41-47,
//                                                          geometry::transform3, triangle::normal): 9 positions, 3 normal floats, intensity
//   ref_probe lightrender <gltf> <W> <H> <spp> <out.ppm>   ... that object appended to scene.objects, then run_raytracer + Image::write
//   ref_probe notexrender <gltf> <W> <H> <spp> <out.ppm>   USE_TEXTURES = false (config.h:31-32; compile-time on): Texture::sample then returns data[0],
//                                                          which is its answer for a one-texel texture: every loaded texture is cut down to its first
//                                                          texel (the Texture objects stay where the materials point), then run_raytracer + Image::write
//   ref_probe sphere   <radius> 0 0 <rays.bin> <out.bin>   intersect_ray_sphere (raytracer.h:61-77, unused by the reference's render loop; the
//                                                          scene-txt ELLIPSOID restates it): (t1, t2) per ray
//   ref_probe texture  <image> 0 0 <out.bin>               geometry::Texture::load_img (the reference's stb_image build, 4 channels
//                                                          forced, geometry.h:584-598): [width, height, texel floats r g b a ...]
//
// Binary formats are little-endian u32/f32 arrays described next to each writer.
#define STB_IMAGE_IMPLEMENTATION
#include "raytracer.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

static void put_u32(std::vector<uint32_t> &o, uint32_t v) { o.push_back(v); }
static void put_f32(std::vector<uint32_t> &o, float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    o.push_back(u);
}
static std::vector<float> read_floats(const char *path) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    size_t n = in.tellg();
    in.seekg(0);
    std::vector<float> v(n / 4);
    in.read(reinterpret_cast<char *>(v.data()), n);
    return v;
}
static void write_words(const char *path, const std::vector<uint32_t> &o) {
    std::ofstream out(path, std::ios::binary);
    out.write(reinterpret_cast<const char *>(o.data()), o.size() * 4);
}

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: ref_probe <mode> <gltf> <W> <H> ...\n");
        return 2;
    }
    std::string mode = argv[1];
    if (mode == "sphere") { // no scene involved: the reference's quadratic solve for a sphere of radius r at the origin
        if (argc < 7)
            return 2;
        const float r = std::strtof(argv[2], nullptr);
        auto f = read_floats(argv[5]);
        std::vector<uint32_t> o;
        for (size_t i = 0; i + 5 < f.size(); i += 6) {
            geometry::ray ray{{f[i], f[i + 1], f[i + 2]}, {f[i + 3], f[i + 4], f[i + 5]}};
            auto [t1, t2] = intersect_ray_sphere(ray, r);
            put_f32(o, t1);
            put_f32(o, t2);
        }
        write_words(argv[6], o);
        return 0;
    }
    if (mode == "texture") { // no scene involved: what the reference's image decoder returns for one file
        std::vector<uint32_t> o;
        try {
            geometry::Texture t = geometry::Texture::load_img(argv[2]);
            put_u32(o, t.width);
            put_u32(o, t.height);
            for (const auto &c : t.data)
                for (int k = 0; k < 4; ++k)
                    put_f32(o, c.val[k]);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "texture: %s\n", e.what());
            return 1;
        }
        write_words(argv[5], o);
        return 0;
    }
    unsigned width = std::strtol(argv[3], nullptr, 10);
    unsigned height = std::strtol(argv[4], nullptr, 10);
    Scene scene = parse_gltf_scene(std::filesystem::path(argv[2]), static_cast<float>(width) / height);
    scene.bg_color = {ENV_MAP_INTENSITY, ENV_MAP_INTENSITY, ENV_MAP_INTENSITY};
    scene.camera.width = width;
    scene.camera.height = height;
    scene.samples = 1;
    if (mode == "bgat" || mode == "envrender") {
        if (argc < 8) {
            std::fprintf(stderr, "usage: ref_probe %s <gltf> <W> <H> <image> ...\n", mode.c_str());
            return 2;
        }
        scene.bg = geometry::Texture::load_img(argv[5]); // main.cpp:29-31, the branch USE_ENV_MAP = false compiles out
        if (mode == "envrender") {
            scene.samples = std::strtol(argv[6], nullptr, 10);
            Image img(width, height, scene.bg_color); // main.cpp:35-43
            run_raytracer(scene, img);
            std::ofstream out(argv[7], std::ios::binary);
            img.write(out);
            return 0;
        }
        std::vector<uint32_t> o;
        auto f = read_floats(argv[6]);
        for (size_t i = 0; i + 2 < f.size(); i += 3) {
            geometry::color3 c = scene.bg_at({f[i], f[i + 1], f[i + 2]});
            put_f32(o, c.r());
            put_f32(o, c.g());
            put_f32(o, c.b());
        }
        write_words(argv[7], o);
        return 0;
    }
    if (mode == "notexrender") {
        if (argc < 7)
            return 2;
        for (auto &t : scene.textures) {
            t.data.resize(1);
            t.width = t.height = 1;
        }
        scene.samples = std::strtol(argv[5], nullptr, 10);
        Image img(width, height, scene.bg_color);
        run_raytracer(scene, img);
        std::ofstream out(argv[6], std::ios::binary);
        img.write(out);
        return 0;
    }
    if (mode == "lighttri" || mode == "lightrender") {
        std::vector<uint32_t> o;
        geometry::triangle tr = *reinterpret_cast<const geometry::triangle *>(&LIGHT_TRIANGLE_RELATIVE_POS);
        tr.a() = scene.camera.position + geometry::transform3(tr.a(), scene.camera.right, scene.camera.up, scene.camera.forward);
        tr.b() = scene.camera.position + geometry::transform3(tr.b(), scene.camera.right, scene.camera.up, scene.camera.forward);
        tr.c() = scene.camera.position + geometry::transform3(tr.c(), scene.camera.right, scene.camera.up, scene.camera.forward);
        if (mode == "lightrender") { // ... appended to the scene as scene.h:480-497 does, then the reference's own render loop
            if (argc < 7)
                return 2;
            scene.objects.emplace_back();
            geometry::Object &ls = scene.objects.back();
            ls.shape = tr;
            ls.material.emission = {LIGHT_TRIANGLE_INTENSITY, LIGHT_TRIANGLE_INTENSITY, LIGHT_TRIANGLE_INTENSITY};
            std::fill(ls.attrs.normals.begin(), ls.attrs.normals.end(), tr.normal());
            std::fill(ls.attrs.tex_coords.begin(), ls.attrs.tex_coords.end(), geometry::vec2(0, 0));
            std::fill(ls.attrs.tangents.begin(), ls.attrs.tangents.end(), geometry::vec3(1, 0, 0));
            scene.samples = std::strtol(argv[5], nullptr, 10);
            Image img(width, height, scene.bg_color);
            run_raytracer(scene, img);
            std::ofstream out(argv[6], std::ios::binary);
            img.write(out);
            return 0;
        }
        for (auto *v : {&tr.a(), &tr.b(), &tr.c()})
            for (float c : v->val)
                put_f32(o, c);
        for (float c : tr.normal().val)
            put_f32(o, c);
        put_f32(o, LIGHT_TRIANGLE_INTENSITY);
        geometry::material dflt;
        for (float c : dflt.color.val)
            put_f32(o, c);
        put_f32(o, dflt.roughness);
        put_f32(o, dflt.metallic);
        put_f32(o, dflt.ior);
        write_words(argv[5], o);
        return 0;
    }
    RaytracerStaticContext ctx(scene);
    const geometry::Object *base = scene.objects.data();
    std::vector<uint32_t> out;

    if (mode == "bvh") {
        // [n_nodes, n_objs, root, nodes(10 each), order(n_objs)] x 2 (scene, light)
        for (const BVH *b : {&ctx.scene_bvh, &ctx.light_bvh}) {
            put_u32(out, b->nodes.size());
            put_u32(out, b->objects.size());
            put_u32(out, b->root);
            for (auto &n : b->nodes) {
                for (int k = 0; k < 3; ++k)
                    put_f32(out, n.bounding_box.vmin().val[k]);
                for (int k = 0; k < 3; ++k)
                    put_f32(out, n.bounding_box.vmax().val[k]);
                put_u32(out, n.left_child);
                put_u32(out, n.right_child);
                put_u32(out, n.obj_begin);
                put_u32(out, n.obj_end);
            }
            for (auto *o : b->objects)
                put_u32(out, (uint32_t)(o - base));
        }
        write_words(argv[5], out);
    } else if (mode == "cast" || mode == "primary") {
        std::vector<geometry::ray> rays;
        const char *out_path;
        if (mode == "cast") {
            auto f = read_floats(argv[5]);
            for (size_t i = 0; i + 5 < f.size(); i += 6)
                rays.push_back({{f[i], f[i + 1], f[i + 2]}, {f[i + 3], f[i + 4], f[i + 5]}});
            out_path = argv[6];
        } else {
            for (unsigned y = 0; y < height; ++y)
                for (unsigned x = 0; x < width; ++x)
                    rays.push_back(gen_ray(scene.camera, x, y));
            out_path = argv[5];
        }
        // per ray: prim, b, c, t, then origin/dir (6 floats) so tests can re-cast the very same rays
        for (auto &r : rays) {
            intersection_res res;
            if (ctx.scene_bvh.root != NO_CHILD)
                res = ctx.scene_bvh.intersect_ray(r, EPS, ctx.scene_bvh.root);
            put_u32(out, res.has_value() ? (uint32_t)(res->second - base) : 0xFFFFFFFFu);
            put_f32(out, res.has_value() ? res->first.x() : 0.0f);
            put_f32(out, res.has_value() ? res->first.y() : 0.0f);
            put_f32(out, res.has_value() ? res->first.z() : 0.0f);
            for (int k = 0; k < 3; ++k)
                put_f32(out, r.start.val[k]);
            for (int k = 0; k < 3; ++k)
                put_f32(out, r.dir.val[k]);
        }
        write_words(out_path, out);
    } else if (mode == "lightpdf") {
        auto f = read_floats(argv[5]);
        bvh_mix_dist dist{&ctx.light_bvh};
        for (size_t i = 0; i + 5 < f.size(); i += 6) {
            geometry::vec3 x{f[i], f[i + 1], f[i + 2]}, d{f[i + 3], f[i + 4], f[i + 5]};
            float p = ctx.light_bvh.objects.empty() ? 0.0f : dist.pdf(x, {0, 0, 1}, d);
            put_f32(out, p);
        }
        write_words(argv[6], out);
    } else if (mode == "scene") {
        // [n, then per object 9 pos + 9 normals + 6 uv + 9 tangents], camera (pos, right, up, forward, fov_x)
        put_u32(out, scene.objects.size());
        for (auto &o : scene.objects) {
            for (auto &v : o.shape.vertices)
                for (float c : v.val)
                    put_f32(out, c);
            for (auto &v : o.attrs.normals)
                for (float c : v.val)
                    put_f32(out, c);
            for (auto &v : o.attrs.tex_coords)
                for (float c : v.val)
                    put_f32(out, c);
            for (auto &v : o.attrs.tangents)
                for (float c : v.val)
                    put_f32(out, c);
        }
        for (auto *v : {&scene.camera.position, &scene.camera.right, &scene.camera.up, &scene.camera.forward})
            for (float c : v->val)
                put_f32(out, c);
        put_f32(out, scene.camera.fov_x);
        write_words(argv[5], out);
    } else {
        std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    return 0;
}
