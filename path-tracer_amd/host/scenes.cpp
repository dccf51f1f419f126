// Scene recipes of the reference (src/lib.rs:674-1077), written against the builder mirror, plus the synthetic
// one-million-sphere cloud used for deep-BVH stress. Every random number the reference takes from thread_rng
// comes from the scene's seeded stream here, in the reference's call order. The recipes are constant tables: they
// follow the reference's functions statement by statement, local names included, so that each literal can be checked
// against the line it comes from.
#include "ptscene.hpp"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

namespace ptgpu {

namespace {

struct ImageAsset {
    std::vector<uint8_t> rgb8;
    int width = 0, height = 0;
};
std::mutex g_assets_mutex;
std::map<std::string, ImageAsset> g_assets;
struct MissingAsset {
    std::string name;
};
bool build_named_scene_impl(const std::string &name, double aspect, int64_t param, Scene &s, pt_camera &camera,
                            std::string &error);

pt_camera make_camera(Vec3 lookfrom, Vec3 lookat, double vfov, double aspect, double aperture, double focus) {
    Camera cam(lookfrom, lookat, Vec3(0.0, 1.0, 0.0), vfov, aspect, aperture, focus, 0.0, 1.0);
    return cam.params();
}

// lib.rs:674-684
void two_spheres(Scene &s) {
    Material red = s.lambertian(s.constant_texture(Vec3(1.0, 0.0, 0.0)), 0.0);
    Material blue = s.lambertian(s.constant_texture(Vec3(0.0, 0.0, 1.0)), 0.0);
    std::vector<Hitable> list = {
        s.sphere(Vec3(0.0, 0.0, -1.0), 0.5, red),
        s.sphere(Vec3(0.0, 10.0, 0.0), 10.0, blue),
    };
    s.set_root(s.hitable_list(list));
}

// lib.rs:687-710
void four_spheres(Scene &s) {
    Material red = s.lambertian(s.constant_texture(Vec3(0.9, 0.0, 0.0)), 0.0);
    Material blue = s.lambertian(s.constant_texture(Vec3(0.3, 0.3, 0.3)), 0.0);
    Material green = s.lambertian(s.constant_texture(Vec3(0.0, 0.9, 0.0)), 0.0);
    Material yellow = s.lambertian(s.constant_texture(Vec3(0.9, 0.9, 0.0)), 0.0);
    Material glass = s.dielectric(1.6);
    Material metal = s.metal(Vec3::from_float(1.0), 0.0);
    std::vector<Hitable> list = {
        s.sphere(Vec3(4.0, -0.3, 0.7), 0.3, red),      s.sphere(Vec3(0.0, -100.5, -1.0), 100.0, blue),
        s.sphere(Vec3(1.0, 0.0, -1.0), 0.5, green),    s.sphere(Vec3(-1.0, 0.0, 0.0), 0.5, yellow),
        s.sphere(Vec3(0.4, -0.25, -0.3), 0.25, glass), s.sphere(Vec3(2.5, -0.15, -0.4), 0.25, glass),
        s.sphere(Vec3(2.0, 0.0, -1.0), 0.5, metal),    s.sphere(Vec3(1.6, 0.0, 1.0), 0.5, metal),
    };
    s.set_root(s.bvh_from_list(list, 0.0, 1.0));
}

// lib.rs:713-765
void random_scene(Scene &s, double t_min, double t_max) {
    SceneRng &rng = s.rng();
    Texture checker = s.checker_texture(s.constant_texture(Vec3(0.2, 0.3, 0.1)),
                                        s.constant_texture(Vec3(0.9, 0.9, 0.9)));
    std::vector<Hitable> list;
    list.push_back(s.sphere(Vec3(0.0, -1000.0, 0.0), 1000.0, s.lambertian(checker, 0.0)));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            double choose_mat = rng.rand();
            double cx = (double)a + 0.9 * rng.rand();
            double cz = (double)b + 0.9 * rng.rand();
            Vec3 center(cx, 0.2, cz);
            if (length(center - Vec3(4.0, 0.2, 0.0)) > 0.9) {
                Material material;
                bool is_emissive = false;
                if (choose_mat < 0.6) { // diffuse
                    is_emissive = rng.rand() < 0.1;
                    double emissive = is_emissive ? 30.0 : 0.0;
                    double r0 = rng.rand(), r1 = rng.rand(), r2 = rng.rand(), r3 = rng.rand(), r4 = rng.rand(),
                           r5 = rng.rand();
                    material = s.lambertian(s.constant_texture(Vec3(r0 * r1, r2 * r3, r4 * r5)), emissive);
                } else if (choose_mat < 0.8) { // metal
                    double r0 = rng.rand(), r1 = rng.rand(), r2 = rng.rand(), r3 = rng.rand();
                    material = s.metal(Vec3(0.5 * (1.0 + r0), 0.5 * (1.0 + r1), 0.5 * (1.0 + r2)), 0.2 * r3);
                } else { // glass
                    material = s.dielectric(1.5);
                }
                double radius = is_emissive ? 1.0 : 0.2;
                center.y += is_emissive ? 10.0 : 0.0;
                list.push_back(s.sphere(center, radius, material));
            }
        }
    }
    list.push_back(s.sphere(Vec3(0.0, 1.0, 0.0), 1.0, s.dielectric(1.5)));
    list.push_back(s.sphere(Vec3(-4.0, 1.0, 0.0), 1.0, s.lambertian(s.constant_texture(Vec3(0.4, 0.2, 0.1)), 0.0)));
    list.push_back(s.sphere(Vec3(4.0, 1.0, 0.0), 1.0, s.metal(Vec3(0.7, 0.6, 0.5), 0.0)));
    s.set_root(s.bvh_from_list(list, t_min, t_max));
}

// lib.rs:769-775
void two_perlin_spheres(Scene &s) {
    Texture perlin = s.noise_texture(4.0);
    std::vector<Hitable> list = {
        s.sphere(Vec3(0.0, -1000.0, 0.0), 1000.0, s.lambertian(perlin, 0.0)),
        s.sphere(Vec3(0.0, 2.0, 0.0), 2.0, s.lambertian(perlin, 0.0)),
    };
    s.set_root(s.bvh_from_list(list, 0.0, 1.0));
}

// ImageTexture::new(EARTH_TEXTURE_BYTES) (src/texture.rs:84-98, src/lib.rs:767): the reference embeds
// assets/textures/earthmap.jpg and decodes it with the `image` crate when the scene is built. This library decodes
// nothing: the host registers the decoded RGB8 texels under the asset's file name (pt_asset_register_image) before it
// builds a recipe that uses them. There is no stand-in: a recipe whose asset is missing fails.
Texture earth_texture(Scene &s) {
    std::lock_guard<std::mutex> lock(g_assets_mutex);
    auto it = g_assets.find("earthmap.jpg");
    if (it == g_assets.end()) throw MissingAsset{"earthmap.jpg"};
    return s.image_texture(it->second.rgb8.data(), it->second.width, it->second.height);
}

// lib.rs:777-782
void textured_sphere(Scene &s) {
    std::vector<Hitable> list = {
        s.sphere(Vec3(0.0, 2.0, 0.0), 2.0, s.lambertian(earth_texture(s), 0.0)),
        s.sphere(Vec3(0.0, -1000.0, 0.0), 1000.0, s.lambertian(s.constant_texture(Vec3::from_float(0.4)), 0.0)),
    };
    s.set_root(s.bvh_from_list(list, 0.0, 1.0));
}

// lib.rs:784-795
void simple_light(Scene &s) {
    Texture perlin = s.noise_texture(4.0);
    std::vector<Hitable> list;
    list.push_back(s.sphere(Vec3(0.0, -1000.0, 0.0), 1000.0, s.lambertian(perlin, 0.0)));
    list.push_back(s.sphere(Vec3(0.0, 2.0, 0.0), 2.0, s.lambertian(perlin, 0.0)));
    Material diffuse = s.diffuse_light(s.constant_texture(Vec3::from_float(0.4)));
    list.push_back(s.sphere(Vec3(0.0, 7.0, 0.0), 2.0, diffuse));
    list.push_back(s.rect(3.0, 5.0, 1.0, 3.0, -2.0, Axis::Z, diffuse));
    s.set_root(s.bvh_from_list(list, 0.0, 1.0));
}

// lib.rs:797-892
pt_camera cornell_box(Scene &s, double aspect) {
    MaterialBuilder material_builder(s);
    Material red_mat = material_builder.with_texture(s.constant_texture(Vec3(0.65, 0.05, 0.05))).lambertian();
    Material green_mat = material_builder.with_texture(s.constant_texture(Vec3(0.12, 0.45, 0.15))).lambertian();
    Material white_mat = material_builder.with_texture(s.constant_texture(Vec3::from_float(0.73))).lambertian();
    Material light = material_builder.with_texture(s.constant_texture(Vec3::from_float(15.0))).diffuse_light();
    Material alluminium = material_builder.set_albedo(Vec3(0.8, 0.85, 0.88)).metal();
    (void)alluminium;
    Material glass = material_builder.set_refraction_index(1.5).dielectric();

    SceneBuilder scene_builder(s);
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 555.0, Axis::X, green_mat)).flip_normals();
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 0.0, Axis::X, red_mat));
    scene_builder.add_hitable(s.rect(213.0, 343.0, 227.0, 332.0, 554.0, Axis::Y, light)).flip_normals();
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 555.0, Axis::Y, white_mat)).flip_normals();
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 0.0, Axis::Y, white_mat));
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 555.0, Axis::Z, white_mat)).flip_normals();
    scene_builder.add_hitable(s.sphere(Vec3(190.0, 90.0, 190.0), 90.0, glass));
    scene_builder.add_hitable(s.axis_aligned_box(Vec3::new_zero_vector(), Vec3(165.0, 330.0, 165.0), white_mat))
        .rotate_y(15.0)
        .translate(Vec3(265.0, 0.0, 295.0));
    s.set_root(scene_builder.as_bvh());
    return make_camera(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect, 0.0, 10.0);
}

// lib.rs:894-988. book_semantics flips the light so that it faces the room (the reference leaves it unflipped,
// so its DiffuseLight never emits downwards); the caller then also samples that light alone.
pt_camera cornell_smoke(Scene &s, double aspect, bool book_semantics) {
    MaterialBuilder material_builder(s);
    Material red_mat = material_builder.with_texture(s.constant_texture(Vec3(0.65, 0.05, 0.05))).lambertian();
    Material green_mat = material_builder.with_texture(s.constant_texture(Vec3(0.12, 0.45, 0.15))).lambertian();
    Material white_mat = material_builder.with_texture(s.constant_texture(Vec3::from_float(0.73))).lambertian();
    Material light = material_builder.with_texture(s.constant_texture(Vec3::from_float(7.0))).diffuse_light();

    SceneBuilder scene_builder(s);
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 555.0, Axis::X, green_mat)).flip_normals();
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 0.0, Axis::X, red_mat));
    scene_builder.add_hitable(s.rect(113.0, 443.0, 127.0, 432.0, 554.0, Axis::Y, light));
    if (book_semantics) scene_builder.flip_normals();
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 555.0, Axis::Y, white_mat)).flip_normals();
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 0.0, Axis::Y, white_mat));
    scene_builder.add_hitable(s.rect(0.0, 555.0, 0.0, 555.0, 555.0, Axis::Z, white_mat)).flip_normals();

    Hitable medium_boundary;
    {
        SceneBuilder inner(s);
        inner.add_hitable(s.axis_aligned_box(Vec3::new_zero_vector(), Vec3(165.0, 165.0, 165.0), white_mat))
            .rotate_y(-18.0)
            .translate(Vec3(130.0, 0.0, 65.0));
        medium_boundary = inner.as_hitable();
    }
    scene_builder.add_hitable(s.constant_medium(medium_boundary, 0.01, s.constant_texture(Vec3::from_float(1.0))));
    {
        SceneBuilder inner(s);
        inner.add_hitable(s.axis_aligned_box(Vec3::new_zero_vector(), Vec3(165.0, 330.0, 165.0), white_mat))
            .rotate_y(15.0)
            .translate(Vec3(265.0, 0.0, 295.0));
        medium_boundary = inner.as_hitable();
    }
    scene_builder.add_hitable(s.constant_medium(medium_boundary, 0.01, s.constant_texture(Vec3::from_float(0.0))));
    s.set_root(scene_builder.as_bvh());
    return make_camera(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect, 0.0, 10.0);
}

// lib.rs:990-1077
void final_book_two(Scene &s) {
    SceneRng &rng = s.rng();
    MaterialBuilder material_builder(s);
    Material white = material_builder.with_texture(s.constant_texture(Vec3(0.73, 0.73, 0.73))).lambertian();
    Material ground = material_builder.with_texture(s.constant_texture(Vec3(0.48, 0.83, 0.53))).lambertian();
    Material light = material_builder.with_texture(s.constant_texture(Vec3::from_float(7.0))).diffuse_light();

    SceneBuilder scene_builder(s);
    SceneBuilder floor_scene_builder(s);
    const int num_boxes = 20;
    for (int i = 0; i < num_boxes; ++i)
        for (int j = 0; j < num_boxes; ++j) {
            double w = 100.0;
            double x0 = -1000.0 + (double)i * w;
            double z0 = -1000.0 + (double)j * w;
            double y0 = 0.0;
            double x1 = x0 + w;
            double y1 = 100.0 * (rng.rand() + 0.01);
            double z1 = z0 + w;
            floor_scene_builder.add_hitable(s.axis_aligned_box(Vec3(x0, y0, z0), Vec3(x1, y1, z1), ground));
        }
    scene_builder.add_hitable(floor_scene_builder.as_bvh_node());
    scene_builder.add_hitable(s.rect(123.0, 423.0, 147.0, 412.0, 554.0, Axis::Y, light));

    Vec3 center(400.0, 400.0, 200.0);
    scene_builder.add_hitable(s.moving_sphere(center, center + Vec3(30.0, 0.0, 0.0), 0.0, 1.0, 50.0,
                                              s.lambertian(s.constant_texture(Vec3(0.7, 0.3, 0.1)), 0.0)));
    scene_builder.add_hitable(s.sphere(Vec3(260.0, 150.0, 45.0), 50.0, s.dielectric(1.5)));
    scene_builder.add_hitable(s.sphere(Vec3(0.0, 150.0, 145.0), 50.0, s.metal(Vec3(0.8, 0.8, 0.9), 10.0)));

    Hitable boundary = s.sphere(Vec3(360.0, 150.0, 145.0), 70.0, s.dielectric(1.5));
    scene_builder.add_hitable(boundary);
    scene_builder.add_hitable(s.constant_medium(boundary, 0.2, s.constant_texture(Vec3(0.2, 0.4, 0.9))));
    boundary = s.sphere(Vec3(0.0, 0.0, 0.0), 5000.0, s.dielectric(1.5));
    scene_builder.add_hitable(boundary);
    scene_builder.add_hitable(s.constant_medium(boundary, 0.0001, s.constant_texture(Vec3::from_float(1.0))));

    Material earth_material = material_builder.with_texture(earth_texture(s)).lambertian();
    scene_builder.add_hitable(s.sphere(Vec3(400.0, 200.0, 400.0), 100.0, earth_material));
    scene_builder.add_hitable(s.sphere(Vec3(220.0, 280.0, 300.0), 80.0, s.lambertian(s.noise_texture(0.1), 0.0)));

    const int ns = 1000;
    SceneBuilder sphere_scene_builder(s);
    for (int k = 0; k < ns; ++k) {
        double x = 165.0 * rng.rand(), y = 165.0 * rng.rand(), z = 165.0 * rng.rand();
        sphere_scene_builder.add_hitable(s.sphere(Vec3(x, y, z), 10.0, white));
    }
    scene_builder.add_hitable(sphere_scene_builder.as_bvh_node()).rotate_y(15.0).translate(Vec3(-100.0, 270.0, 395.0));
    s.set_root(scene_builder.as_bvh());
}

// Synthetic deep-BVH stress scene: n spheres of radius 1.5 with centres uniform in [0,1000)^3, materials
// 60 % diffuse / 20 % metal / 20 % glass as in random_scene, 2 % of the diffuse ones emissive.
pt_camera sphere_cloud(Scene &s, double aspect, int64_t n) {
    SceneRng &rng = s.rng();
    std::vector<Hitable> list;
    list.reserve((size_t)n);
    Material glass = s.dielectric(1.5);
    for (int64_t k = 0; k < n; ++k) {
        double x = 1000.0 * rng.rand(), y = 1000.0 * rng.rand(), z = 1000.0 * rng.rand();
        double choose_mat = rng.rand();
        Material material;
        if (choose_mat < 0.6) {
            bool is_emissive = rng.rand() < 0.02;
            double r0 = rng.rand(), r1 = rng.rand(), r2 = rng.rand(), r3 = rng.rand(), r4 = rng.rand(),
                   r5 = rng.rand();
            material = s.lambertian(s.constant_texture(Vec3(r0 * r1, r2 * r3, r4 * r5)), is_emissive ? 30.0 : 0.0);
        } else if (choose_mat < 0.8) {
            double r0 = rng.rand(), r1 = rng.rand(), r2 = rng.rand(), r3 = rng.rand();
            material = s.metal(Vec3(0.5 * (1.0 + r0), 0.5 * (1.0 + r1), 0.5 * (1.0 + r2)), 0.2 * r3);
        } else {
            material = glass;
        }
        list.push_back(s.sphere(Vec3(x, y, z), 1.5, material));
    }
    s.set_root(s.bvh_from_list(list, 0.0, 1.0));
    return make_camera(Vec3(500.0, 500.0, -1400.0), Vec3(500.0, 500.0, 500.0), 40.0, aspect, 0.0, 10.0);
}

} // namespace

bool build_named_scene(const std::string &name, double aspect, uint64_t seed, int64_t param, Scene &s,
                       pt_camera &camera, std::string &error) {
    (void)seed;
    try {
        return build_named_scene_impl(name, aspect, param, s, camera, error);
    } catch (const MissingAsset &m) {
        error = "missing asset: " + m.name + " (register its RGB8 texels with pt_asset_register_image first)";
        return false;
    }
}

bool register_image_asset(const std::string &name, const uint8_t *rgb8, int width, int height) {
    if (name.empty() || !rgb8 || width <= 0 || height <= 0) return false;
    ImageAsset a;
    a.rgb8.assign(rgb8, rgb8 + (size_t)width * height * 3);
    a.width = width;
    a.height = height;
    std::lock_guard<std::mutex> lock(g_assets_mutex);
    g_assets[name] = std::move(a);
    return true;
}

namespace {
bool build_named_scene_impl(const std::string &name, double aspect, int64_t param, Scene &s, pt_camera &camera,
                            std::string &error) {
    // Cameras the reference keeps in comments for the non-Cornell scenes (lib.rs:308-323).
    pt_camera book_cam = make_camera(Vec3(26.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect, 0.0, 10.0);
    if (name == "two_spheres") {
        two_spheres(s);
        camera = make_camera(Vec3(-2.0, 2.0, 1.0), Vec3(0.0, 0.0, 0.0), 40.0, aspect, 0.0, 10.0);
    } else if (name == "four_spheres") {
        four_spheres(s);
        camera = make_camera(Vec3(-2.0, 2.0, 1.0), Vec3(0.0, 0.0, 0.0), 40.0, aspect, 0.0, 10.0);
    } else if (name == "random_scene") {
        random_scene(s, 0.0, 1000.0); // lib.rs:301
        camera = book_cam;
    } else if (name == "two_perlin_spheres") {
        two_perlin_spheres(s);
        camera = book_cam;
    } else if (name == "textured_sphere") {
        textured_sphere(s);
        camera = book_cam;
    } else if (name == "simple_light") {
        simple_light(s);
        camera = book_cam;
    } else if (name == "cornell_box") {
        camera = cornell_box(s, aspect);
    } else if (name == "cornell_smoke") {
        camera = cornell_smoke(s, aspect, param == 1);
    } else if (name == "final_book_two") {
        final_book_two(s);
        camera = make_camera(Vec3(478.0, 278.0, -600.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect, 0.0, 10.0);
    } else if (name == "sphere_cloud") {
        if (param <= 0) param = 1000000;
        camera = sphere_cloud(s, aspect, param);
    } else {
        error = "unknown scene name: " + name;
        return false;
    }
    if (name == "final_book_two" && param == 1) {
        // the scene's own light instead of the Cornell pair: the glass-sphere light shape of trace.rs:125 lies in the
        // middle of this scene's floor and media, and every sample taken from inside it is NaN (sphere.rs:92-94)
        Material dummy = s.dummy_material();
        s.set_lights({s.rect(123.0, 423.0, 147.0, 412.0, 554.0, Axis::Y, dummy)});
    } else if (name == "cornell_smoke" && param == 1) {
        // book variant: importance-sample this scene's own light only. The reference's hard-coded pair
        // (trace.rs:123-128) includes a sphere at (190,90,190) that the smoke box overlaps, and Sphere::pdf_value
        // from inside it is sqrt of a negative number (sphere.rs:92-94).
        Material dummy = s.dummy_material();
        s.set_lights({s.rect(113.0, 443.0, 127.0, 432.0, 554.0, Axis::Y, dummy)});
    } else {
        s.set_default_lights(); // trace.rs:123-128: the same two light shapes for every scene
    }
    if (!s.error().empty()) {
        error = s.error();
        return false;
    }
    return true;
}
} // namespace

} // namespace ptgpu
